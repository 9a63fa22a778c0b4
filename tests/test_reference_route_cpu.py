"""CPU: the reference's OWN model-instantiation code lands on the HIP classes after `bind.install()`.

`/root/reference/src/utilities/model_utils.py` imports `JointCTCAttentionEncoderDecoder{,Config}` by name and builds them directly (:193, :199), so swapping
`bind_all` alone is not enough.  The scenario runs in a subprocess (it edits sys.modules / the Auto registry): tests/ref_route_script.py imports the reference's
`utilities.model_utils` (absent third-party modules stubbed), installs, and calls `instantiate_aed_model` down all three branches plus `instantiate_ctc_model`.
Needs the reference tree (this container only; skipped where /root/reference is absent, e.g. on the GPU box)."""
import os
import subprocess
import sys
import textwrap

import pytest

REF_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF_SRC, "utilities")), reason="reference tree not present")


def _run(args, cwd=ROOT):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=ROOT)
    return subprocess.run([sys.executable, *args], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)


@needs_ref
@pytest.mark.parametrize("order", ["import_first", "install_first"])
def test_reference_instantiate_functions_build_hip_models(tmp_path, order):
    r = _run([os.path.join(HERE, "ref_route_script.py"), REF_SRC, str(tmp_path), order])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = r.stdout
    assert "ALL OK" in out
    for what in ("instantiate_aed_model(from_encoder_decoder_config) huggingface_asr_amd.modeling_joint.JointCTCAttentionEncoderDecoder decoder huggingface_asr_amd.modeling_joint.GPT2LMMultiHeadModel",
                 "instantiate_aed_model(from_encoder_decoder_pretrained) huggingface_asr_amd.modeling_joint.JointCTCAttentionEncoderDecoder",
                 "instantiate_aed_model(from_pretrained) huggingface_asr_amd.modeling_joint.JointCTCAttentionEncoderDecoder",
                 "instantiate_ctc_model huggingface_asr_amd.modeling_ebranchformer",
                 "CustomModelForCausalLM.from_config -> huggingface_asr_amd.modeling_joint",
                 "whisper branch (model_utils.py:183) -> transformers class, huggingface_asr_amd.whisper encoder forward"):
        assert "OK " + what in out, out[-3000:]


@needs_ref
def test_launcher_runs_a_script_on_the_hip_classes(tmp_path):
    """`python -m huggingface_asr_amd.launch <script>`: the script's own `from ... import` lines receive the HIP classes and our bind_all."""
    script = tmp_path / "trainer_like.py"
    script.write_text(textwrap.dedent('''
        import sys
        from utilities.bind import bind_all
        from models.ctc_encoder_plus_autoregressive_decoder import JointCTCAttentionEncoderDecoder, JointCTCAttentionEncoderDecoderConfig
        from models.encoders.e_branchformer import Wav2Vec2EBranchformerForCTC
        from models.decoders.multi_head_gpt2 import GPT2LMMultiHeadModel
        if __name__ == "__main__":
            bind_all()
            from transformers import AutoModelForSpeechSeq2Seq
            print("ARGV", sys.argv[1:])
            print("CLASSES", JointCTCAttentionEncoderDecoder.__module__, Wav2Vec2EBranchformerForCTC.__module__, GPT2LMMultiHeadModel.__module__, bind_all.__module__)
            print("AUTO", AutoModelForSpeechSeq2Seq._model_mapping[JointCTCAttentionEncoderDecoderConfig].__module__)
    '''))
    r = _run(["-m", "huggingface_asr_amd.launch", "--reference-src", REF_SRC, str(script), "--flag=1"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "ARGV ['--flag=1']" in r.stdout
    assert "CLASSES huggingface_asr_amd.modeling_joint huggingface_asr_amd.modeling_ebranchformer huggingface_asr_amd.modeling_joint huggingface_asr_amd.bind" in r.stdout
    assert "AUTO huggingface_asr_amd.modeling_joint" in r.stdout


def test_from_encoder_decoder_pretrained_kwarg_routing(tmp_path):
    """no reference needed: encoder_/decoder_ prefixes go to the sub-configs, the rest to the joint config (reference :138-235)"""
    from huggingface_asr_amd import shapes
    from huggingface_asr_amd.bind import bind_all
    from huggingface_asr_amd.configuration_ebranchformer import Wav2Vec2EBranchformerConfig
    from huggingface_asr_amd.modeling_ebranchformer import Wav2Vec2EBranchformerForCTC
    from huggingface_asr_amd.modeling_joint import GPT2LMMultiHeadModel, GPT2MultiHeadConfig, JointCTCAttentionEncoderDecoder
    bind_all()
    base = dict(shapes.TINY); base.pop("num_fbanks")
    enc = Wav2Vec2EBranchformerForCTC(Wav2Vec2EBranchformerConfig(**base))
    dec = GPT2LMMultiHeadModel(GPT2MultiHeadConfig(vocab_size=base["vocab_size"], n_embd=32, n_layer=1, n_head=2, n_positions=32, lsm_factor=0.0, pos_emb_fixed=False,
                                                   add_cross_attention=True, is_decoder=True))
    enc.save_pretrained(tmp_path / "e"); dec.save_pretrained(tmp_path / "d")
    m = JointCTCAttentionEncoderDecoder.from_encoder_decoder_pretrained(str(tmp_path / "e"), str(tmp_path / "d"), encoder_layerdrop=0.0, decoder_lsm_factor=0.2,
                                                                        ctc_weight=0.25, decoder_start_token_id=1, pad_token_id=3)
    assert m.config.ctc_weight == 0.25 and m.config.decoder_start_token_id == 1 and m.config.pad_token_id == 3
    assert m.config.decoder.lsm_factor == 0.2 and m.lsm_factor == 0.2 and m.config.encoder.layerdrop == 0.0
    assert m.enc_loss_weight == 0.25 and abs(m.dec_loss_weight - 0.75) < 1e-12
    assert hasattr(m, "enc_to_dec_proj")                      # 64 -> 32
    with pytest.raises(ValueError):
        JointCTCAttentionEncoderDecoder.from_encoder_decoder_pretrained(None, str(tmp_path / "d"))
    with pytest.raises(TypeError):                            # a torch decoder is refused: no silent PyTorch fallback
        from transformers import GPT2Config, GPT2LMHeadModel
        JointCTCAttentionEncoderDecoder(encoder=enc, decoder=GPT2LMHeadModel(GPT2Config(n_layer=1, n_embd=32, n_head=2, vocab_size=50)))
