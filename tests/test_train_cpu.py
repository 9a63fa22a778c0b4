"""CPU tests of the training step's host logic (no kernels): the flat parameter store's packed layout <-> the reference's
state-dict names, the weight-decay mask (HF Trainer rule), and the data-parallel gradient all-reduce over gloo, world size 2."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from huggingface_asr_amd import shapes, synth
from huggingface_asr_amd.train import GradSync, ParamStore, _enc_map, encoder_specs


def _sd(cfg, seed=3):
    return {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), seed).items()}


@pytest.mark.parametrize("extra", [{}, {"position_embeddings_type": "rotary"}])
def test_packed_layout_roundtrips_every_reference_parameter(extra):
    cfg = dict(shapes.TINY, **extra)
    sd = _sd(cfg)
    specs = encoder_specs(cfg)
    mp_ = _enc_map(cfg)
    store = ParamStore(specs, "cpu")
    covered = {}
    for s in specs:
        packed = mp_[s.name][0](sd).reshape(s.shape)
        store.p(s.name).copy_(packed)
        for key, back in mp_[s.name][1]:
            covered[key] = back(store.p(s.name))
    assert set(covered) == set(sd), (sorted(set(sd) - set(covered))[:5], sorted(set(covered) - set(sd))[:5])
    for k, v in sd.items():
        assert covered[k].shape == v.shape and torch.equal(covered[k], v), k
    # offsets are 64-element aligned (16-B aligned bf16 rows for the GEMM loads), ranges do not overlap
    offs = sorted((store.off[n], n) for n in store.order)
    for (o, n), (o2, _) in zip(offs, offs[1:]):
        assert o % 64 == 0 and o + torch.Size(store.specs[n].shape).numel() <= o2


def test_weight_decay_mask_follows_hf_trainer_rule():
    """HF Trainer: decay everything except LayerNorm parameters and names containing 'bias' (so pos_bias_u / pos_bias_v are excluded)."""
    cfg = dict(shapes.TINY)
    specs = {s.name: s for s in encoder_specs(cfg)}
    assert specs["l0.ff1_w1"].decay and specs["conv1_w"].decay and specs["l0.csgu_w"].decay and specs["l1.mrg_dw_w"].decay and specs["head_w"].decay
    for n in ("l0.ff1_b1", "l0.att_ln_g", "l0.att_ln_b", "l0.att_u", "l0.att_v", "head_b", "enc_ln_g", "l1.csgu_ln_g", "conv2_b"):
        assert not specs[n].decay, n
    store = ParamStore(list(specs.values()), "cpu")
    o = store.off["l0.ff1_w1"]
    assert int(store.decay[o]) == 1 and int(store.decay[store.off["l0.ff1_b1"]]) == 0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    from huggingface_asr_amd import parallel as P
    P.init("gloo")
    n = 1000
    for overlap in (False, True):                                     # one merged collective after the backward / one per bucket as it is final
        flat = torch.arange(n, dtype=torch.float32) * (rank + 1)      # stand-in for this rank's flat gradient
        sync = GradSync(flat, overlap=overlap)
        assert sync.on and sync.world == world and sync.overlap == overlap
        for lo, hi in ((768, 1000), (256, 768), (0, 256)):           # head bucket, layer buckets in reverse order, front end
            sync.launch(lo, hi)
        assert len(sync.pending) == (3 if overlap else 0)
        sync.wait()
        assert not sync.pending and sync._span is None
        out[(rank, overlap)] = flat.clone()
    P.barrier()
    torch.distributed.destroy_process_group()


def test_gradsync_gloo_world2_sums_every_bucket():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    want = torch.arange(1000, dtype=torch.float32) * 3
    for key in ((0, False), (1, False), (0, True), (1, True)):
        assert torch.equal(out[key], want), key


def test_gradsync_single_process_is_a_noop():
    flat = torch.ones(10)
    s = GradSync(flat)
    s.launch(0, 10); s.wait()
    assert not s.on and s.world == 1 and torch.equal(flat, torch.ones(10))


def test_decoder_layout_roundtrips_reference_names():
    """GPT-2 Conv1D weights are stored (in, out) in the reference: packed as (out, in) rows for the GEMM, transposed back on export."""
    from helpers import TINY_DEC
    from huggingface_asr_amd.train_aed import _dec_map, decoder_specs
    for fixed in (False, True):
        c = dict(TINY_DEC, pos_emb_fixed=fixed, tie_word_embeddings=False)
        specs = decoder_specs(c, 64, True)
        mp_ = _dec_map(c, True)
        d, V, L = c["n_embd"], c["vocab_size"], c["n_layer"]
        sd = {"enc_to_dec_proj.weight": torch.randn(d, 64), "enc_to_dec_proj.bias": torch.randn(d), "decoder.lm_head.weight": torch.randn(V, d),
              "decoder.additional_lm_heads.0.weight": torch.randn(V, d), "decoder.transformer.ln_f.weight": torch.randn(d), "decoder.transformer.ln_f.bias": torch.randn(d)}
        if fixed:
            sd["decoder.transformer.wte.emb_layers.0.weight"] = torch.randn(V, d)
        else:
            sd["decoder.transformer.wte.weight"] = torch.randn(V, d); sd["decoder.transformer.wpe.weight"] = torch.randn(c["n_positions"], d)
        for l in range(L):
            r = f"decoder.transformer.h.{l}."
            for n, shp in (("ln_1", None), ("ln_cross_attn", None), ("ln_2", None)):
                sd[r + n + ".weight"] = torch.randn(d); sd[r + n + ".bias"] = torch.randn(d)
            for n, (i, o) in (("attn.c_attn", (d, 3 * d)), ("attn.c_proj", (d, d)), ("crossattention.q_attn", (d, d)), ("crossattention.c_attn", (d, 2 * d)),
                              ("crossattention.c_proj", (d, d)), ("mlp.c_fc", (d, 4 * d)), ("mlp.c_proj", (4 * d, d))):
                sd[r + n + ".weight"] = torch.randn(i, o); sd[r + n + ".bias"] = torch.randn(o)
        store = ParamStore(specs, "cpu")
        covered = {}
        for s_ in specs:
            store.p(s_.name).copy_(mp_[s_.name][0](sd).reshape(s_.shape))
            for key, back in mp_[s_.name][1]:
                covered[key] = back(store.p(s_.name))
        assert set(covered) == set(sd)
        for k, v in sd.items():
            assert torch.equal(covered[k], v), k
        assert store.specs["h0.wqkv"].shape == (3 * d, d) and store.specs["h0.wpr"].shape == (d, 4 * d)
