"""BASELINE.json config 5: DeCRED_base-shaped encoder-decoder (E-Branchformer-base + 8x512 GPT-2 with an auxiliary head),
bs=1, one synthetic 10 s clip: greedy and CTC-joint beam decoding latency on 1 GPU (random weights -> fixed-length decode)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import fbank as FB, shapes, synth
from huggingface_asr_amd.decoder import JointAEDEngine, generate, generate_stepwise

dev = "cuda:0"
enc_cfg = dict(shapes.BASE, vocab_size=5000, ctc_zero_infinity=True, ctc_loss_reduction="mean")
dec_cfg = dict(vocab_size=5001, n_embd=512, n_layer=8, n_head=8, n_positions=256, head_locations=[5], head_weights=[0.4, 0.6], lsm_factor=0.1,
               pos_emb_fixed=True)
jcfg = dict(ctc_weight=0.3, pad_token_id=5000, decoder_start_token_id=2)
names = dict(shapes.param_shapes(enc_cfg))
sd = {"encoder." + k: torch.from_numpy(synth.init_param(0, "encoder." + k, s)) for k, s in names.items()}
d, V = 512, 5001
def P(n, s): sd[n] = torch.from_numpy(synth.init_param(0, n, s))
P("decoder.transformer.wte.emb_layers.0.weight", (V, d))
for l in range(8):
    p = f"decoder.transformer.h.{l}."
    for n, s in [("ln_1.weight", (d,)), ("ln_1.bias", (d,)), ("attn.c_attn.weight", (d, 3 * d)), ("attn.c_attn.bias", (3 * d,)),
                 ("attn.c_proj.weight", (d, d)), ("attn.c_proj.bias", (d,)), ("ln_cross_attn.weight", (d,)), ("ln_cross_attn.bias", (d,)),
                 ("crossattention.q_attn.weight", (d, d)), ("crossattention.q_attn.bias", (d,)), ("crossattention.c_attn.weight", (d, 2 * d)),
                 ("crossattention.c_attn.bias", (2 * d,)), ("crossattention.c_proj.weight", (d, d)), ("crossattention.c_proj.bias", (d,)),
                 ("ln_2.weight", (d,)), ("ln_2.bias", (d,)), ("mlp.c_fc.weight", (d, 4 * d)), ("mlp.c_fc.bias", (4 * d,)),
                 ("mlp.c_proj.weight", (4 * d, d)), ("mlp.c_proj.bias", (d,))]:
        P(p + n, s)
P("decoder.transformer.ln_f.weight", (d,)); P("decoder.transformer.ln_f.bias", (d,))
P("decoder.lm_head.weight", (V, d)); P("decoder.additional_lm_heads.0.weight", (V, d))
eng = JointAEDEngine(enc_cfg, dec_cfg, jcfg, dev)
eng.load_state_dict(sd)
wave = torch.from_numpy(synth.waveforms(1, 1, 160000)).to(dev)
tb = FB.FbankTables(80)
for W, maxlen in ((1, 40), (5, 40)):
    res = {}
    for name, fn in (("device-resident loop", generate), ("host loop", generate_stepwise)):
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            feats, frames = FB.fbank_gpu(wave, tb, pad_frames_to=100)
            out = fn(eng, feats, frames, num_beams=W, max_length=maxlen, ctc_weight=0.3, eos_token_id=1)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        n = len(out[0]["tokens"])
        res[name] = out
        print(f"beams={W} {name}: {dt*1e3:.1f} ms end-to-end (fbank+encoder+{n} tokens), {dt*1e3/max(n-1,1):.2f} ms/token incl. encoder")
    print(f"beams={W}: same hypotheses {res['device-resident loop'][0]['hypotheses'] == res['host loop'][0]['hypotheses']}")
