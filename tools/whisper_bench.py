"""BASELINE.json config 4: Whisper-small encoder (d=768, 12 layers, 12 heads, ffn 3072), 30 s / 80-mel inputs, bf16, B=16 on 1 GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import synth
from huggingface_asr_amd.whisper import WhisperEncoderEngine, WhisperFrontend

dev = "cuda:0"
cfg = dict(d_model=768, encoder_layers=12, encoder_attention_heads=12, encoder_ffn_dim=3072)
d, F = 768, 3072
sd = {}
def P(n, s): sd[n] = torch.from_numpy(synth.init_param(0, n, s))
P("conv1.weight", (d, 80, 3)); P("conv1.bias", (d,)); P("conv2.weight", (d, d, 3)); P("conv2.bias", (d,)); P("embed_positions.weight", (1500, d))
P("layer_norm.weight", (d,)); P("layer_norm.bias", (d,))
for l in range(12):
    p = f"layers.{l}."
    for n, s in [("self_attn_layer_norm.weight", (d,)), ("self_attn_layer_norm.bias", (d,)), ("self_attn.q_proj.weight", (d, d)), ("self_attn.q_proj.bias", (d,)),
                 ("self_attn.k_proj.weight", (d, d)), ("self_attn.v_proj.weight", (d, d)), ("self_attn.v_proj.bias", (d,)), ("self_attn.out_proj.weight", (d, d)),
                 ("self_attn.out_proj.bias", (d,)), ("final_layer_norm.weight", (d,)), ("final_layer_norm.bias", (d,)), ("fc1.weight", (F, d)), ("fc1.bias", (F,)),
                 ("fc2.weight", (d, F)), ("fc2.bias", (d,))]:
        P(p + n, s)
eng = WhisperEncoderEngine(cfg, dev); eng.load_state_dict(sd)
B = int(os.environ.get("B", "16"))
wave = torch.from_numpy(synth.waveforms(5, B, 480000)).to(dev)
fe = WhisperFrontend(80)
def step():
    _, cl = fe(wave, want_features=False)
    return eng.forward(features_cl=cl)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 10
for _ in range(K): out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"whisper-small encoder + log-mel, B={B} x 30 s: {dt*1e3:.2f} ms/step -> {B*30/dt:.0f} audio-s/s; out {tuple(out.shape)}")
# independent batches in flight on streams of their own (the engine holds only weights: the same engine serves every lane), as bench.py does for the headline config
for lanes in (2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    K = 12 * lanes
    def run(n):
        for j in range(n):
            with torch.cuda.stream(streams[j % lanes]):
                step()
    run(2 * lanes)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"  {lanes} batches in flight: {dt*1e3:.2f} ms/step -> {B*30/dt:.0f} audio-s/s")
