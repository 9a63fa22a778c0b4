"""Where the eight-wave attention forward spends its time: shader-clock readings (s_memtime, lane 0 of every wave) from an instrumented build of attention.hip
(-DATTN_STAMPS, tools/bin/libattn_stamps.so; not part of the library).  Prints, over all waves, the median / p10 / p90 of every stamp relative to the wave's first."""
import ctypes as C, math, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "bin", "libattn_stamps.so")
src = os.path.join(os.path.dirname(here), "huggingface_asr_amd", "csrc")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(src, "attention.hip")):      # the instrumented build is not part of the library: made here, on demand
    import subprocess, tempfile
    os.makedirs(os.path.dirname(so), exist_ok=True)
    with tempfile.NamedTemporaryFile("w", suffix=".cpp", delete=False) as f:
        f.write('extern "C" void mi_record_hip_error(int, const char*, int) {}\n')
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-result", "-munsafe-fp-atomics", "-Xclang", "-target-feature", "-Xclang",
                    "-packed-fp32-ops", "-DATTN_STAMPS", "-shared", os.path.join(src, "attention.hip"), f.name, "-o", so], check=True, stderr=subprocess.DEVNULL)
L = C.CDLL(so)
dev = "cuda:0"
B, T, H, hd = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (32, 250, 4, 128)))
rel = int(sys.argv[5]) if len(sys.argv) > 5 else 1
d = H * hd
g = torch.Generator(device=dev).manual_seed(1)
qkv = (torch.randn(B * T, 3 * d, device=dev, generator=g) * 0.5).to(torch.bfloat16)
pos = (torch.randn(2 * T - 1, d, device=dev, generator=g) * 0.5).to(torch.bfloat16)
u = torch.randn(d, device=dev, generator=g) * 0.1
v = torch.randn(d, device=dev, generator=g) * 0.1
out = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
nblk = B * H * ((T + 127) // 128)
st = torch.zeros(nblk * 8 * 32, dtype=torch.int64, device=dev)
vp, i64, i32, f32 = C.c_void_p, C.c_long, C.c_int, C.c_float
L.mi_attention_qkv_stamps.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, i32, vp, vp]
def call():
    rc = L.mi_attention_qkv_stamps(qkv.data_ptr(), 3 * d, qkv.data_ptr() + 2 * d, 3 * d, qkv.data_ptr() + 4 * d, 3 * d, pos.data_ptr() if rel else None, d, u.data_ptr() if rel else None, v.data_ptr() if rel else None,
                                   None, out.data_ptr(), d, B, T, H, hd, 1.0 / math.sqrt(hd), 0, st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
for _ in range(3): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
print(f"launch (events around one call): {e0.elapsed_time(e1) * 1000:.1f} us")
s = st.cpu().numpy().reshape(nblk, 8, 32).astype(np.int64)
names = {0: "entry", 1: "prologue DMA issued", 2: "prologue vmcnt(0)", 3: "prologue barrier", 24: "loop done", 25: "merge barrier 1", 26: "exchange written", 27: "merge barrier 2", 28: "stores issued", 29: "stores landed"}
for uu in range(4):
    names.update({4 + 3 * uu: f"interval {uu} top", 5 + 3 * uu: f"interval {uu} waits done", 6 + 3 * uu: f"interval {uu} barrier"})
names.update({16: "2nd tile P1: first reads issued", 17: "2nd tile P1: DMA issued", 18: "2nd tile P1: MFMA stream done", 19: "2nd tile P2: reads issued", 20: "2nd tile P2: DMA issued", 21: "2nd tile P2: soft-max done"})
names.update({22: "end of P1 of the wave's 2nd tile", 23: "end of P2 of the wave's 2nd tile"})
rel0 = s - s[:, :, :1]
blk0 = s[:, :, 0].min(axis=1)
print(f"clock: block lifetime (first entry -> last 'stores landed') median {np.median(s[:, :, 29].max(axis=1) - blk0):.0f} cycles")
print(f"entry spread over all blocks: {s[:, :, 0].max() - s[:, :, 0].min()} cycles (clocks of different XCDs may not be aligned)")
for half in (0, 1):
  print(f" waves with s = {half}:")
  for i in sorted(names):
    col = rel0[:, 4 * half:4 * half + 4, i].reshape(-1)
    col = col[s[:, 4 * half:4 * half + 4, i].reshape(-1) != 0]
    if col.size == 0: continue
    print(f"    {i:2d} {names[i]:32s} median {np.median(col):8.0f}  p10 {np.percentile(col, 10):8.0f}  p90 {np.percentile(col, 90):8.0f}")
