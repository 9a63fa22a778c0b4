import os, sys, ctypes, numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
V = ctypes.CDLL(os.path.join(here, "libdma_victim.so"))
V.aggressor_launch.restype = ctypes.c_int
V.aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
V.victim_launch.restype = ctypes.c_int
V.victim_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
big = torch.randn(64 * 1024 * 1024, device=dev)           # 256 MiB
big2 = torch.randn(64 * 1024 * 1024, device=dev)
table = torch.ones(64, device=dev)
sink = torch.zeros(4, device=dev)
counter = torch.zeros(2, dtype=torch.int32, device=dev)
recs = torch.zeros(1024, 5, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
def run(mode, label, ablocks=512, aiters=4000, vblocks=512, viters=3000):
    counter.zero_(); recs.zero_(); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    with torch.cuda.stream(side):
        assert V.victim_launch(table.data_ptr(), big2.data_ptr(), big2.numel() * 4, vblocks, viters, counter.data_ptr(), recs.data_ptr(), 1024, side.cuda_stream) == 0
    if mode >= 0:
        assert V.aggressor_launch(mode, big.data_ptr(), big.numel() * 4, ablocks, aiters, sink.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    t1.record(); torch.cuda.synchronize()
    n = int(counter[0].item())
    print(f"{label}: {n} bad words   ({t0.elapsed_time(t1):.1f} ms)", flush=True)
    if n:
        r = recs[:min(n, 1024)].cpu().numpy()
        lanes = np.unique(r[:, 1] & 63); words = np.unique(r[:, 3]); vals = np.unique(r[:, 4].view(np.uint32))
        print("   lanes", lanes.tolist()[:64], "words", words.tolist(), "values", [hex(v) for v in vals[:8]])
run(-1, "victim alone")
for rep in range(2):
    run(0, "victim || aggressor global_load_lds_dwordx4")
    run(1, "victim || aggressor raw_buffer_load_lds 16B")
    run(2, "victim || aggressor plain loads + ds_write")
