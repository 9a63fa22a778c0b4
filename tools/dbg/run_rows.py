import os, sys, ctypes, torch
here = os.path.dirname(os.path.abspath(__file__))
S = ctypes.CDLL(os.path.join(here, "libl2_stream.so"))
S.rows_launch.restype = ctypes.c_int
S.rows_launch.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
buf = torch.randint(0, 255, (1 << 28,), dtype=torch.uint8, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
CLK = 2.4e9
def run(label, ld, rows, ksteps, kb=128, blocks=512, tiles=40):
    nrb = rows // 256
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): S.rows_launch(buf.data_ptr(), ld, nrb, ksteps, tiles, kb, blocks, sink.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): S.rows_launch(buf.data_ptr(), ld, nrb, ksteps, tiles, kb, blocks, sink.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / 3
    b = blocks * tiles * ksteps * 32768
    print(f"{label:58s} {b / dt / 1e12:6.2f} TB/s {b / dt / CLK / 256:5.1f} B/clk/CU", flush=True)
run("K=512  (ld 1024 B), 8192 rows (8 MB), 8 steps/tile", 1024, 8192, 8)
run("K=2048 (ld 4096 B), 8192 rows (32 MB), 32 steps/tile", 4096, 8192, 32)
run("K=512 padded ld 1024+128 B", 1152, 8192, 8)
run("K=2048 padded ld 4096+128 B", 4224, 8192, 32)
run("K=512, 2048 rows only (W-like, 2 MB)", 1024, 2048, 8)
run("contiguous slab: ld 128 B (rows back to back), 32 KB/step", 128, 65536, 1, kb=0)
run("K=512 one block per CU", 1024, 8192, 8, blocks=256)
