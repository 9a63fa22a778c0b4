import ctypes, os, torch
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstore.so"))
L.store_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for (M, N) in ((8000, 2048), (8000, 512), (8000, 1536)):
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda:0")
    for mode in (0, 1, 0, 1):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3): L.store_launch(mode, C.data_ptr(), N, M, N, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # interleave with a big unrelated write so the target is not L2-resident from the previous launch
        big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda:0")
        ts = []
        for _ in range(10):
            big.zero_()
            e0.record(); L.store_launch(mode, C.data_ptr(), N, M, N, st); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"M {M} N {N} mode {mode}: {ts[len(ts) // 2]:6.1f} us  ({M * N * 2 / ts[len(ts) // 2] / 1e6:5.2f} TB/s)", flush=True)
