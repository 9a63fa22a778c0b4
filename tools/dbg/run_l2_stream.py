import os, sys, ctypes, torch
here = os.path.dirname(os.path.abspath(__file__))
S = ctypes.CDLL(os.path.join(here, "libl2_stream.so"))
S.stream_launch.restype = ctypes.c_int
S.stream_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
buf = torch.randint(0, 255, (1 << 30,), dtype=torch.uint8, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
names = {0: "LDS-DMA (global_load_lds x4)", 1: "global_load_dwordx4 -> VGPR", 2: "global_load + ds_write_b128", 3: "half LDS-DMA, half VGPR"}
CLK = 2.4e9
for region_mb in (8, 24, 1000):
    region = region_mb << 20
    for blocks in (256, 512):
        for mode in (0, 1, 2, 3):
            row = []
            for depth in (1, 2, 3, 4):
                iters = 2000
                st = torch.cuda.current_stream().cuda_stream
                for _ in range(2): S.stream_launch(mode, depth, buf.data_ptr(), region, blocks, iters, sink.data_ptr(), st)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3): S.stream_launch(mode, depth, buf.data_ptr(), region, blocks, iters, sink.data_ptr(), st)
                e1.record(); torch.cuda.synchronize()
                dt = e0.elapsed_time(e1) * 1e-3 / 3
                bytes_ = blocks * iters * 16384
                row.append(f"d{depth}: {bytes_ / dt / 1e12:5.2f} TB/s {bytes_ / dt / CLK / 256:5.1f} B/clk/CU")
            print(f"region {region_mb:4d} MB blocks {blocks} {names[mode]:32s} " + " | ".join(row), flush=True)
