"""Independent forward steps in flight together (serving / throughput mode).

One forward of the encoder is a chain of ~240 dependent launches, most of them one tile per CU: each has a fill and a drain that nothing in the SAME step can
cover.  A second, independent step can: `ForwardPipeline` keeps `lanes` engines (each with its own workspace; the same weights) on `lanes` HIP streams and hands
consecutive submissions to consecutive lanes.  A submission's results are the bits the engine gives alone (tests/test_gpu_encoder.py).  Measured on the headline
config (base encoder, 32 x 10 s per step): 4.97 -> 4.24 ms per step with two lanes, 4.16 with three (tools/two_stream_bench.py).

The library's own kernels contain no packed-f32 arithmetic (csrc/build.py enforces it), which is what made co-resident kernels from two streams safe
("Concurrent kernels"); the training path stays single-stream (RCCL's kernels are not ours to rebuild)."""
from __future__ import annotations

import os

import torch

from .engine import EBranchformerEngine


def reserve_hw_queues(lanes: int) -> int:
    """The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and reads that variable once, when it initialises:
    with four lanes next to the default stream two of them take turns on one queue (measured, base encoder, wide tiles: 4.61 ms per step; with 8 queues 3.85).
    Call this BEFORE the first HIP call of the process (torch.cuda.is_available() is one); it leaves a value the user exported alone.
    -> the queue count the runtime will see."""
    want = 8 if lanes > 3 else 4
    have = os.environ.get("GPU_MAX_HW_QUEUES")
    if have is None and want > 4:
        if torch.cuda.is_initialized():
            raise RuntimeError("reserve_hw_queues: the HIP runtime is already initialised; export GPU_MAX_HW_QUEUES=8 before starting the process")
        os.environ["GPU_MAX_HW_QUEUES"] = str(want)
        return want
    return int(have) if have is not None else 4


class ForwardPipeline:
    def __init__(self, cfg: dict, device, state_dict: dict, lanes: int = 2, wide_tiles: bool = False):
        if lanes < 1:
            raise ValueError("lanes >= 1")
        self.device = torch.device(device)
        if lanes > 3 and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
            import warnings
            warnings.warn("ForwardPipeline: more than 3 lanes on the HIP runtime's default 4 hardware queues — two lanes will take turns on one queue (measured: 4.6 instead of "
                          "3.8 ms per step); call pipeline.reserve_hw_queues(lanes) before the first HIP call of the process, or export GPU_MAX_HW_QUEUES=8", RuntimeWarning)
        self.engines = []
        for i in range(lanes):
            e = EBranchformerEngine(cfg, self.device)
            e.wide_tiles = bool(wide_tiles)          # mi_ebf_config.wide_tiles: fewer, fatter GEMM blocks per launch — pays with >= 3 lanes
            if i == 0:
                e.load_state_dict(state_dict)
            else:
                e.share_weights_from(self.engines[0])        # one packed weight table for every lane; a lane owns its workspace and position cache only
            self.engines.append(e)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(lanes)] if lanes > 1 else [None]
        self.events = [None] * lanes                 # the event recorded behind each lane's latest submission (wait on it before reading that lane's results on another stream)
        self._next = 0

    @property
    def lanes(self) -> int:
        return len(self.engines)

    def submit(self, fn):
        """fn(engine, lane) is enqueued on the next lane's stream (behind that lane's earlier work only) -> (lane, fn's return value).
        `self.events[lane]` is recorded behind it: wait on that event (or torch.cuda.synchronize()) before reading the results on another stream."""
        lane = self._next
        self._next = (lane + 1) % len(self.engines)
        st = self.streams[lane]
        if st is None:
            return lane, fn(self.engines[lane], lane)
        with torch.cuda.stream(st):
            out = fn(self.engines[lane], lane)
            ev = torch.cuda.Event()
            ev.record(st)
            self.events[lane] = ev
            return lane, out

    def reset(self):
        self._next = 0
