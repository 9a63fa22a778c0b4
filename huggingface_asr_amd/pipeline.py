"""Independent forward steps in flight together (serving / throughput mode).

One forward of the encoder is a chain of ~240 dependent launches, most of them one tile per CU: each has a fill and a drain that nothing in the SAME step can
cover.  A second, independent step can: `ForwardPipeline` keeps `lanes` engines (each with its own workspace; the same weights) on `lanes` HIP streams and hands
consecutive submissions to consecutive lanes.  A submission's results are the bits the engine gives alone (tests/test_gpu_encoder.py).  Measured on the headline
config (base encoder, 32 x 10 s per step): 4.97 -> 4.24 ms per step with two lanes, 4.16 with three (tools/two_stream_bench.py).

The library's own kernels contain no packed-f32 arithmetic (csrc/build.py enforces it), which is what made co-resident kernels from two streams safe
("Concurrent kernels"); the training path stays single-stream (RCCL's kernels are not ours to rebuild)."""
from __future__ import annotations

import torch

from .engine import EBranchformerEngine


class ForwardPipeline:
    def __init__(self, cfg: dict, device, state_dict: dict, lanes: int = 2):
        if lanes < 1:
            raise ValueError("lanes >= 1")
        self.device = torch.device(device)
        self.engines = []
        for _ in range(lanes):
            e = EBranchformerEngine(cfg, self.device)
            e.load_state_dict(state_dict)
            self.engines.append(e)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(lanes)] if lanes > 1 else [None]
        self._next = 0

    @property
    def lanes(self) -> int:
        return len(self.engines)

    def submit(self, fn):
        """fn(engine, lane) is enqueued on the next lane's stream (behind that lane's earlier work only) -> (lane, fn's return value).
        The caller synchronises (torch.cuda.synchronize(), or an event recorded on `self.streams[lane]`) before reading results on another stream."""
        lane = self._next
        self._next = (lane + 1) % len(self.engines)
        st = self.streams[lane]
        if st is None:
            return lane, fn(self.engines[lane], lane)
        with torch.cuda.stream(st):
            return lane, fn(self.engines[lane], lane)

    def reset(self):
        self._next = 0
