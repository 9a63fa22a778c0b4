"""Host logic of the throughput mode that needs no GPU: the hardware-queue reservation bench.py makes before its first HIP call."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
from huggingface_asr_amd.pipeline import reserve_hw_queues
print(reserve_hw_queues(int(sys.argv[1])), os.environ.get("GPU_MAX_HW_QUEUES"))
"""


def _run(lanes, env_value=None):
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    if env_value is not None:
        env["GPU_MAX_HW_QUEUES"] = env_value
    out = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT), str(lanes)], env=env, capture_output=True, text=True, check=True).stdout.split()
    return int(out[0]), out[1]


def test_hw_queue_reservation_follows_the_lane_count_and_leaves_an_exported_value_alone():
    assert _run(2) == (4, "None")                 # up to three lanes fit the runtime's default of four queues: nothing is exported
    assert _run(3) == (4, "None")
    assert _run(4) == (8, "8")                    # four lanes + the default stream: eight queues, exported before the HIP runtime reads it
    assert _run(6) == (8, "8")
    assert _run(4, "2") == (2, "2")               # the user's value wins
    assert _run(2, "16") == (16, "16")
