"""does the CTC prefix scorer on its own stream slow the decoder step down?  event-timed duration of every mi_gpt2_step inside generate(), with and without CTC scoring
usage: python tools/decode_step_timing.py [W]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_bench.py")).read().split("wave = torch.from_numpy")[0])
wave = torch.from_numpy(synth.waveforms(1, 1, 160000)).to(dev)
tb = FB.FbankTables(80)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1
orig = eng.dec.step
evs = []
def timed_step(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(*a, **k); e1.record()
    evs.append((e0, e1))
    return r
eng.dec.step = timed_step
for w in (0.3, 0.0, 0.3, 0.0):
    for rep in range(4):
        evs.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        feats, frames = FB.fbank_gpu(wave, tb, pad_frames_to=100)
        out = generate(eng, feats, frames, num_beams=W, max_length=40, ctc_weight=w, eos_token_id=1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    d = [a.elapsed_time(b) for a, b in evs]
    print(f"W={W} ctc_weight={w}: {dt:.1f} ms end to end; decoder step {sum(d) / len(d) * 1e3:.0f} us on average over {len(d)} tokens (first {d[0] * 1e3:.0f}, last {d[-1] * 1e3:.0f})")
