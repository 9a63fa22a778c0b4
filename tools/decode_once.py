"""one greedy + one beam-5 decode of the config-5 model (for profiling): python tools/decode_once.py [W]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_bench.py")).read().split("wave = torch.from_numpy")[0])
wave = torch.from_numpy(synth.waveforms(1, 1, 160000)).to(dev)
tb = FB.FbankTables(80)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    feats, frames = FB.fbank_gpu(wave, tb, pad_frames_to=100)
    st = {}
    out = generate(eng, feats, frames, num_beams=W, max_length=40, ctc_weight=float(sys.argv[3]) if len(sys.argv) > 3 else 0.3, eos_token_id=1, run_ahead=int(sys.argv[2]) if len(sys.argv) > 2 else 2, stats=st)
    torch.cuda.synchronize(); print(f"W={W}: {(time.perf_counter() - t0) * 1e3:.1f} ms end to end; host spent {st['host_loop_ms']:.1f} ms enqueuing {st['steps']} steps")
