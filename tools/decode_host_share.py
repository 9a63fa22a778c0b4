import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_bench.py")).read().split("wave = torch.from_numpy")[0])
wave = torch.from_numpy(synth.waveforms(1, 1, 160000)).to(dev)
tb = FB.FbankTables(80)
for W in (1, 5):
    for rep in range(4):
        st = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        feats, frames = FB.fbank_gpu(wave, tb, pad_frames_to=100)
        t1 = time.perf_counter()
        out = generate(eng, feats, frames, num_beams=W, max_length=40, ctc_weight=0.3, eos_token_id=1, stats=st)
        t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"W={W}: e2e {(t3-t0)*1e3:.1f} ms; fbank enqueue {(t1-t0)*1e3:.2f}; generate() returned after {(t2-t1)*1e3:.1f} ms; host token loop {st.get('host_loop_ms'):.1f} ms for {st.get('steps')} steps; tail sync {(t3-t2)*1e3:.2f}")
    # encoder alone
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.encode(feats, frames)
    torch.cuda.synchronize(); print(f"   encode alone {(time.perf_counter()-t0)/5*1e3:.2f} ms")
