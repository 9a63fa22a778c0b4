"""cProfile of the host side of generate() (config 5, W = 5, 40 tokens): where the Python loop's time per token goes"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_bench.py")).read().split("wave = torch.from_numpy")[0])
wave = torch.from_numpy(synth.waveforms(1, 1, 160000)).to(dev)
tb = FB.FbankTables(80)
feats, frames = FB.fbank_gpu(wave, tb, pad_frames_to=100)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(3):
    generate(eng, feats, frames, num_beams=W, max_length=40, ctc_weight=0.3, eos_token_id=1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    generate(eng, feats, frames, num_beams=W, max_length=40, ctc_weight=0.3, eos_token_id=1)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(32)
