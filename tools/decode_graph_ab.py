"""decoder token step (config-5 shapes, bs = 1): eager C call against a hipGraph replay of the same launches, W = 1 and W = 5"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_bench.py")).read().split("wave = torch.from_numpy")[0])
wave = torch.from_numpy(synth.waveforms(1, 1, 160000)).to(dev)
tb = FB.FbankTables(80)
feats, frames = FB.fbank_gpu(wave, tb, pad_frames_to=100)
enc_out, enc_bf, T2, key_len = eng.encode(feats, frames)
d = enc_bf.shape[1]


def T(f, n=200):
    f(); f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for W in (1, 5):
    enc_rep = enc_bf.view(1, T2, d).repeat_interleave(W, 0).reshape(W * T2, d)
    key_rep = key_len.repeat_interleave(W)
    kvs = eng.dec.cross_kv(enc_rep)
    cache = eng.dec.init_cache(W, 64)
    ids = torch.full((W, 1), 2, dtype=torch.long, device=dev)

    def step():
        cache["past"] = 10
        return eng.dec.step(ids, cache, kvs, T2, key_rep)
    t_eager = T(step)
    ref = step().clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = step()
    t_graph = T(g.replay)
    g.replay(); torch.cuda.synchronize()
    print(f"W={W}: eager {t_eager:.3f} ms, graph replay {t_graph:.3f} ms, same bits {bool(torch.equal(out, ref))}")
