"""where does a decode token go?  (bs=1, beam 5, config-5 shapes)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_bench.py")).read().split("wave = torch.from_numpy")[0])
from huggingface_asr_amd import ops
from huggingface_asr_amd.decoding import CTCRescorerLogitsProcessor
wave = torch.from_numpy(synth.waveforms(1, 1, 160000)).to(dev)
tb = FB.FbankTables(80)
def T(f, n=20):
    f(); f()                                            # warm-up (first use of a kernel loads its code object)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
W = 5
t_enc, (feats, frames) = T(lambda: FB.fbank_gpu(wave, tb, pad_frames_to=100), 5)
t_e2, enc = T(lambda: eng.encode(feats, frames), 5)
enc_out, enc_bf, T2, key_len = enc
d = enc_bf.shape[1]
enc_rep = enc_bf.view(1, T2, d).repeat_interleave(W, 0).reshape(W * T2, d)
key_rep = key_len.repeat_interleave(W)
t_kv, kvs = T(lambda: eng.dec.cross_kv(enc_rep), 5)
cache = eng.dec.init_cache(W, 64)
ids = torch.full((W, 1), 2, dtype=torch.long, device=dev)
def step():
    cache["past"] = 10
    return eng.dec.step(ids, cache, kvs, T2, key_rep)
t_step, logits = T(step)
def step_py():
    cache["past"] = 10
    return eng.dec.step_py(ids, cache, kvs, T2, key_rep)
t_steppy, _ = T(step_py)
t_lse, scores = T(lambda: logits - ops.row_lse(logits.contiguous())[:, None])
lens = enc_out["outer_len"].clamp(max=T2)
proc = CTCRescorerLogitsProcessor(enc_out["logits"], lens, 5000, 1, 0, 0.3, W, -1, False, 1.0)
idsl = torch.full((W, 11), 7, dtype=torch.long, device=dev); idsl[:, 0] = 2
proc(idsl[:, :1], scores.clone())
t_proc, sc2 = T(lambda: proc(idsl, scores.clone()))
t_topk, _ = T(lambda: [t.cpu() for t in (sc2.view(1, -1)).topk(2 * W, dim=1)])
bi = torch.arange(W, device=dev)
t_reo, _ = T(lambda: eng.dec.reorder_cache(cache, bi))
print(f"fbank {t_enc:.2f} ms, encoder {t_e2:.2f}, cross_kv {t_kv:.2f} | per token: step(C) {t_step:.3f}, step(py) {t_steppy:.3f}, log-softmax {t_lse:.3f}, "
      f"ctc proc {t_proc:.3f}, topk+D2H {t_topk:.3f}, reorder {t_reo:.3f}")
