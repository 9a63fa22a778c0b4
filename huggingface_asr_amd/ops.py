"""Thin torch-tensor wrappers over the C ABI (one per entry point in include/hfasr_hip.h).

PyTorch is used for device memory and streams only; every op here runs a hand-written HIP kernel on
`torch.cuda.current_stream()`.  All tensors must be CUDA(HIP) tensors; there is no CPU path.
"""
from __future__ import annotations

import math

import torch

from . import _lib

BF16 = torch.bfloat16


def _p(t):
    return 0 if t is None else t.data_ptr()


_PINNED_STREAM = None          # set by `pinned_stream()`: the training step makes ~1100 C-ABI calls, and asking torch for the current stream costs ~5 us each


def _stream():
    return torch.cuda.current_stream().cuda_stream if _PINNED_STREAM is None else _PINNED_STREAM


class pinned_stream:
    """`with ops.pinned_stream():` — resolve torch's current stream ONCE for the body (a trainer step runs on one stream from start to end) instead of once per
    kernel launch.  Code inside that switches streams with `torch.cuda.stream(...)` must not rely on `_stream()` following it; nesting keeps the outer value."""

    def __enter__(self):
        global _PINNED_STREAM
        self.prev = _PINNED_STREAM
        if _PINNED_STREAM is None:
            _PINNED_STREAM = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, *exc):
        global _PINNED_STREAM
        _PINNED_STREAM = self.prev
        return False


def _req(t: torch.Tensor, dtype=None):
    if not t.is_cuda:
        raise RuntimeError("huggingface_asr_amd ops need device tensors (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return t


def gemm(a, w, bias=None, out=None, *, out_dtype=BF16, act="none", resid=None, alpha=1.0, bias_per_row=False,
         col_remap=None, variant=0):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T); a/w bf16 (last dim contiguous).  `variant`: kernel selection of mi_gemm_bf16_v (tests / A-B tools; 0 = product dispatch)."""
    _req(a, BF16); _req(w, BF16)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=out_dtype)
    ct, ctp = col_remap if col_remap else (0, 0)
    rc = _lib.lib().mi_gemm_bf16_v(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _p(bias),
                                   (2 if bias_per_row else 1) if bias is not None else 0,
                                   out.data_ptr(), out.stride(0), int(out.dtype == torch.float32),
                                   _p(resid), resid.stride(0) if resid is not None else 0, float(alpha),
                                   {"none": 0, "gelu": 1, "gelu_new": 2}[act], M, N, K, ct, ctp, int(variant), _stream())
    _lib.check(rc, "mi_gemm_bf16")
    return out


def conv2d_first_gelu(x, w, bias, stride=2, pad=1, causal=False):
    """x (B,T,F) f32, w (C,K*K) f32 -> channels-last (B,T1,F1,C) bf16 = gelu(conv)."""
    B, T, F = x.shape
    Cc, KK = w.shape
    K = int(round(math.sqrt(KK)))
    T1, F1 = (T + 2 * pad - K) // stride + 1, (F + 2 * pad - K) // stride + 1
    out = torch.empty((B, T1, F1, Cc), device=x.device, dtype=BF16)
    pl = 2 * pad if causal else pad
    rc = _lib.lib().mi_conv2d_first_gelu(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), B, T, F, Cc, K,
                                         stride, pl, pl, T1, F1, _stream())
    _lib.check(rc, "mi_conv2d_first_gelu")
    return out


def conv2d_cl(x, w, bias, K=3, stride=2, pad=1, causal=False, act="gelu", variant=0):
    """x (B,T,F,Cin) bf16 channels-last, w (Cout, KH*KW*Cin) bf16 -> (B,T',F',Cout) bf16.  K / pad may be (time, freq) pairs
    (a Conv1d over time is K=(k,1), pad=(p,0) on an F=1 layout)."""
    B, T, F, Cin = x.shape
    Cout = w.shape[0]
    KH, KW = (K, K) if isinstance(K, int) else K
    pt, pf = (pad, pad) if isinstance(pad, int) else pad
    T1, F1 = (T + 2 * pt - KH) // stride + 1, (F + 2 * pf - KW) // stride + 1
    out = torch.empty((B, T1, F1, Cout), device=x.device, dtype=BF16)
    plt, plf = (2 * pt, 2 * pf) if causal else (pt, pf)
    rc = _lib.lib().mi_conv2d_cl_bf16_v(x.data_ptr(), w.data_ptr(), _p(bias), out.data_ptr(), B, T, F, Cin, Cout, KH, KW, stride,
                                        plt, plf, T1, F1, {"none": 0, "gelu": 1}[act], int(variant), _stream())
    _lib.check(rc, "mi_conv2d_cl_bf16")
    return out


def _geo_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def conv2d_first_geo(x, w, bias, K=(3, 3), stride=(2, 2), pad=(1, 1), act="gelu"):
    """Conv2d(1 -> C) of general geometry: x (B,T,F) f32, w (C, KH*KW) f32 -> channels-last (B,T1,F1,C) bf16; act "gelu" or "none" (raw pre-activation)."""
    B, T, F = x.shape
    Cc = w.shape[0]
    (KH, KW), (st, sf), (pt, pf) = K, stride, pad
    T1, F1 = _geo_out(T, KH, st, pt), _geo_out(F, KW, sf, pf)
    out = torch.empty((B, T1, F1, Cc), device=x.device, dtype=BF16)
    _lib.check(_lib.lib().mi_conv2d_first_geo(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), B, T, F, Cc, KH, KW, st, sf, pt, pf, T1, F1,
                                              {"none": 0, "gelu": 1}[act], _stream()), "mi_conv2d_first_geo")
    return out


def conv2d_first_gated_gelu(x, w, bias, gw, gbias, stride=2, pad=1):
    """GatedConv2d first layer fused (extractors.py:23-32): GELU((conv + b) * sigmoid(gate + bg)); 3x3 only."""
    B, T, F = x.shape
    Cc, KK = w.shape
    K = int(round(math.sqrt(KK)))
    T1, F1 = _geo_out(T, K, stride, pad), _geo_out(F, K, stride, pad)
    out = torch.empty((B, T1, F1, Cc), device=x.device, dtype=BF16)
    _lib.check(_lib.lib().mi_conv2d_first_gated_gelu(x.data_ptr(), w.data_ptr(), bias.data_ptr(), gw.data_ptr(), gbias.data_ptr(), out.data_ptr(), B, T, F, Cc, K,
                                                     stride, pad, pad, T1, F1, _stream()), "mi_conv2d_first_gated_gelu")
    return out


def conv2d_cl_geo(x, w, bias, K=(3, 3), stride=(2, 2), pad=(1, 1), act="gelu", gated=False):
    """conv2d_cl with (time, freq) kernel / stride / pad pairs; gated: w (2*Cout, K) / bias (2*Cout) interleaved [conv 32 ; gate 32] -> GELU(conv * sigmoid(gate))."""
    B, T, F, Cin = x.shape
    Cout = w.shape[0] // (2 if gated else 1)
    (KH, KW), (st, sf), (pt, pf) = K, stride, pad
    T1, F1 = _geo_out(T, KH, st, pt), _geo_out(F, KW, sf, pf)
    out = torch.empty((B, T1, F1, Cout), device=x.device, dtype=BF16)
    _lib.check(_lib.lib().mi_conv2d_cl_geo_bf16(x.data_ptr(), w.data_ptr(), _p(bias), out.data_ptr(), B, T, F, Cin, Cout, KH, KW, st, sf, pt, pf, T1, F1,
                                                {"none": 0, "gelu": 1}[act], int(gated), _stream()), "mi_conv2d_cl_geo_bf16")
    return out


def gated_act(z, g, B, T, Fq, C, share=1, blk=0, out=None):
    """GELU(z * sigmoid(g)) -> (B*T*Fq, C) bf16; z rows (b,t,f), g rows (b, t // share, f); blk > 0: z is g, columns interleaved [conv blk | gate blk]."""
    z2, g2 = z.reshape(-1, z.shape[-1]), g.reshape(-1, g.shape[-1])
    if out is None:
        out = torch.empty((B * T * Fq, C), device=z.device, dtype=BF16)
    _lib.check(_lib.lib().mi_gated_act_bf16(z2.data_ptr(), z2.stride(0), g2.data_ptr(), g2.stride(0), out.data_ptr(), out.stride(0), B, T, Fq, C, share, blk, _stream()),
               "mi_gated_act_bf16")
    return out


def gemm_lnfold(xb, wf, colsum, cbias, stats, npart, eps=1e-5, act="none", out=None):
    """act(LN(x) W^T + b) with the LayerNorm folded into the GEMM (csrc/gemm_args.hpp): xb = bf16(x) (M,K), wf = bf16(W diag(gamma)) (N,K), colsum = sum_k wf, cbias = W beta + b,
    stats (M,32) fp32 = per-row partial (sum, sumsq) pairs of x (npart pairs)."""
    M, K = xb.shape
    N = wf.shape[0]
    if out is None:
        out = torch.empty((M, N), device=xb.device, dtype=BF16)
    _lib.check(_lib.lib().mi_gemm_lnfold_bf16(xb.data_ptr(), xb.stride(0), wf.data_ptr(), wf.stride(0), colsum.data_ptr(), cbias.data_ptr(), stats.data_ptr(), int(npart), float(eps),
                                              out.data_ptr(), out.stride(0), {"none": 0, "gelu": 1}[act], M, N, K, _stream()), "mi_gemm_lnfold_bf16")
    return out


def gemm_resid_stats(a, w, bias, resid, alpha=1.0, wide=False, variant=None):
    """-> (C fp32 = resid + alpha (a W^T + b), C2 = bf16(C), stats (M,32) fp32 with one (sum, sumsq) pair per 32 columns of C — per 64 columns with `wide`,
    the 256 x 256 tile of the throughput mode (mi_ebf_config.wide_tiles))"""
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty((M, N), device=a.device, dtype=torch.float32)
    c2 = torch.empty((M, N), device=a.device, dtype=BF16)
    st = torch.zeros((M, 32), device=a.device, dtype=torch.float32)
    _lib.check(_lib.lib().mi_gemm_resid_stats_f32_v(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _p(bias), c.data_ptr(), c.stride(0), _p(resid), resid.stride(0) if resid is not None else 0,
                                                    float(alpha), c2.data_ptr(), c2.stride(0), st.data_ptr(), M, N, K, (40 if wide else 0) if variant is None else int(variant), _stream()), "mi_gemm_resid_stats_f32_v")
    return c, c2, st


def layernorm_fold(x, ln=None, eps=1e-5, lengths=None, T=1):
    """y = [LN](mask(x)) -> (y fp32, bf16(y), stats (M,32) with (sum, sumsq) of y in pair 0)"""
    M, d = x.shape
    g, b = ln if ln else (None, None)
    y = torch.empty_like(x)
    yb = torch.empty((M, d), device=x.device, dtype=BF16)
    st = torch.zeros((M, 32), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().mi_layernorm_fold(x.data_ptr(), x.stride(0), _p(lengths), T, _p(g), _p(b), float(eps), y.data_ptr(), y.stride(0), yb.data_ptr(), yb.stride(0), st.data_ptr(),
                                            M, d, _stream()), "mi_layernorm_fold")
    return y, yb, st


def layernorm_chain(x, *, lengths=None, T=1, ln1=None, eps1=1e-5, store_y=None, lna=None, eps2=1e-5, outa=None,
                    outa32=None, lnb=None, outb=None):
    """see csrc/norm.hip; x (M,d) f32; ln* = (gamma, beta) f32."""
    M, d = x.shape
    g1, b1 = ln1 if ln1 else (None, None)
    ga, ba = lna if lna else (None, None)
    gb, bb = lnb if lnb else (None, None)
    rc = _lib.lib().mi_layernorm_chain(x.data_ptr(), x.stride(0), _p(lengths), T, _p(g1), _p(b1), eps1,
                                       _p(store_y), store_y.stride(0) if store_y is not None else 0,
                                       _p(ga), _p(ba), eps2, _p(outa), outa.stride(0) if outa is not None else 0,
                                       _p(outa32), outa32.stride(0) if outa32 is not None else 0,
                                       _p(gb), _p(bb), _p(outb), outb.stride(0) if outb is not None else 0, M, d, _stream())
    _lib.check(rc, "mi_layernorm_chain")


def rotary(x, cos, sin, T, H):
    M, d = x.shape
    out = torch.empty_like(x)
    rc = _lib.lib().mi_rotary_bf16(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), cos.data_ptr(), sin.data_ptr(),
                                   M, T, H, d // H, _stream())
    _lib.check(rc, "mi_rotary_bf16")
    return out


def attention(q, k, vt, Tp, B, T, H, *, pos=None, bias_u=None, bias_v=None, lengths=None, causal=False):
    """q,k (B*T, >=d) bf16 views; vt (d, B*Tp) bf16; pos (2T-1, d) bf16 or None -> ctx (B*T, d) bf16."""
    d = vt.shape[0]
    hd = d // H
    out = torch.empty((B * T, d), device=q.device, dtype=BF16)
    rc = _lib.lib().mi_attention_bf16(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), vt.data_ptr(), vt.stride(0), Tp,
                                      _p(pos), pos.stride(0) if pos is not None else 0, _p(bias_u), _p(bias_v), _p(lengths),
                                      out.data_ptr(), out.stride(0), B, T, H, hd, 1.0 / math.sqrt(hd), int(causal), _stream())
    _lib.check(rc, "mi_attention_bf16")
    return out


def attention_qkv(qkv, B, T, H, *, pos=None, bias_u=None, bias_v=None, lengths=None, causal=False, lse=None, drop=None, variant=0):
    """LDS-staged attention on a fused (B*T, 3d) bf16 projection [Q|K|V]; head size 64 or 128.
    lse: (B, H, T) fp32 that receives the rows' log-sum-exp (training forward; the backward's `ops_train.attn_bwd_probs` reads it).
    drop = (p, seed, stream_id): attention-probability dropout (with lse only)."""
    d = qkv.shape[1] // 3
    hd = d // H
    out = torch.empty((B * T, d), device=qkv.device, dtype=BF16)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    if lse is not None:
        dp, dseed, dsid = drop if drop is not None else (0.0, 0, 0)
        rc = _lib.lib().mi_attention_qkv_lse_bf16(q.data_ptr(), qkv.stride(0), k.data_ptr(), qkv.stride(0), v.data_ptr(), qkv.stride(0),
                                                  _p(pos), pos.stride(0) if pos is not None else 0, _p(bias_u), _p(bias_v), _p(lengths),
                                                  out.data_ptr(), out.stride(0), lse.data_ptr(), B, T, H, hd, 1.0 / math.sqrt(hd), int(causal),
                                                  float(dp), int(dseed) & 0xFFFFFFFF, int(dsid) & 0xFFFFFFFF, _stream())
        _lib.check(rc, "mi_attention_qkv_lse_bf16")
        return out
    if variant:      # measurement only (tools/attn_ab.py): the four-wave kernel of rounds 1-3
        rc = _lib.lib().mi_attention_qkv_bf16_v(q.data_ptr(), qkv.stride(0), k.data_ptr(), qkv.stride(0), v.data_ptr(), qkv.stride(0),
                                                _p(pos), pos.stride(0) if pos is not None else 0, _p(bias_u), _p(bias_v), _p(lengths),
                                                out.data_ptr(), out.stride(0), B, T, 0, 0, H, hd, 1.0 / math.sqrt(hd), int(causal), int(variant), _stream())
        _lib.check(rc, "mi_attention_qkv_bf16_v")
        return out
    rc = _lib.lib().mi_attention_qkv_bf16(q.data_ptr(), qkv.stride(0), k.data_ptr(), qkv.stride(0), v.data_ptr(), qkv.stride(0),
                                          _p(pos), pos.stride(0) if pos is not None else 0, _p(bias_u), _p(bias_v), _p(lengths),
                                          out.data_ptr(), out.stride(0), B, T, 0, 0, H, hd, 1.0 / math.sqrt(hd), int(causal), _stream())
    _lib.check(rc, "mi_attention_qkv_bf16")
    return out


def attention_general(q, k, v, B, Tq, Tk, H, *, lengths=None, causal=False, out=None, kv_bstride=0):
    """q (B*Tq, .) / k, v (B*Tk, .) bf16 row views with head h at columns [h*hd, (h+1)*hd) -> context (B*Tq, d) bf16.
    Cross-attention (Tk = encoder frames, `lengths` = valid keys) and KV-cache steps (causal offset Tk - Tq)."""
    d = q.shape[1]
    hd = d // H
    if out is None:
        out = torch.empty((B * Tq, d), device=q.device, dtype=BF16)
    rc = _lib.lib().mi_attention_qkv_bf16(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                          0, 0, 0, 0, _p(lengths), out.data_ptr(), out.stride(0), B, Tq, Tk, kv_bstride, H, hd,
                                          1.0 / math.sqrt(hd), int(causal), _stream())
    _lib.check(rc, "mi_attention_qkv_bf16")
    return out


def row_stats(x, eps=1e-5):
    M, d = x.shape
    st = torch.empty((M, 2), device=x.device, dtype=torch.float32)
    rc = _lib.lib().mi_row_stats_bf16(x.data_ptr(), x.stride(0), d, eps, st.data_ptr(), M, _stream())
    _lib.check(rc, "mi_row_stats_bf16")
    return st


def csgu(u, gamma, beta, w, bias, B, T, *, pad_left=None, dilation=1, act=0, eps=1e-5, stats=None):
    """u (B*T, 2C) bf16 = [x_r | x_g] -> x_r * act(dwconv(LN(x_g)) + b)  (B*T, C) bf16.  stats: row_stats(u[:, C:], eps) when the caller already holds it."""
    M, C2 = u.shape
    Cc = C2 // 2
    K = w.shape[-1]
    st = row_stats(u[:, Cc:], eps) if stats is None else stats
    out = torch.empty((M, Cc), device=u.device, dtype=BF16)
    rc = _lib.lib().mi_csgu_bf16(u.data_ptr(), u.stride(0), st.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(),
                                 _p(bias), out.data_ptr(), out.stride(0), B, T, Cc, K,
                                 (K - 1) // 2 if pad_left is None else pad_left, dilation, act, _stream())
    _lib.check(rc, "mi_csgu_bf16")
    return out


def csgu_conv(u, stats, gamma, beta, w, bias, B, T, *, pad_left=None, dilation=1):
    """the CSGU conv alone (split form: csgu_use_linear_after_conv / non-identity activation on the training path): dwconv(LN(x_g)) + b  (B*T, C) bf16;
    stats = row_stats(u[:, C:])"""
    M, C2 = u.shape
    Cc = C2 // 2
    K = w.shape[-1]
    out = torch.empty((M, Cc), device=u.device, dtype=BF16)
    rc = _lib.lib().mi_csgu_conv_bf16(u.data_ptr(), u.stride(0), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(), _p(bias),
                                      out.data_ptr(), out.stride(0), B, T, Cc, K, (K - 1) // 2 if pad_left is None else int(pad_left), int(dilation), _stream())
    _lib.check(rc, "mi_csgu_conv_bf16")
    return out


def gate_act_mul(r, g, act=0):
    """r * act(g) on (M, C) bf16 rows (r may be a column slice); act 0 identity, 1 gelu, 2 relu, 3 silu"""
    M, Cc = g.shape
    out = torch.empty((M, Cc), device=g.device, dtype=BF16)
    rc = _lib.lib().mi_gate_act_mul_bf16(r.data_ptr(), r.stride(0), g.data_ptr(), g.stride(0), out.data_ptr(), out.stride(0), M, Cc, int(act), _stream())
    _lib.check(rc, "mi_gate_act_mul_bf16")
    return out


def dwconv_residual(m, w, bias, B, T, pad_left=None):
    M, Cc = m.shape
    K = w.shape[-1]
    out = torch.empty_like(m)
    rc = _lib.lib().mi_dwconv_residual_bf16(m.data_ptr(), m.stride(0), w.data_ptr(), _p(bias), out.data_ptr(), out.stride(0),
                                            B, T, Cc, K, (K - 1) // 2 if pad_left is None else int(pad_left), _stream())
    _lib.check(rc, "mi_dwconv_residual_bf16")
    return out


def gemm_lse(a, w, bias, out):
    """the CTC head in one pass over its logits: out (M, ld >= N) fp32 [:, :N] = a W^T + b, N = w.shape[0], and -> lse (M) fp32 = the rows' log-sum-exp over those N
    columns, out of the GEMM's epilogue (mi_gemm_lse_f32); shapes outside that kernel: the GEMM followed by row_lse."""
    _req(a, BF16); _req(w, BF16)
    M, K = a.shape
    N = w.shape[0]
    L = _lib.lib()
    lse = torch.empty((M,), device=a.device, dtype=torch.float32)
    ws = torch.empty((int(L.mi_gemm_lse_workspace_floats(M, N)),), device=a.device, dtype=torch.float32)
    rc = L.mi_gemm_lse_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _p(bias), out.data_ptr(), out.stride(0), lse.data_ptr(), ws.data_ptr(), M, N, K, _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        gemm(a, w, bias, out=out)
        return row_lse(out[:, :N])
    _lib.check(rc, "mi_gemm_lse_f32")
    return lse


def row_lse(x):
    M, V = x.shape
    out = torch.empty((M,), device=x.device, dtype=torch.float32)
    rc = _lib.lib().mi_row_lse(x.data_ptr(), x.stride(0), 0 if x.dtype == torch.float32 else 1, V, out.data_ptr(), M, _stream())
    _lib.check(rc, "mi_row_lse")
    return out


def ctc_loss(logits, labels, in_len, *, reduction="mean", zero_infinity=False, lse=None):
    """logits (B,T,V+1) f32|bf16 (blank = last class), labels (B,U) int64 (<0 = padding), in_len (B) int32.
    Returns (loss scalar tensor | per-utterance nll for reduction='none', nll (B), tgt_len (B))."""
    B, T, V1 = logits.shape
    labels = labels.contiguous()
    U = labels.shape[1]
    if lse is None:
        lse = row_lse(logits.reshape(B * T, V1))
    nll = torch.empty((B,), device=logits.device, dtype=torch.float32)
    tl = torch.empty((B,), device=logits.device, dtype=torch.int32)
    loss = torch.empty((1,), device=logits.device, dtype=torch.float32)
    rc = _lib.lib().mi_ctc_loss_fwd(logits.data_ptr(), logits.stride(0), logits.stride(1), 0 if logits.dtype == torch.float32 else 1,
                                    lse.data_ptr(), T, labels.data_ptr(), U, in_len.data_ptr(), V1 - 1, B,
                                    1 if reduction == "mean" else 0, int(zero_infinity), nll.data_ptr(), tl.data_ptr(),
                                    loss.data_ptr(), _stream())
    _lib.check(rc, "mi_ctc_loss_fwd")
    if reduction == "none":
        out = torch.where(torch.isinf(nll), torch.zeros_like(nll), nll) if zero_infinity else nll
        return out, nll, tl
    return loss[0], nll, tl


def embed_tokens(ids, wte, pos, *, scale=1.0, pos_offset=0, U=None):
    """ids (B,U) int64 -> (B*U, d) fp32 = wte[ids]*scale + pos[pos_offset + u]."""
    ids = ids.contiguous()
    M = ids.numel()
    U = ids.shape[-1] if U is None else U
    V, d = wte.shape
    out = torch.empty((M, d), device=ids.device, dtype=torch.float32)
    rc = _lib.lib().mi_embed_tokens(ids.data_ptr(), wte.data_ptr(), float(scale), pos.data_ptr(), pos_offset, U, d, M, V, out.data_ptr(), _stream())
    _lib.check(rc, "mi_embed_tokens")
    return out


def ce_label_smoothing(logits, labels, *, shift=1, eps=0.0):
    """mean over valid targets of the label-smoothed CE of logits[b,u] vs labels[b,u+shift] (ignore < 0). logits (B,U,V) fp32."""
    B, U, V = logits.shape
    labels = labels.contiguous()
    acc = torch.zeros((2,), device=logits.device, dtype=torch.float32)
    rows = torch.empty((B * (U - shift),), device=logits.device, dtype=torch.float32)       # per-row losses; summed by one block in a fixed order
    rc = _lib.lib().mi_ce_label_smoothing(logits.data_ptr(), logits.stride(1), labels.data_ptr(), B, U, shift, V, float(eps), acc.data_ptr(), rows.data_ptr(), _stream())
    _lib.check(rc, "mi_ce_label_smoothing")
    return acc[0] / acc[1]


def cast_bf16(x):
    """(M,d) fp32 -> bf16 (round to nearest even) on the HIP path."""
    M, d = x.shape
    out = torch.empty((M, d), device=x.device, dtype=BF16)
    rc = _lib.lib().mi_cast_f32_bf16(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), M, d, _stream())
    _lib.check(rc, "mi_cast_f32_bf16")
    return out
