"""torch-tensor wrappers over the training-step entry points of the C ABI (include/hfasr_hip.h, "training step" block).

Same rules as ops.py: device tensors only, kernels run on torch.cuda.current_stream(), no CPU path.
Parameter-gradient outputs (dgamma, dbeta, dw, db, colsum targets ...) ACCUMULATE into the tensors passed in.
"""
from __future__ import annotations

import math

import torch

from . import _lib
from .ops import BF16, _p, _req, _stream, gemm

F32 = torch.float32


def _L():
    return _lib.lib()


_DW_WS = {}


def _dw_ws(device, nfloats):
    """scratch for per-block partial sums (LayerNorm / depthwise-conv parameter gradients); stream-ordered reuse"""
    ws = _DW_WS.get(device)
    if ws is None or ws.numel() < nfloats:
        ws = _DW_WS[device] = torch.empty(max(nfloats, 512 * 2 * 2048), device=device, dtype=F32)
    return ws.data_ptr()


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def transpose(x, Mp=None, out=None):
    """x (M,N) bf16 (row stride free) -> (N, Mp) bf16, columns M..Mp zero."""
    _req(x, BF16)
    M, N = x.shape
    Mp = pad64(M) if Mp is None else Mp
    if out is None:
        out = torch.empty((N, Mp), device=x.device, dtype=BF16)
    _lib.check(_L().mi_transpose_bf16(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), M, N, Mp, _stream()), "mi_transpose_bf16")
    return out


def colsum_(out, x):
    """out (N) f32 += column sums of x (M,N) f32|bf16."""
    M, N = x.shape
    ws = _dw_ws(x.device, int(_L().mi_colsum_workspace_floats(M, N)))      # per-chunk partial sums, added in chunk order (no atomics)
    _lib.check(_L().mi_colsum(x.data_ptr(), x.stride(0), 0 if x.dtype == F32 else 1, M, N, out.data_ptr(), ws, _stream()), "mi_colsum")


def colsum2_acc_(out_a, out_b, a, b):
    """out_a (N) += column sums of a (M, N) f32, out_b (N) += column sums of b (same shape): rows added in a fixed order, one launch"""
    M, N = a.shape
    assert b.shape == a.shape and a.stride(0) == b.stride(0)
    _lib.check(_L().mi_colsum2_acc_f32(a.data_ptr(), b.data_ptr(), a.stride(0), M, N, out_a.data_ptr(), out_b.data_ptr(), _stream()), "mi_colsum2_acc_f32")


def colsum_cast(x):
    """-> (N) bf16 = column sums of x (M, N) f32, rows added in order (M small)"""
    M, N = x.shape
    out = torch.empty((N,), device=x.device, dtype=BF16)
    _lib.check(_L().mi_colsum_cast_bf16(x.data_ptr(), x.stride(0), M, N, out.data_ptr(), _stream()), "mi_colsum_cast_bf16")
    return out


KIND = {"gelu": 1, "gelu_new": 2}


def act_fwd(pre, kind="gelu", out=None, drop=None):
    """out = act(pre); with drop = (p, seed, stream_id): out = dropout(act(pre)) in the same pass (bit-identical to act_fwd + dropout_)."""
    M, N = pre.shape
    if out is None:
        out = torch.empty((M, N), device=pre.device, dtype=BF16)
    if drop is not None:
        p, seed, sid = drop
        _lib.check(_L().mi_act_dropout_fwd_bf16(pre.data_ptr(), pre.stride(0), out.data_ptr(), out.stride(0), M, N, KIND[kind], float(p),
                                                int(seed) & 0xFFFFFFFF, int(sid) & 0xFFFFFFFF, _stream()), "mi_act_dropout_fwd_bf16")
        return out
    _lib.check(_L().mi_act_fwd_bf16(pre.data_ptr(), pre.stride(0), out.data_ptr(), out.stride(0), M, N, KIND[kind], _stream()), "mi_act_fwd_bf16")
    return out


def gemm_act_fwd(a, w, bias, kind="gelu", drop=None):
    """(pre, h) = (a W^T + b, dropout(act(pre))) — the FFN-in GEMM with the activation pass in its epilogue where the 256 x 256 kernel takes the shape
    (N % 256 == 0, K % 64 == 0, K >= 128), else the GEMM followed by act_fwd; bit-identical either way."""
    M, K = a.shape
    N = w.shape[0]
    pre = torch.empty((M, N), device=a.device, dtype=BF16)
    h = torch.empty((M, N), device=a.device, dtype=BF16)
    p, seed, sid = drop if drop is not None else (0.0, 0, 0)
    rc = _L().mi_gemm_act_fwd_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _p(bias), pre.data_ptr(), pre.stride(0), h.data_ptr(), h.stride(0),
                                   KIND[kind], float(p), int(seed) & 0xFFFFFFFF, int(sid) & 0xFFFFFFFF, M, N, K, _stream())
    if rc == _lib.ERR_UNSUPPORTED:   # shape outside the fused kernel: the two launches
        gemm(a, w, bias, out=pre)
        return pre, act_fwd(pre, kind, out=h, drop=drop)
    _lib.check(rc, "mi_gemm_act_fwd_bf16")
    return pre, h


def gemm_act_bwd(dy, wT, pre, kind="gelu", drop=None):
    """dropout(dy wT^T) * act'(pre): the dX GEMM of the FFN's output linear with the activation backward in its epilogue (same shape rule / fallback as gemm_act_fwd).
    wT (N, K) = rows of the transposed weight copy, pre (M, N) bf16."""
    M, K = dy.shape
    N = pre.shape[1]
    out = torch.empty((M, N), device=dy.device, dtype=BF16)
    p, seed, sid = drop if drop is not None else (0.0, 0, 0)
    rc = _L().mi_gemm_act_bwd_bf16(dy.data_ptr(), dy.stride(0), wT.data_ptr(), wT.stride(0), pre.data_ptr(), pre.stride(0), out.data_ptr(), out.stride(0),
                                   KIND[kind], float(p), int(seed) & 0xFFFFFFFF, int(sid) & 0xFFFFFFFF, M, N, K, _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return act_bwd(gemm(dy, wT), pre, kind, out=out, drop=drop)
    _lib.check(rc, "mi_gemm_act_bwd_bf16")
    return out


def act_bwd(dy, pre, kind="gelu", out=None, drop=None):
    """out = dy * act'(pre); with drop = (p, seed, stream_id): out = dropout(dy) * act'(pre) (the backward of act_fwd(..., drop=...))."""
    M, N = pre.shape
    if out is None:
        out = torch.empty((M, N), device=pre.device, dtype=BF16)
    if drop is not None:
        p, seed, sid = drop
        _lib.check(_L().mi_act_dropout_bwd_bf16(dy.data_ptr(), dy.stride(0), pre.data_ptr(), pre.stride(0), out.data_ptr(), out.stride(0), M, N,
                                                KIND[kind], float(p), int(seed) & 0xFFFFFFFF, int(sid) & 0xFFFFFFFF, _stream()), "mi_act_dropout_bwd_bf16")
        return out
    _lib.check(_L().mi_act_bwd_bf16(dy.data_ptr(), dy.stride(0), pre.data_ptr(), pre.stride(0), out.data_ptr(), out.stride(0), M, N,
                                    KIND[kind], _stream()), "mi_act_bwd_bf16")
    return out


class LnReduceBatch:
    """Deferred (dgamma | dbeta) reductions of LayerNorm backward passes: `layernorm_bwd(..., defer=batch)` leaves its per-block partial rows in this object's arena;
    `flush()` reduces up to 24 entries in ONE launch (mi_ln_partial_reduce_many).  A 2-MB reduce is all launch latency: the training step has ~100 of them."""
    SLOTS = 24                      # a pair of base-size layers defers 18 reductions: one launch per weight-gradient flush

    MAX_PER = 2 * 512 * 2 * 2048                            # 16 MiB: the largest slot (a depthwise conv's partials, B x C x 32 floats, of up to B * C = 128 Ki)

    def __init__(self, device, floats_per_slot=512 * 2 * 64):
        self.device, self.per = device, floats_per_slot     # slots grow to the largest partial actually asked for (24 x 16 MiB whatever the model was 384 MiB per trainer)
        self.arena = None
        self.items = []
        self.keep = []                                      # caller-owned partial buffers of pending entries
        self.cursor = 0                                     # slots are handed out round-robin: the pending entries always sit in the len(items) most recent ones

    def reserve(self, need):
        """slots of at least `need` floats (<= MAX_PER) from here on.  A larger need than any before re-allocates the arena: pending entries are reduced first."""
        if need > self.MAX_PER:
            raise ValueError(f"LnReduceBatch: {need} floats asked for, a slot holds at most {self.MAX_PER}")
        if need > self.per or self.arena is None:
            self.flush()
            self.per = max(self.per, -(-int(need) // 1024) * 1024)
            self.arena = torch.empty(self.SLOTS * self.per, device=self.device, dtype=F32)
            self.cursor = 0

    def slot(self, need):
        self.reserve(need)
        if len(self.items) == self.SLOTS:                   # every slot holds partial rows that are not reduced yet
            self.flush()
        i = self.cursor
        self.cursor = (self.cursor + 1) % self.SLOTS
        return self.arena[i * self.per:(i + 1) * self.per]

    def add(self, partial, nblk, d, dgamma, dbeta):
        if any(it.dgamma == dgamma.data_ptr() for it in self.items):      # two LayerNorm passes into the same target: their `+=` must not run in one launch
            self.flush()
        self.items.append(_lib.LnRedDesc(partial=partial.data_ptr(), nblk=int(nblk), d=int(d), dgamma=dgamma.data_ptr(), dbeta=dbeta.data_ptr(), kind=0))

    def add_dw(self, partial, rows, C, K, dw, db):
        """the tap-gradient partials a depthwise-conv backward left in `partial` (rows x C x 32; csgu_bwd / dwconv_residual_bwd with defer=): reduced by the same launch"""
        if any(it.dgamma == dw.data_ptr() for it in self.items):
            self.flush()
        self.items.append(_lib.LnRedDesc(partial=partial.data_ptr(), nblk=int(rows), d=int(C), dgamma=dw.data_ptr(), dbeta=db.data_ptr() if db is not None else None, kind=int(K)))

    def add_rows2(self, su, dgu, dgv):
        """su = the [:, :d] view of a (rows, 2d) fp32 buffer of [u | v] rows (attn_bwd_probs): dgu += column sums of the u half, dgv += those of the v half, with the next
        flush.  The buffer is kept alive until then (the caching allocator would otherwise hand it to a later kernel of the same stream before the reduction has read it)."""
        rows, d = su.shape
        assert su.stride(0) == 2 * d and su.dtype == F32
        if len(self.items) == self.SLOTS or any(it.dgamma == dgu.data_ptr() for it in self.items):
            self.flush()
        self.keep.append(su)
        self.items.append(_lib.LnRedDesc(partial=su.data_ptr(), nblk=int(rows), d=int(d), dgamma=dgu.data_ptr(), dbeta=dgv.data_ptr(), kind=0))

    def flush(self):
        if self.items:
            arr = (_lib.LnRedDesc * len(self.items))(*self.items)
            _lib.check(_L().mi_ln_partial_reduce_many(arr, len(self.items), _stream()), "mi_ln_partial_reduce_many")
        self.items, self.keep = [], []


def layernorm_bwd(x, gamma, dy, dx, *, accumulate, dgamma=None, dbeta=None, eps=1e-5, defer=None, cast=None):
    """dx (+)= dLN(x)/dx · dy ; dgamma/dbeta += .  x f32|bf16, dy f32|bf16, dx f32|bf16 (all (M,d) row views).
    defer: an LnReduceBatch — the dgamma / dbeta reduction is left to its next `flush()` (the trainer flushes once per layer).
    cast = (alpha, drop): also returns bf16 (M,d) = alpha * dropout(dx) of the finished rows, written by the same pass (drop = (p, seed, stream_id) | None) —
    what `dropout_(dx, ..., out=bf16, alpha=)` / `add_cast(dx, alpha=)` would make next; -> (dx, cast tensor)."""
    M, d = x.shape
    if cast is not None:
        alpha, drop = cast
        pdrop, seed, sid = drop if drop is not None else (0.0, 0, 0)
        if defer is not None and dgamma is not None:
            import ctypes as C
            part = defer.slot(512 * 2 * d)
            nblk = C.c_int(0)
            out = torch.empty((M, d), device=x.device, dtype=BF16)
            _lib.check(_L().mi_layernorm_bwd_partial_cast(x.data_ptr(), x.stride(0), int(x.dtype == BF16), gamma.data_ptr(), float(eps),
                                                          dy.data_ptr(), dy.stride(0), int(dy.dtype == F32), dx.data_ptr(), dx.stride(0), int(dx.dtype == BF16),
                                                          int(accumulate), part.data_ptr(), C.byref(nblk), out.data_ptr(), out.stride(0), float(alpha), float(pdrop),
                                                          int(seed) & 0xFFFFFFFF, int(sid) & 0xFFFFFFFF, M, d, _stream()), "mi_layernorm_bwd_partial_cast")
            defer.add(part, nblk.value, d, dgamma, dbeta)
            return dx, out
        layernorm_bwd(x, gamma, dy, dx, accumulate=accumulate, dgamma=dgamma, dbeta=dbeta, eps=eps, defer=defer)       # frozen affine pair / immediate form: two passes
        return dx, (dropout_(dx, pdrop, seed, sid, out=torch.empty((M, d), device=x.device, dtype=BF16), alpha=alpha) if pdrop > 0 else add_cast(dx, alpha=alpha))
    if defer is not None and dgamma is not None:
        import ctypes as C
        part = defer.slot(512 * 2 * d)
        nblk = C.c_int(0)
        _lib.check(_L().mi_layernorm_bwd_partial(x.data_ptr(), x.stride(0), int(x.dtype == BF16), gamma.data_ptr(), float(eps),
                                                 dy.data_ptr(), dy.stride(0), int(dy.dtype == F32), dx.data_ptr(), dx.stride(0), int(dx.dtype == BF16),
                                                 int(accumulate), part.data_ptr(), C.byref(nblk), M, d, _stream()), "mi_layernorm_bwd_partial")
        defer.add(part, nblk.value, d, dgamma, dbeta)
        return dx
    ws = _dw_ws(x.device, 512 * 2 * 2048) if dgamma is not None else 0
    _lib.check(_L().mi_layernorm_bwd(x.data_ptr(), x.stride(0), int(x.dtype == BF16), gamma.data_ptr(), float(eps),
                                     dy.data_ptr(), dy.stride(0), int(dy.dtype == F32), dx.data_ptr(), dx.stride(0), int(dx.dtype == BF16),
                                     int(accumulate), _p(dgamma), _p(dbeta), ws, M, d, _stream()), "mi_layernorm_bwd")
    return dx


def layernorm_bwd_dual(x, gamma, dy, gamma2, dy2, dx, *, accumulate, dgamma, dbeta, dgamma2, dbeta2, defer, eps=1e-5, cast=None):
    """two LayerNorms of the same rows x (same eps) in ONE pass: dx (+)= dLN(x; gamma)/dx · dy + dLN(x; gamma2)/dx · dy2, both affine gradients deferred to `defer`
    (an LnReduceBatch).  d <= 512.  cast as in layernorm_bwd; -> dx | (dx, cast tensor)."""
    import ctypes as C
    M, d = x.shape
    out = None
    alpha, pdrop, seed, sid = 1.0, 0.0, 0, 0
    if cast is not None:
        alpha, drop = cast
        pdrop, seed, sid = drop if drop is not None else (0.0, 0, 0)
        out = torch.empty((M, d), device=x.device, dtype=BF16)
    defer.reserve(512 * 2 * d)                       # (a growing arena flushes: size it before the pair is taken)
    if len(defer.items) + 2 > defer.SLOTS:          # both partial sets must stay pending together
        defer.flush()
    p1, p2 = defer.slot(512 * 2 * d), defer.slot(512 * 2 * d)
    nblk = C.c_int(0)
    _lib.check(_L().mi_layernorm_bwd_dual_partial(x.data_ptr(), x.stride(0), int(x.dtype == BF16), float(eps), gamma.data_ptr(), dy.data_ptr(), dy.stride(0), int(dy.dtype == F32),
                                                  gamma2.data_ptr(), dy2.data_ptr(), dy2.stride(0), int(dy2.dtype == F32), dx.data_ptr(), dx.stride(0), int(dx.dtype == BF16),
                                                  int(accumulate), p1.data_ptr(), p2.data_ptr(), C.byref(nblk), _p(out), out.stride(0) if out is not None else 0, float(alpha),
                                                  float(pdrop), int(seed) & 0xFFFFFFFF, int(sid) & 0xFFFFFFFF, M, d, _stream()), "mi_layernorm_bwd_dual_partial")
    defer.add(p1, nblk.value, d, dgamma, dbeta)
    defer.add(p2, nblk.value, d, dgamma2, dbeta2)
    return dx if cast is None else (dx, out)


def axpy_(a, b, alpha=1.0):
    """a += alpha * b (contiguous f32)."""
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    _lib.check(_L().mi_axpy_f32(a.data_ptr(), b.data_ptr(), a.numel(), float(alpha), _stream()), "mi_axpy_f32")


def scale_(a, alpha):
    _lib.check(_L().mi_scale_f32(a.data_ptr(), a.numel(), float(alpha), _stream()), "mi_scale_f32")


def scale_by_device_scalar_(a, alpha):
    """a *= alpha for a one-element f32 device tensor `alpha`, without a host sync; costs a launch and nothing else when alpha == 1."""
    assert a.is_contiguous() and a.dtype == F32 and alpha.numel() == 1
    al = alpha.detach().reshape(1).to(device=a.device, dtype=F32)
    _lib.check(_L().mi_scale_dev_f32(a.data_ptr(), a.numel(), al.data_ptr(), _stream()), "mi_scale_dev_f32")


def add_cast(a, b=None, alpha=1.0, out=None):
    """bf16(alpha * (a [+ b])) for f32 (M,N) views."""
    M, N = a.shape
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=BF16)
    _lib.check(_L().mi_add2_cast_bf16(a.data_ptr(), a.stride(0), _p(b), b.stride(0) if b is not None else 0, out.data_ptr(), out.stride(0),
                                      M, N, float(alpha), _stream()), "mi_add2_cast_bf16")
    return out


def add_rowvec(x, vec, out=None):
    M, N = x.shape
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=BF16)
    _lib.check(_L().mi_add_rowvec_bf16(x.data_ptr(), x.stride(0), vec.data_ptr(), out.data_ptr(), out.stride(0), M, N, _stream()), "mi_add_rowvec_bf16")
    return out


def add_rowvec2(x, u, v):
    """-> (bf16(x + u), bf16(x + v)) from one pass over x (bit-identical to two add_rowvec calls)"""
    M, N = x.shape
    ou = torch.empty((M, N), device=x.device, dtype=BF16)
    ov = torch.empty((M, N), device=x.device, dtype=BF16)
    _lib.check(_L().mi_add_rowvec2_bf16(x.data_ptr(), x.stride(0), u.data_ptr(), v.data_ptr(), ou.data_ptr(), ov.data_ptr(), N, M, N, _stream()), "mi_add_rowvec2_bf16")
    return ou, ov


def mask_rows_(x, lengths, T):
    M, N = x.shape
    _lib.check(_L().mi_mask_rows_f32(x.data_ptr(), x.stride(0), lengths.data_ptr(), T, M, N, _stream()), "mi_mask_rows_f32")


def spec_mask_apply_(x, time_mask, embed, feat_mask, T):
    M, N = x.shape
    _lib.check(_L().mi_spec_mask_apply(x.data_ptr(), x.stride(0), _p(time_mask), _p(embed), _p(feat_mask), T, M, N, _stream()), "mi_spec_mask_apply")


def spec_mask_bwd_(dx, time_mask, dembed, feat_mask, T):
    M, N = dx.shape
    ws = _dw_ws(dx.device, ((M + 127) // 128) * N) if (dembed is not None and time_mask is not None) else 0
    _lib.check(_L().mi_spec_mask_bwd(dx.data_ptr(), dx.stride(0), _p(time_mask), _p(dembed), _p(feat_mask), T, M, N, ws, _stream()), "mi_spec_mask_bwd")


def bgemm(A, a_str, B, b_str, C, c_str, Z1, Z2, M, N, K, *, alpha=1.0, accumulate=False, band=None, m_valid=None):
    """C[z1,z2][m][n] = alpha * sum_k A[..][m][k] B[..][n][k] (+C). a_str = (z1, z2, m, k) element strides, b_str = (z1, z2, n, k),
    c_str = (z1, z2, m); A/B bf16 storage, C f32|bf16 with unit column stride.
    band = (T, a, cg): A is banded along K — K = cg * T rows and row (u, i) is zero outside columns [a - i, a - i + T) (the un-shifted relative-position gradient):
    the kernel skips the all-zero k tiles.  m_valid (Z2) int32: rows m >= m_valid[z2] of A are zero for batch entry z2 (keys beyond an utterance's length in
    P^T / dS^T): those M tiles are stored as zeros without being read or multiplied (mi_bgemm_sparse_bf16)."""
    bt, ba, bc = band if band is not None else (0, 0, 0)
    _lib.check(_L().mi_bgemm_sparse_bf16(A.data_ptr(), *[int(s) for s in a_str], B.data_ptr(), *[int(s) for s in b_str],
                                         C.data_ptr(), *[int(s) for s in c_str], int(C.dtype == F32), int(accumulate), float(alpha),
                                         Z1, Z2, M, N, K, int(bt), int(ba), int(bc), _p(m_valid), _stream()), "mi_bgemm_sparse_bf16")
    return C


def band_geometry(T: int):
    """(pad, ldbd) of `attn_bwd_probs`' dbd: columns are relative positions shifted by pad so that every 32-query wave's band starts on a 32-column boundary."""
    pad = (32 - T) % 32
    return pad, (pad + 2 * T - 1 + 31) // 32 * 32


def attention_x_lse(q, k, v, B, Tq, Tk, H, *, lengths=None, causal=False, drop=None):
    """Fused training forward for separate q (B*Tq, .) / k, v (B*Tk, .) bf16 row views (head size 64 / 128, no relative positions): -> (ctx (B*Tq, d) bf16,
    lse (B, H, Tq) fp32).  drop = (p, seed, stream_id): probability dropout with the mask of the generic softmax kernels for element ((h B + b) Tq + i) Tk + j."""
    d = q.shape[1]
    hd = d // H
    out = torch.empty((B * Tq, d), device=q.device, dtype=BF16)
    lse = torch.empty((B, H, Tq), device=q.device, dtype=F32)
    dp, dseed, dsid = drop if drop is not None else (0.0, 0, 0)
    _lib.check(_L().mi_attention_x_lse_bf16(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), _p(lengths), out.data_ptr(), out.stride(0),
                                            lse.data_ptr(), B, Tq, Tk, H, hd, 1.0 / math.sqrt(hd), int(causal), float(dp), int(dseed) & 0xFFFFFFFF, int(dsid) & 0xFFFFFFFF,
                                            _stream()), "mi_attention_x_lse_bf16")
    return out, lse


def attn_x_bwd_probs(q, k, v, B, Tq, Tk, H, ctx, dctx, lse, dq, *, lengths=None, causal=False, drop=None):
    """First half of the fused backward for `attention_x_lse`: -> prob (the dropped probabilities when drop is given), ds (H, B, Tq, Ts) bf16, Ts = Tk rounded up to 32;
    dq (B*Tq, .) bf16 row view receives dS K."""
    d = ctx.shape[1]
    hd = d // H
    Ts = (Tk + 31) // 32 * 32
    prob = torch.empty((H, B, Tq, Ts), device=q.device, dtype=BF16)
    ds = torch.empty((H, B, Tq, Ts), device=q.device, dtype=BF16)
    dp, dseed, dsid = drop if drop is not None else (0.0, 0, 0)
    _lib.check(_L().mi_attention_x_bwd_probs(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), _p(lengths), ctx.data_ptr(), ctx.stride(0),
                                             dctx.data_ptr(), dctx.stride(0), lse.data_ptr(), prob.data_ptr(), ds.data_ptr(), Ts, dq.data_ptr(), dq.stride(0),
                                             B, Tq, Tk, H, hd, 1.0 / math.sqrt(hd), int(causal), float(dp), int(dseed) & 0xFFFFFFFF, int(dsid) & 0xFFFFFFFF, _stream()),
               "mi_attention_x_bwd_probs")
    return prob, ds


_DBD_CACHE = {}          # (device, H, B, T, Ps) -> zero-filled dbd buffer of the sparse-writes walk; at most three shapes per process (least recently used goes first)


def _dbd_static(dev, H, B, T, Ps):
    """the sparse-writes walk's dbd: zero-filled ONCE per shape and then only ever written by that walk (relative positions outside a row's maximal band stay zero from
    launch to launch).  Length-bucketed training alternates between a few shapes: three buffers are kept, so that a shape coming back does not pay its 400-MB fill again."""
    key = (str(dev), H, B, T, Ps)
    buf = _DBD_CACHE.pop(key, None)
    if buf is None:
        while len(_DBD_CACHE) >= 3:
            _DBD_CACHE.pop(next(iter(_DBD_CACHE)))
        buf = torch.zeros((H, B, T, Ps), device=dev, dtype=BF16)
    _DBD_CACHE[key] = buf                                   # (re-)inserted last: most recently used
    return buf


def release_static_buffers():
    """frees the buffers `_dbd_static` keeps between steps (call when a trainer is done with the device)"""
    _DBD_CACHE.clear()


def attn_bwd_probs(qkv, B, T, H, ctx, dctx, lse, dq, *, pos=None, bias_u=None, bias_v=None, lengths=None, causal=False, drop=None, qb=False, sparse=False):
    """First half of the fused attention backward (head size 64 / 128, no probability dropout): -> prob, ds (H, B, T, Ts) bf16 and, with relative positions,
    dbd (H, B, T, Ps) bf16 with dbd[i][T-1-i+j + pad] = ds[i][j] (else None).  Ts = T rounded up to 32; (pad, Ps) = band_geometry(T).  Everything is written.
    dq (B*T, d) bf16 row view receives the query gradient dS K + dBD P; with positions also -> (su, sv): (rows, d) fp32 whose column sums are the gradients
    of pos_bias_u / pos_bias_v.  drop = the (p, seed, stream_id) the forward used: prob is then the DROPPED probabilities (what multiplied V).
    qb (with positions): also -> (qu, qv) (B*T, d) bf16 = q + pos_bias_u, q + pos_bias_v as the kernel's own operands (what `add_rowvec2(q, u, v)` makes), appended to the result.
    sparse: the zeros nobody reads are not written (mi_attention_qkv_bwd_probs_f, flag 1): dbd is then ONE zero-filled buffer per shape that every call re-uses (the caller
    consumes it before the next call on the same stream), and prob / ds hold garbage from the key length rounded up to 128 on — read them only through `bgemm(..., m_valid=lengths)`."""
    d = qkv.shape[1] // 3
    dp, dseed, dsid = drop if drop is not None else (0.0, 0, 0)
    hd = d // H
    Ts = (T + 31) // 32 * 32
    pad, Ps = band_geometry(T)
    dev = qkv.device
    prob = torch.empty((H, B, T, Ts), device=dev, dtype=BF16)
    ds = torch.empty((H, B, T, Ts), device=dev, dtype=BF16)
    rel = pos is not None
    dbd = (_dbd_static(dev, H, B, T, Ps) if sparse else torch.empty((H, B, T, Ps), device=dev, dtype=BF16)) if rel else None
    nw = 4 * ((T + 127) // 128)
    suv = torch.empty((B * nw, 2 * d), device=dev, dtype=F32) if rel else None         # rows of [u | v]: `LnReduceBatch.add_rows2(su, ...)` can defer their column sums
    su, sv = (suv[:, :d], suv[:, d:]) if rel else (None, None)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    qu = torch.empty((B * T, d), device=dev, dtype=BF16) if (qb and rel) else None
    qv = torch.empty((B * T, d), device=dev, dtype=BF16) if (qb and rel) else None
    _lib.check(_L().mi_attention_qkv_bwd_probs_f(q.data_ptr(), qkv.stride(0), k.data_ptr(), qkv.stride(0), v.data_ptr(), qkv.stride(0),
                                                  _p(pos), pos.stride(0) if rel else 0, _p(bias_u), _p(bias_v), _p(lengths),
                                                  ctx.data_ptr(), ctx.stride(0), dctx.data_ptr(), dctx.stride(0), lse.data_ptr(),
                                                  prob.data_ptr(), ds.data_ptr(), Ts, _p(dbd), Ps, pad, dq.data_ptr(), dq.stride(0), _p(su), _p(sv), _p(qu), _p(qv), d,
                                                  B, T, H, hd, 1.0 / math.sqrt(hd), int(causal), float(dp), int(dseed) & 0xFFFFFFFF, int(dsid) & 0xFFFFFFFF, int(bool(sparse)), _stream()),
               "mi_attention_qkv_bwd_probs_f")
    return (prob, ds, dbd, su, sv, qu, qv) if qb else (prob, ds, dbd, su, sv)


def pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def attn_softmax_fwd(ac, bd, lengths, H, B, Tq, Tk, scale, causal=False, drop=None):
    """ac (H,B,Tq,lds) f32, bd (H,B,Tq,ldp) f32 | None (row strides = last-dim sizes) -> prob (H,B,Tq,lds) bf16.
    drop = (p, seed, stream_id): also returns the dropped probabilities (for the PV product); prob itself stays un-dropped."""
    lds = ac.shape[-1]
    ldp = bd.shape[-1] if bd is not None else 0
    prob = torch.empty((H, B, Tq, lds), device=ac.device, dtype=BF16)
    p, seed, sid = drop if drop else (0.0, 0, 0)
    pd = torch.empty_like(prob) if p > 0 else None
    _lib.check(_L().mi_attn_softmax_fwd(ac.data_ptr(), _p(bd), _p(lengths), prob.data_ptr(), _p(pd), H, B, Tq, Tk, lds, ldp, float(scale), int(causal),
                                        float(p), int(seed) & 0xFFFFFFFF, int(sid) & 0xFFFFFFFF, _stream()), "mi_attn_softmax_fwd")
    return (prob, pd) if drop else prob


def attn_softmax_bwd(prob, dp, H, B, Tq, Tk, scale, want_dbd=False, drop=None):
    lds = prob.shape[-1]
    ldp = pad8(2 * Tq - 1)
    ds = torch.empty((H, B, Tq, lds), device=prob.device, dtype=BF16)
    dbd = torch.empty((H, B, Tq, ldp), device=prob.device, dtype=BF16) if want_dbd else None
    p, seed, sid = drop if drop else (0.0, 0, 0)
    _lib.check(_L().mi_attn_softmax_bwd(prob.data_ptr(), dp.data_ptr(), ds.data_ptr(), _p(dbd), H, B, Tq, Tk, lds, ldp, float(scale),
                                        float(p), int(seed) & 0xFFFFFFFF, int(sid) & 0xFFFFFFFF, _stream()), "mi_attn_softmax_bwd")
    return ds, dbd


def dropout_(x, p, seed, stream_id, out=None, alpha=1.0):
    """out (default: in place) = alpha * x * keep / (1 - p) on an (M,N) f32|bf16 row view; mask = f(seed, stream_id, m*N + n)."""
    M, N = x.shape
    out = x if out is None else out
    _lib.check(_L().mi_dropout(x.data_ptr(), x.stride(0), int(x.dtype == BF16), out.data_ptr(), out.stride(0), int(out.dtype == BF16), M, N,
                               float(alpha), float(p), int(seed) & 0xFFFFFFFF, int(stream_id) & 0xFFFFFFFF, _stream()), "mi_dropout")
    return out


def gemm_dropout(a, w, bias, p, seed, stream_id, *, resid=None, alpha=1.0, out=None):
    """[resid +] alpha * dropout(a W^T + b): fp32 (M,N) when `resid` is given (the residual add of a dropped linear output), else bf16 (out may be a column view);
    one launch where the 128 x 128 kernel takes the shape, else the GEMM followed by the dropout kernel — same mask either way."""
    M, K = a.shape
    N = w.shape[0]
    f32 = resid is not None
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=F32 if f32 else BF16)
    rc = _L().mi_gemm_dropout_bf16(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _p(bias), out.data_ptr(), out.stride(0), int(f32),
                                   _p(resid), resid.stride(0) if f32 else 0, float(alpha), float(p), int(seed) & 0xFFFFFFFF, int(stream_id) & 0xFFFFFFFF,
                                   M, N, K, _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        if f32:
            return dropout_add(resid, gemm(a, w, bias, out_dtype=F32), alpha, p, seed, stream_id)
        gemm(a, w, bias, out=out)
        return dropout_(out, p, seed, stream_id)
    _lib.check(rc, "mi_gemm_dropout_bf16")
    return out


def dropout_add(resid, t, alpha, p, seed, stream_id):
    """resid + alpha * dropout(t)  (f32 (M,N))."""
    M, N = t.shape
    y = torch.empty((M, N), device=t.device, dtype=F32)
    _lib.check(_L().mi_dropout_add_f32(y.data_ptr(), y.stride(0), resid.data_ptr(), resid.stride(0), t.data_ptr(), t.stride(0), M, N, float(alpha),
                                       float(p), int(seed) & 0xFFFFFFFF, int(stream_id) & 0xFFFFFFFF, _stream()), "mi_dropout_add_f32")
    return y


def csgu_bwd(u, stats, gamma, beta, w, bias, ds, dr, dgn, dw, db, B, T, pad_left=None, dilation=1, defer=None):
    """dr None: `ds` is the gradient of the conv output itself (split form, ops.csgu_conv).  pad_left / dilation: None / 1 = the symmetric conv; the causal encoder passes ((K-1)*dil, dil = (K-1)//2), as ops.csgu does.
    defer: an LnReduceBatch — the tap / bias gradients' cross-utterance sum is left to its next flush (one launch for several layers' reductions)."""
    M, C2 = u.shape
    Cc = C2 // 2
    K = w.shape[-1]
    pl = (K - 1) // 2 if pad_left is None else int(pad_left)
    rows = B * (((T + 63) // 64) if dilation > 1 else 1)
    if defer is not None and rows * Cc * 32 <= defer.MAX_PER:
        ws = defer.slot(rows * Cc * 32)
        _lib.check(_L().mi_csgu_bwd_bf16(u.data_ptr(), u.stride(0), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(), _p(bias),
                                         ds.data_ptr(), ds.stride(0), _p(dr), dr.stride(0) if dr is not None else 0, dgn.data_ptr(), dgn.stride(0),
                                         None, None, B, T, Cc, K, pl, int(dilation), ws.data_ptr(), _stream()), "mi_csgu_bwd_bf16")
        defer.add_dw(ws, rows, Cc, K, dw, db)
        return
    _lib.check(_L().mi_csgu_bwd_bf16(u.data_ptr(), u.stride(0), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(), _p(bias),
                                     ds.data_ptr(), ds.stride(0), _p(dr), dr.stride(0) if dr is not None else 0, dgn.data_ptr(), dgn.stride(0),
                                     dw.data_ptr(), _p(db), B, T, Cc, K, pl, int(dilation), _dw_ws(u.device, B * Cc * 32 * (((T + 63) // 64) if dilation > 1 else 1)), _stream()), "mi_csgu_bwd_bf16")


def gate_act_mul_bwd(r, g, ds, dr, act=0):
    """backward of ops.gate_act_mul: writes dr = ds * act(g) into `dr` (a column slice is fine), returns dg = ds * r * act'(g)"""
    M, Cc = g.shape
    dg = torch.empty((M, Cc), device=g.device, dtype=BF16)
    _lib.check(_L().mi_gate_act_mul_bwd_bf16(r.data_ptr(), r.stride(0), g.data_ptr(), g.stride(0), ds.data_ptr(), ds.stride(0), dr.data_ptr(), dr.stride(0),
                                             dg.data_ptr(), dg.stride(0), M, Cc, int(act), _stream()), "mi_gate_act_mul_bwd_bf16")
    return dg


def dwconv_residual_bwd(m, w, dy, dm, dw, db, B, T, pad_left=None, defer=None):
    """defer: as in csgu_bwd"""
    M, Cc = m.shape
    K = w.shape[-1]
    pl = (K - 1) // 2 if pad_left is None else int(pad_left)
    if defer is not None and B * Cc * 32 <= defer.MAX_PER:
        ws = defer.slot(B * Cc * 32)
        _lib.check(_L().mi_dwconv_residual_bwd_bf16(m.data_ptr(), m.stride(0), w.data_ptr(), dy.data_ptr(), dy.stride(0), dm.data_ptr(), dm.stride(0),
                                                    None, None, B, T, Cc, K, pl, 1, ws.data_ptr(), _stream()), "mi_dwconv_residual_bwd_bf16")
        defer.add_dw(ws, B, Cc, K, dw, db)
        return
    _lib.check(_L().mi_dwconv_residual_bwd_bf16(m.data_ptr(), m.stride(0), w.data_ptr(), dy.data_ptr(), dy.stride(0), dm.data_ptr(), dm.stride(0),
                                                dw.data_ptr(), _p(db), B, T, Cc, K, pl, 1, _dw_ws(m.device, B * Cc * 32), _stream()), "mi_dwconv_residual_bwd_bf16")


def im2col(x, K, stride, pad, T1, F1):
    """pad = the LEFT / TOP pad (2 * conv pad for the causal front end)"""
    B, T, F, Cin = x.shape
    col = torch.empty((B * T1 * F1, K * K * Cin), device=x.device, dtype=BF16)
    _lib.check(_L().mi_im2col_cl_bf16(x.data_ptr(), col.data_ptr(), B, T, F, Cin, K, K, stride, pad, pad, T1, F1, _stream()), "mi_im2col_cl_bf16")
    return col


def conv2d_wgrad_(dw, dy, x, K, stride, pad, T1, F1, db=None, n_store=None):
    """dw (n_store, K*K*Cin) f32 += dy[:, :N]^T · im2col(x) without the im2col buffer (the gather happens in the GEMM's LDS-DMA source addresses);
    x (B,T,F,Cin) bf16 channels-last, dy (B*T1*F1, N) bf16, pad = the leading pad.  Cin % 128 != 0: the im2col + gemm_tn_ pair."""
    B, T, F, Cin = x.shape
    M, N = dy.shape
    n_store = dw.shape[0] if n_store is None else n_store
    if Cin % 128:
        return gemm_tn_(dw, dy, im2col(x, K, stride, pad, T1, F1), n_store=n_store, db=db)
    nbytes = _L().mi_gemm_tn_workspace_bytes(M, N, K * K * Cin)
    ws = _TN_WS.get(dy.device)
    if nbytes and (ws is None or ws.numel() < nbytes):
        ws = _TN_WS[dy.device] = torch.empty(nbytes, device=dy.device, dtype=torch.uint8)
    _lib.check(_L().mi_conv2d_wgrad_cl_bf16(dy.data_ptr(), dy.stride(0), x.data_ptr(), dw.data_ptr(), dw.stride(0), _p(db), B, T, F, Cin, K, K, stride, pad, pad,
                                            T1, F1, N, n_store, ws.data_ptr() if nbytes else 0, nbytes, _stream()), "mi_conv2d_wgrad_cl_bf16")
    return dw


def im2col_geo(x, K, stride, pad, T1, F1):
    """im2col with (time, freq) kernel / stride / pad pairs"""
    B, T, F, Cin = x.shape
    (KH, KW), (st, sf), (pt, pf) = K, stride, pad
    col = torch.empty((B * T1 * F1, KH * KW * Cin), device=x.device, dtype=BF16)
    _lib.check(_L().mi_im2col_cl_geo_bf16(x.data_ptr(), col.data_ptr(), B, T, F, Cin, KH, KW, st, sf, pt, pf, T1, F1, _stream()), "mi_im2col_cl_geo_bf16")
    return col


def col2im(dcol, shape_in, K, stride, pad, T1, F1, out=None):
    """din (B,T,F,Cin) bf16 (+= when `out` is given) = col2im(dcol (B*T1*F1, KH*KW*Cin))"""
    B, T, F, Cin = shape_in
    (KH, KW), (st, sf), (pt, pf) = K, stride, pad
    acc = out is not None
    if out is None:
        out = torch.empty((B, T, F, Cin), device=dcol.device, dtype=BF16)
    _lib.check(_L().mi_col2im_cl_bf16(dcol.data_ptr(), out.data_ptr(), B, T, F, Cin, KH, KW, st, sf, pt, pf, T1, F1, int(acc), _stream()), "mi_col2im_cl_bf16")
    return out


def gated_act_bwd(dout, z, g, B, T, Fq, C, share=1, dz=None, dg=None):
    """backward of ops.gated_act (plain columns): -> dz like z (B*T*Fq, C), dg like g (B*(T//share)*Fq, C); dz / dg may be column views of one buffer"""
    if dz is None:
        dz = torch.empty((B * T * Fq, C), device=z.device, dtype=BF16)
    if dg is None:
        dg = torch.empty((B * (T // share) * Fq, C), device=z.device, dtype=BF16)
    _lib.check(_L().mi_gated_act_bwd_bf16(dout.data_ptr(), dout.stride(0), z.data_ptr(), z.stride(0), g.data_ptr(), g.stride(0), dz.data_ptr(), dz.stride(0),
                                          dg.data_ptr(), dg.stride(0), B, T, Fq, C, share, _stream()), "mi_gated_act_bwd_bf16")
    return dz, dg


def conv2d_first_wgrad(x, dy, dw, db, K, stride, pad):
    """dw (C, KH*KW) +=, db (C) += from dy (B,T1,F1,C) bf16 = gradient of the raw output of Conv2d(1 -> C, K, stride, pad) over x (B,T,F) f32"""
    B, T, F = x.shape
    _, T1, F1, Cc = dy.shape
    (KH, KW), (st, sf), (pt, pf) = K, stride, pad
    ws = _dw_ws(x.device, int(_L().mi_conv2d_first_wgrad_workspace_floats(B, Cc, KH, KW, T1, F1)))
    _lib.check(_L().mi_conv2d_first_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), B, T, F, Cc, KH, KW, st, sf, pt, pf, T1, F1, ws, _stream()),
               "mi_conv2d_first_wgrad")


def conv2d_first_bwd(x, w, bias, dcol, dw, db, K, stride, pad, T1, F1, K2, stride2, pad2, T2, F2):
    B, T, F = x.shape
    Cc = w.shape[0]
    ws = _dw_ws(x.device, int(_L().mi_conv2d_first_bwd_workspace_floats(B, Cc, T1, F1)))      # one partial row of 10 C floats per block, added in block order
    _lib.check(_L().mi_conv2d_first_bwd(x.data_ptr(), w.data_ptr(), bias.data_ptr(), dcol.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                        B, T, F, Cc, K, stride, pad, pad, T1, F1, K2, stride2, pad2, pad2, T2, F2, ws, _stream()), "mi_conv2d_first_bwd")


def conv2d_s2k3_dgrad_supported(B, T1, F1, C1, T2, F2, pad):
    """elements of the phase buffers of `conv2d_first_bwd_phases`, 0 when the geometry needs the im2col-gradient path"""
    return int(_L().mi_conv2d_s2k3_dgrad_elems(B, T1, F1, C1, T2, F2, pad, pad))


def conv2d_first_bwd_phases(x, w, bias, dy2, wT2, dw, db, K, stride, pad, T1, F1, pad2, T2, F2):
    """Backward of the front end below conv2's pre-activation gradient `dy2` (B*T2*F2, C2) bf16, with conv2 = 3x3 / stride 2: conv2's input gradient as four stride-1
    implicit-GEMM convolutions into phase buffers (mi_conv2d_s2k3_dgrad_bf16; wT2 (9*C1, >= C2) = the transposed weight copy, re-packed per call: the weights move every
    step), then conv1's GELU', weight and bias gradient reading them (mi_conv2d_first_bwd_phases).  The (B*T2*F2, 9*C1) im2col gradient is never formed."""
    B, T, F = x.shape
    C1, C2 = w.shape[0], dy2.shape[1]
    assert dy2.is_contiguous() and dy2.shape[0] == B * T2 * F2 and wT2.shape[0] == 9 * C1
    n = conv2d_s2k3_dgrad_supported(B, T1, F1, C1, T2, F2, pad2)
    assert n > 0
    packed = torch.empty(9 * C1 * C2, device=x.device, dtype=BF16)
    phases = torch.empty(n, device=x.device, dtype=BF16)
    _lib.check(_L().mi_conv2d_s2k3_dgrad_pack_bf16(wT2.data_ptr(), wT2.stride(0), packed.data_ptr(), C1, C2, _stream()), "mi_conv2d_s2k3_dgrad_pack_bf16")
    _lib.check(_L().mi_conv2d_s2k3_dgrad_bf16(dy2.data_ptr(), packed.data_ptr(), phases.data_ptr(), B, T1, F1, C1, T2, F2, C2, pad2, pad2, _stream()), "mi_conv2d_s2k3_dgrad_bf16")
    ws = _dw_ws(x.device, int(_L().mi_conv2d_first_bwd_workspace_floats(B, C1, T1, F1)))
    _lib.check(_L().mi_conv2d_first_bwd_phases(x.data_ptr(), w.data_ptr(), bias.data_ptr(), phases.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                               B, T, F, C1, K, stride, pad, pad, T1, F1, pad2, pad2, T2, F2, ws, _stream()), "mi_conv2d_first_bwd_phases")
    return phases


def ctc_loss_bwd(logits, lse, labels, in_len, nll, *, reduction="mean", gscale=1.0, ldo=None):
    """-> dlogits (B*T, ldo) bf16 of gscale * ctc_loss(reduction), pad columns zero."""
    B, T, V1 = logits.shape
    labels = labels.contiguous()
    U = labels.shape[1]
    ldo = (V1 + 7) // 8 * 8 if ldo is None else ldo
    nbytes = _L().mi_ctc_bwd_workspace_bytes(B, T, U)
    ws = torch.empty(nbytes, device=logits.device, dtype=torch.uint8)
    out = torch.empty((B * T, ldo), device=logits.device, dtype=BF16)
    _lib.check(_L().mi_ctc_loss_bwd(logits.data_ptr(), logits.stride(0), logits.stride(1), 0 if logits.dtype == F32 else 1, lse.data_ptr(), T,
                                    labels.data_ptr(), U, in_len.data_ptr(), V1 - 1, B, 1 if reduction == "mean" else 0, nll.data_ptr(),
                                    float(gscale), ws.data_ptr(), nbytes, out.data_ptr(), ldo, _stream()), "mi_ctc_loss_bwd")
    return out


def ctc_loss_bwd_nll(logits, lse, labels, in_len, *, reduction="mean", zero_infinity=False, gscale=1.0, ldo=None):
    """loss and gradient from ONE pair of recursions: -> (dlogits (B*T, ldo) bf16, loss scalar tensor, nll (B)) as `ops.ctc_loss` + `ctc_loss_bwd` give them, the
    per-utterance nll taken from the backward's own alpha recursion (no forward loss kernel); None when the target is too long for that kernel (callers run the two)."""
    B, T, V1 = logits.shape
    labels = labels.contiguous()
    U = labels.shape[1]
    ldo = (V1 + 7) // 8 * 8 if ldo is None else ldo
    nbytes = _L().mi_ctc_bwd_workspace_bytes(B, T, U)
    ws = torch.empty(nbytes, device=logits.device, dtype=torch.uint8)
    out = torch.empty((B * T, ldo), device=logits.device, dtype=BF16)
    nll = torch.empty((B,), device=logits.device, dtype=F32)
    tl = torch.empty((B,), device=logits.device, dtype=torch.int32)
    loss = torch.empty((1,), device=logits.device, dtype=F32)
    rc = _L().mi_ctc_loss_bwd_nll(logits.data_ptr(), logits.stride(0), logits.stride(1), 0 if logits.dtype == F32 else 1, lse.data_ptr(), T,
                                  labels.data_ptr(), U, in_len.data_ptr(), V1 - 1, B, 1 if reduction == "mean" else 0, int(zero_infinity), float(gscale),
                                  ws.data_ptr(), nbytes, out.data_ptr(), ldo, nll.data_ptr(), tl.data_ptr(), loss.data_ptr(), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "mi_ctc_loss_bwd_nll")
    return out, loss[0], nll


def ce_label_smoothing_bwd(logits, labels, acc, *, shift=1, eps=0.0, weight=1.0, ldo=None):
    B, U, V = logits.shape
    labels = labels.contiguous()
    ldo = (V + 7) // 8 * 8 if ldo is None else ldo
    out = torch.empty((B * U, ldo), device=logits.device, dtype=BF16)
    _lib.check(_L().mi_ce_label_smoothing_bwd(logits.data_ptr(), logits.stride(1), labels.data_ptr(), B, U, shift, V, float(eps), float(weight),
                                              acc.data_ptr(), out.data_ptr(), ldo, _stream()), "mi_ce_label_smoothing_bwd")
    return out


def embed_tokens_bwd(ids, dx, dwte, dwpe=None, *, scale=1.0, pos_offset=0, heavy_id=None):
    """dwte[ids[m]] += scale * dx[m]; dwpe[pos_offset + m % U] += dx[m].  heavy_id: a token expected on a large share of the rows (the padding token the shifted decoder
    input is filled with): summed as a masked column sum instead of one block's walk over its rows."""
    ids = ids.contiguous()
    M = ids.numel()
    U = ids.shape[-1]
    V, d = dwte.shape
    ws = _dw_ws(dx.device, (int(_L().mi_embed_tokens_bwd_workspace_bytes(M, d, V)) + 3) // 4)
    _lib.check(_L().mi_embed_tokens_bwd(ids.data_ptr(), dx.data_ptr(), float(scale), pos_offset, U, d, M, V, dwte.data_ptr(), _p(dwpe),
                                        int(heavy_id) if (heavy_id is not None and 0 <= int(heavy_id) < V) else -1, ws, _stream()), "mi_embed_tokens_bwd")


def softmax_vec(w, out=None):
    """softmax of a short f32 vector on the device (layer-mixing weights, bestrq.py:242)"""
    out = torch.empty_like(w) if out is None else out
    _lib.check(_L().mi_softmax_vec_f32(w.data_ptr(), w.numel(), out.data_ptr(), _stream()), "mi_softmax_vec_f32")
    return out


def softmax_vec_bwd_(dw, s, g):
    """dw += s * (g - <s, g>)"""
    _lib.check(_L().mi_softmax_vec_bwd_f32(s.data_ptr(), g.data_ptr(), s.numel(), dw.data_ptr(), _stream()), "mi_softmax_vec_bwd_f32")


def axpy_dev_(a, b, alpha_dev, overwrite=False):
    """a = (0 if overwrite else a) + alpha_dev[0] * b — contiguous f32, the coefficient is a one-element DEVICE tensor (view)."""
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel() and a.dtype == b.dtype == torch.float32
    _lib.check(_L().mi_axpy_dev_f32(a.data_ptr(), b.data_ptr(), a.numel(), alpha_dev.data_ptr(), int(overwrite), _stream()), "mi_axpy_dev_f32")
    return a


def dot_(acc, a, b):
    """acc[0] += <a, b> (contiguous f32; deterministic)"""
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel() and a.dtype == b.dtype == torch.float32
    _lib.check(_L().mi_dot_f32(a.data_ptr(), b.data_ptr(), a.numel(), acc.data_ptr(), _dw_ws(a.device, 1024), _stream()), "mi_dot_f32")


def sumsq_(acc, x):
    _lib.check(_L().mi_sumsq_f32(x.data_ptr(), x.numel(), acc.data_ptr(), _dw_ws(x.device, 1024), _stream()), "mi_sumsq_f32")


def clip_coef(sumsq, max_norm, out, skip_above=0.0):
    """out (3 floats) <- [norm, clip coefficient, skip flag]; skip when the norm is non-finite or above `skip_above` (> 0)"""
    assert out.numel() >= 3
    _lib.check(_L().mi_clip_coef(sumsq.data_ptr(), float(max_norm), float(skip_above or 0.0), out.data_ptr(), _stream()), "mi_clip_coef")


def adamw_step_(p, g, m, v, decay, *, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, step=1, norm_coef=None, mirror=None):
    _lib.check(_L().mi_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(decay), p.numel(), float(lr), float(betas[0]),
                                  float(betas[1]), float(eps), float(weight_decay), int(step), _p(norm_coef), _p(mirror), _stream()), "mi_adamw_step")


_TN_WS = {}


class TnBatch:
    """Deferred weight-gradient GEMMs: `gemm_tn_(..., defer=batch)` / `linear_bwd(..., defer=batch)` only record (dW, dY, X, db) — the tensors stay referenced, so their
    memory is not reused — and `flush()` runs them as ONE grouped launch (mi_gemm_tn_group_bf16): a base-size layer's ten dW GEMMs have ~230 output tiles of 256 x 128 between
    them, enough to fill the chip without splitting M, hence no slabs and no reduce passes.  Below `MIN_TILES` tiles a non-final flush keeps recording (small models:
    several layers per launch); a final flush with less than half of that (a mostly frozen model) runs the problems one by one on the split-M path.  Callers must not modify a recorded dY / X in place before the flush."""
    MAX, MIN_TILES, FULL_TILES = 48, 128, 200

    def __init__(self):
        self.items = []
        self._overwrite = False
        self._written = set()           # targets (dW / db pointers) written since `overwrite` was armed: a later launch of the same backward that names one of them must ADD

    @property
    def overwrite(self):
        return self._overwrite

    @overwrite.setter
    def overwrite(self, v):             # set by the trainer for the first backward after zero_grad: the grouped launches write their targets instead of adding into them
        self._overwrite = bool(v)
        self._written = set()

    def add(self, dw, dy, x, n_store, db):
        self.items.append((dw, dy, x, n_store, db))
        if len(self.items) == self.MAX:
            self.flush()

    def tiles(self, tk=128):
        return sum(-(-dy.shape[1] // 256) * -(-x.shape[1] // tk) for _, dy, x, _, _ in self.items)          # 256 (n) x tk (k) output tiles, one block per CU

    def flush(self, final=True, room=12, tile_k=0):
        """runs what is recorded.  final=False: only once the recorded problems fill the chip with 256 x 256 output tiles (FULL_TILES: two base-size layers), or when
        `room` more problems would not fit — otherwise they stay recorded and False is returned: the caller's gradients are not final yet.  The kernel takes
        256 x 256 tiles when there are >= 200 of them, else 256 x 128 (tile_k = 0; a small model's 48 problems have ~130 / ~220)."""
        if not self.items:
            return True
        if not final and self.tiles(256) < self.FULL_TILES and len(self.items) + room <= self.MAX:
            return False
        items, self.items = self.items, []
        # overwrite (the first backward after zero_grad): a launch STORES its targets, so every target may be named once per launch and must not have been written by an
        # earlier launch of this backward (the auto-flush at MAX and the per-layer-pair flushes split a backward over several launches: a weight shared across them — tied,
        # or a head reused by an intermediate loss — would otherwise lose its first contribution).  Problems whose dW or db was already written — by an earlier launch or by
        # an earlier problem of this one — leave the grouped launch and ADD afterwards, one by one, in recording order; the rest of the launch still stores.
        again = []
        if self._overwrite:
            fresh = []
            for it in items:
                t = [it[0].data_ptr()] + ([it[4].data_ptr()] if it[4] is not None else [])
                (again if any(q in self._written for q in t) else fresh).append(it)
                self._written.update(t)
            items = fresh
        self._launch(items, tile_k, ow=self._overwrite)
        for dw, dy, x, n_store, db in again:
            gemm_tn_(dw, dy, x, n_store=n_store, db=db)
        return True

    def _launch(self, items, tile_k, ow):
        if not items:
            return
        if sum(-(-dy.shape[1] // 256) * -(-x.shape[1] // 128) for _, dy, x, _, _ in items) < self.MIN_TILES // 2:
            for dw, dy, x, n_store, db in items:          # the one-by-one path always adds (the trainer's targets are zero after zero_grad: adding is storing)
                gemm_tn_(dw, dy, x, n_store=n_store, db=db)
            return
        import ctypes as C
        n = len(items)
        vp, lg, it = (C.c_void_p * n), (C.c_long * n), (C.c_int * n)
        _lib.check(_L().mi_gemm_tn_group_ow_bf16(
            n, vp(*[dy.data_ptr() for _, dy, _, _, _ in items]), lg(*[dy.stride(0) for _, dy, _, _, _ in items]),
            vp(*[x.data_ptr() for _, _, x, _, _ in items]), lg(*[x.stride(0) for _, _, x, _, _ in items]),
            vp(*[dw.data_ptr() for dw, _, _, _, _ in items]), lg(*[dw.stride(0) for dw, _, _, _, _ in items]),
            vp(*[(db.data_ptr() if db is not None else None) for _, _, _, _, db in items]),
            it(*[dy.shape[0] for _, dy, _, _, _ in items]), it(*[dy.shape[1] for _, dy, _, _, _ in items]), it(*[x.shape[1] for _, _, x, _, _ in items]),
            it(*[ns for _, _, _, ns, _ in items]), int(tile_k), int(bool(ow)), _stream()), "mi_gemm_tn_group_bf16")


def gemm_tn_(dw, dy, x, n_store=None, db=None, variant=0, defer=None):
    """dw (n_store, K) f32 += dy[:, :N]^T · x   (dy (M,N) bf16, x (M,K) bf16 row views; contraction over rows, no transposes).
    db (n_store) f32: the bias gradient db += column sums of dy, computed from the same LDS tiles.
    defer: a TnBatch — the product is recorded and runs with the batch's next flush()."""
    M, N = dy.shape
    K = x.shape[1]
    n_store = dw.shape[0] if n_store is None else n_store
    if defer is not None:
        defer.add(dw, dy, x, n_store, db)
        return dw
    nbytes = _L().mi_gemm_tn_workspace_bytes(M, N, K)
    key = dy.device
    ws = _TN_WS.get(key)
    if nbytes and (ws is None or ws.numel() < nbytes):
        ws = _TN_WS[key] = torch.empty(nbytes, device=dy.device, dtype=torch.uint8)
    _lib.check(_L().mi_gemm_tn_bf16(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), dw.stride(0), _p(db), M, N, K, n_store,
                                    ws.data_ptr() if nbytes else 0, nbytes, int(variant), _stream()), "mi_gemm_tn_bf16")
    return dw


# ---------------------------------------------------------------------------------------------------------------- composites
def linear_bwd(dy, x, wT, *, dw=None, db=None, dx_out=None, dx_dtype=BF16, need_dx=True, defer=None):
    """Backward of y = x W^T + b for bf16 row-major activations.
    dy (M,N) bf16, x (M,K) bf16, wT (K,N) bf16 (the transposed copy of W the trainer keeps).
    dx = dy · W (GEMM with W^T as the (N',K') operand); dW (N,K) f32 += dy^T · x (row-contraction GEMM, gemm_tn.hip); db += colsum(dy)."""
    M, N = dy.shape
    K = x.shape[1]
    dx = None
    if need_dx:
        dx = gemm(dy, wT[:, :N], out=dx_out, out_dtype=dx_dtype)
    if dw is not None:
        gemm_tn_(dw, dy, x, db=db, defer=defer)    # bias gradient fused into the weight-gradient GEMM
    elif db is not None:
        colsum_(db, dy)
    return dx
