"""GPU parity of the training-step kernels (backward + optimizer), op by op: each C-ABI entry point against torch autograd
(CPU, fp32) of the ORACLE's restatement of the forward op, on identical bf16-rounded inputs.

Tolerances: bf16-stored gradients at ~1 bf16 ulp relative + a small absolute floor scaled to the gradient magnitude;
fp32 parameter gradients (sums over thousands of bf16 products) at 1 % of the tensor's max |value|."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import ebranchformer_ref as R

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
BF = torch.bfloat16


def _o():
    from huggingface_asr_amd import ops, ops_train
    return ops, ops_train


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bfr(x):
    return x.to(BF).float()


def dev16(x):
    return x.to(DEV, BF)


def close(got, want, rel=1.2e-2, floor=2e-2, what=""):
    """|got - want| <= floor * max|want| + rel * |want|"""
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs()
    tol = floor * want.abs().max() + rel * want.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g} (max |want| {float(want.abs().max()):.4g})"


def test_transpose_colsum():
    ops, T = _o()
    x = bfr(rnd(250, 136, seed=1))
    xt = T.transpose(dev16(x))
    assert xt.shape == (136, 256)
    assert torch.equal(xt[:, :250].float().cpu(), x.t())
    assert float(xt[:, 250:].float().abs().max()) == 0.0
    view = dev16(torch.cat([x, x], 1))[:, 136:]                      # strided view
    assert torch.equal(T.transpose(view)[:, :250].float().cpu(), x.t())
    out = torch.zeros(136, device=DEV)
    T.colsum_(out, dev16(x)); T.colsum_(out, x.to(DEV))
    torch.testing.assert_close(out.cpu(), 2 * x.sum(0), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("kind", ["gelu", "gelu_new"])
def test_act_fwd_bwd(kind):
    ops, T = _o()
    pre = bfr(rnd(100, 256, seed=2, scale=2.0)).requires_grad_(True)
    dy = bfr(rnd(100, 256, seed=3))
    f = (lambda v: F.gelu(v)) if kind == "gelu" else (lambda v: F.gelu(v, approximate="tanh"))
    y = f(pre)
    y.backward(dy)
    close(T.act_fwd(dev16(pre.detach()), kind), y.detach(), floor=2e-3, what="act fwd")
    close(T.act_bwd(dev16(dy), dev16(pre.detach()), kind), pre.grad, floor=2e-3, what="act bwd")


@pytest.mark.parametrize("d,xbf", [(64, False), (512, False), (1024, True), (768, False), (2048, True)])
def test_layernorm_bwd(d, xbf):
    ops, T = _o()
    M = 300
    x = rnd(M, d, seed=4, scale=1.5) + 0.3
    if xbf:
        x = bfr(x)
    g, b = 1 + 0.1 * rnd(d, seed=5), 0.1 * rnd(d, seed=6)
    dy = bfr(rnd(M, d, seed=7))
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.layer_norm(xr, (d,), gr, br, 1e-5).backward(dy)
    base = rnd(M, d, seed=8)
    dx = base.clone().to(DEV)
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    T.layernorm_bwd(x.to(DEV, BF) if xbf else x.to(DEV), g.to(DEV), dev16(dy), dx, accumulate=True, dgamma=dg, dbeta=db)
    torch.testing.assert_close(dx.cpu() - base, xr.grad, atol=2e-4, rtol=1e-3)
    torch.testing.assert_close(dg.cpu(), gr.grad, atol=2e-3, rtol=1e-3)
    torch.testing.assert_close(db.cpu(), br.grad, atol=2e-3, rtol=1e-3)
    dxb = torch.empty(M, d, device=DEV, dtype=BF)
    T.layernorm_bwd(x.to(DEV, BF) if xbf else x.to(DEV), g.to(DEV), dy.to(DEV), dxb, accumulate=False)
    close(dxb, xr.grad, floor=4e-3, what="ln bwd bf16 out")


@pytest.mark.parametrize("M,N,K", [(250, 512, 256), (1000, 64, 128), (333, 136, 72)])
def test_linear_bwd(M, N, K):
    ops, T = _o()
    x, w, dy = bfr(rnd(M, K, seed=1)), bfr(rnd(N, K, seed=2, scale=K ** -0.5)), bfr(rnd(M, N, seed=3))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), torch.zeros(N, requires_grad=True)
    F.linear(xr, wr, br).backward(dy)
    dw = torch.zeros(N, K, device=DEV)
    db = torch.zeros(N, device=DEV)
    dx = T.linear_bwd(dev16(dy), dev16(x), dev16(w.t().contiguous()), dw=dw, db=db)
    close(dx, xr.grad, what="dx")
    torch.testing.assert_close(dw.cpu(), wr.grad, atol=2e-3 * float(wr.grad.abs().max()), rtol=1e-3)
    torch.testing.assert_close(db.cpu(), br.grad, atol=1e-3, rtol=1e-4)
    T.linear_bwd(dev16(dy), dev16(x), dev16(w.t().contiguous()), dw=dw, need_dx=False)        # accumulates
    torch.testing.assert_close(dw.cpu(), 2 * wr.grad, atol=4e-3 * float(wr.grad.abs().max()), rtol=1e-3)


@pytest.mark.parametrize("variant", [0, 1])          # 0: 128 x 128 output tiles (product), 1: 128 x 64
@pytest.mark.parametrize("M,N,K,ns", [(8000, 512, 512, 512), (499, 64, 64, 64), (1000, 136, 264, 131), (4000, 5056, 64, 5001), (130, 128, 2304, 128), (8000, 2048, 512, 2048)])
def test_gemm_tn_weight_gradient(M, N, K, ns, variant):
    ops, T = _o()
    dy, x = bfr(rnd(M, N, seed=1)), bfr(rnd(M, K, seed=2))
    want = (dy.t() @ x)[:ns]
    base = rnd(ns, K, seed=3)
    dw = base.clone().to(DEV)
    db = torch.ones(ns, device=DEV)
    T.gemm_tn_(dw, dev16(dy), dev16(x), n_store=ns, db=db, variant=variant)
    torch.testing.assert_close(dw.cpu() - base, want, atol=3e-3 * float(want.abs().max()), rtol=1e-3)
    torch.testing.assert_close(db.cpu() - 1.0, dy.sum(0)[:ns], atol=2e-3 * float(dy.sum(0).abs().max()) + 1e-3, rtol=1e-3)      # fused bias gradient
    # strided operand views (a column block of a wider activation)
    wide = dev16(torch.cat([x, dy], 1))
    dw2 = torch.zeros(ns, K, device=DEV)
    T.gemm_tn_(dw2, wide[:, K:], wide[:, :K], n_store=ns, variant=variant)
    torch.testing.assert_close(dw2.cpu(), want, atol=3e-3 * float(want.abs().max()), rtol=1e-3)


@pytest.mark.parametrize("tile_k", [128, 256, 0])
def test_grouped_weight_gradients_one_launch(tile_k):
    """ops_train.TnBatch / mi_gemm_tn_group_bf16: problems of different M, ragged N / K, stored-row limits and bias gradients as ONE launch, on both output
    tiles (256 x 128, 256 x 256) and with the entry point choosing; each adds into its dW in place."""
    ops, T = _o()
    shapes = [(2000, 512, 512, 512, True), (333, 136, 264, 131, True), (2000, 2048, 512, 2048, False), (777, 64, 1024, 64, True), (1500, 520, 72, 520, False),
              (2000, 512, 2048, 512, True)]
    b = T.TnBatch()
    keep, wants = [], []
    for i, (M, N, K, ns, bias) in enumerate(shapes):
        dy, x = bfr(rnd(M, N, seed=10 + i)), bfr(rnd(M, K, seed=30 + i))
        base = rnd(ns, K, seed=50 + i)
        dw = base.clone().to(DEV)
        db = torch.ones(ns, device=DEV) if bias else None
        T.gemm_tn_(dw, dev16(dy), dev16(x), n_store=ns, db=db, defer=b)
        keep.append((dw, db, base))
        wants.append(((dy.t() @ x)[:ns], dy.sum(0)[:ns]))
    assert len(b.items) == len(shapes) and b.flush(final=False, room=100, tile_k=tile_k) is True          # no room for `room` more: it must run now
    for (dw, db, base), (want, wb) in zip(keep, wants):
        torch.testing.assert_close(dw.cpu() - base, want, atol=3e-3 * float(want.abs().max()), rtol=1e-3)
        if db is not None:
            torch.testing.assert_close(db.cpu() - 1.0, wb, atol=2e-3 * float(wb.abs().max()) + 1e-3, rtol=1e-3)
    # a non-final flush of a batch that neither fills the chip nor is about to overflow keeps recording
    b2 = T.TnBatch()
    dw = torch.zeros(64, 64, device=DEV)
    T.gemm_tn_(dw, dev16(rnd(100, 64, seed=1)), dev16(rnd(100, 64, seed=2)), defer=b2)
    assert b2.flush(final=False) is False and len(b2.items) == 1 and float(dw.abs().max()) == 0.0
    assert b2.flush() is True and float(dw.abs().max()) > 0.0


def test_overwriting_weight_gradient_launches_add_into_a_target_an_earlier_launch_wrote():
    """in overwrite mode (the first backward after zero_grad) a grouped launch STORES its targets; a backward is split over several launches (auto-flush at 48
    problems, one flush per layer pair), so a target named by two of them — a tied weight, a head reused by an intermediate loss — must be written by the first and ADDED
    to by the second.  Ten 512 x 512 problems per launch (the grouped kernel, not the one-by-one path); targets start as garbage."""
    ops, T = _o()
    b = T.TnBatch()
    b.overwrite = True
    shared = torch.full((512, 512), 7.0, device=DEV)
    sb = torch.full((512,), -3.0, device=DEV)
    want_shared, want_sb = torch.zeros(512, 512), torch.zeros(512)
    others = []
    for launch in range(2):
        for i in range(10):
            dy, x = bfr(rnd(700, 512, seed=100 * launch + i)), bfr(rnd(700, 512, seed=100 * launch + 50 + i))
            if i == 4:
                dw, db = shared, sb
                want_shared += dy.t() @ x
                want_sb += dy.sum(0)
            else:
                dw, db = torch.full((512, 512), 5.0, device=DEV), torch.full((512,), 2.0, device=DEV)
                others.append((dw, db, dy.t() @ x, dy.sum(0)))
            T.gemm_tn_(dw, dev16(dy), dev16(x), db=db, defer=b)
        assert b.tiles() >= b.MIN_TILES // 2 and b.flush() is True
    torch.testing.assert_close(shared.cpu(), want_shared, atol=3e-3 * float(want_shared.abs().max()), rtol=1e-3)
    torch.testing.assert_close(sb.cpu(), want_sb, atol=3e-3 * float(want_sb.abs().max()), rtol=1e-3)
    for dw, db, w, wb in others:                      # written once: the garbage is gone
        torch.testing.assert_close(dw.cpu(), w, atol=3e-3 * float(w.abs().max()), rtol=1e-3)
        torch.testing.assert_close(db.cpu(), wb, atol=3e-3 * float(wb.abs().max()), rtol=1e-3)
    b.overwrite = True                                 # the next zero_grad re-arms it: the same target is stored again
    dy, x = bfr(rnd(700, 512, seed=900)), bfr(rnd(700, 512, seed=901))
    for i in range(10):
        T.gemm_tn_(shared if i == 0 else torch.empty(512, 512, device=DEV), dev16(dy), dev16(x), defer=b)
    b.flush()
    torch.testing.assert_close(shared.cpu(), dy.t() @ x, atol=3e-3 * float((dy.t() @ x).abs().max()), rtol=1e-3)


@pytest.mark.parametrize("kind", ["gelu", "gelu_new"])
@pytest.mark.parametrize("M,N,K,p", [(1000, 512, 256, 0.0), (777, 1024, 512, 0.1), (300, 136, 72, 0.1)])       # the last shape is outside the fused kernel: two launches
def test_ffn_activation_passes_in_the_gemm_epilogues(M, N, K, p, kind):
    """mi_gemm_act_fwd_bf16 / mi_gemm_act_bwd_bf16 against the GEMM + element-wise pair they replace: identical bits, with and without activation dropout."""
    ops, T = _o()
    a, w, b = dev16(rnd(M, K, seed=1)), dev16(rnd(N, K, seed=2, scale=0.1)), rnd(N, seed=3).to(DEV)
    drop = (p, 1234, 77) if p > 0 else None
    pre_want = ops.gemm(a, w, b)
    h_want = T.act_fwd(pre_want, kind, drop=drop)
    pre, h = T.gemm_act_fwd(a, w, b, kind, drop=drop)
    assert torch.equal(pre, pre_want) and torch.equal(h, h_want)
    # values: against torch on the bf16-rounded pre-activation
    ref = F.gelu(pre_want.float().cpu(), approximate="none" if kind == "gelu" else "tanh")
    if drop is None:
        close(h, ref, floor=4e-3, what="act")
    dy, wt = dev16(rnd(M, K, seed=4)), dev16(rnd(N, K, seed=5, scale=0.1))                  # dh = dy (M,K) · wt^T, wt (N,K) rows of the transposed weight
    want = T.act_bwd(ops.gemm(dy, wt), pre_want, kind, drop=drop)
    got = T.gemm_act_bwd(dy, wt, pre_want, kind, drop=drop)
    assert torch.equal(got, want)


@pytest.mark.parametrize("B,T1,F1,Cin,Cout,pad", [(2, 61, 39, 128, 256, 1), (3, 50, 19, 256, 64, 2), (1, 9, 7, 128, 136, 1)])
def test_conv2d_weight_gradient_without_the_im2col_buffer(B, T1, F1, Cin, Cout, pad):
    """mi_conv2d_wgrad_cl_bf16 (rows of the im2col operand gathered by the GEMM's LDS-DMA addresses) against im2col + the row-contraction GEMM and against torch."""
    ops, T = _o()
    K, st = 3, 2
    # the kernels take the LEADING pad only (the trailing side follows from T2 / F2): a geometry where every window starts inside the input
    T2, F2 = (T1 + pad - K) // st + 1, (F1 + pad - K) // st + 1
    x = bfr(rnd(B, T1, F1, Cin, seed=1))
    dy = bfr(rnd(B * T2 * F2, Cout, seed=2))
    xd, dyd = dev16(x), dev16(dy)
    col = T.im2col(xd, K, st, pad, T2, F2)
    want = torch.zeros(Cout, K * K * Cin, device=DEV); wb = torch.zeros(Cout, device=DEV)
    T.gemm_tn_(want, dyd, col, db=wb)
    got = torch.zeros(Cout, K * K * Cin, device=DEV); gb = torch.zeros(Cout, device=DEV)
    T.conv2d_wgrad_(got, dyd, xd, K, st, pad, T2, F2, db=gb)
    torch.testing.assert_close(got, want, atol=1e-4 * float(want.abs().max()), rtol=1e-5)
    torch.testing.assert_close(gb, wb, atol=1e-4 * float(wb.abs().max()), rtol=1e-5)
    # and the definition: conv weight gradient of torch on the leading-padded input
    xp = F.pad(x.permute(0, 3, 1, 2), (pad, K, pad, K)).requires_grad_(False)            # generous trailing pad: windows that run past it read zeros
    w = torch.zeros(Cout, Cin, K, K, requires_grad=True)
    y = F.conv2d(xp, w, stride=st)[:, :, :T2, :F2]
    y.backward(dy.view(B, T2, F2, Cout).permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, K * K * Cin)
    close(got, ref, rel=1e-2, floor=3e-3, what="conv2 weight gradient")


@pytest.mark.parametrize("M,N,K", [(1000, 512, 2048), (777, 256, 1024), (300, 512, 512), (200, 136, 72)])      # the last shape is outside the fused kernel: two launches
def test_linear_output_dropout_in_the_gemm_epilogue(M, N, K):
    """mi_gemm_dropout_bf16 against the GEMM + dropout kernels it replaces: the fp32 residual form (x + alpha * dropout(linear)) and the bf16 in-place form."""
    ops, T = _o()
    a, w, b = dev16(rnd(M, K, seed=1)), dev16(rnd(N, K, seed=2, scale=0.05)), rnd(N, seed=3).to(DEV)
    x = rnd(M, N, seed=4).to(DEV)
    p, seed, sid = 0.1, 99, 1234
    want = T.dropout_add(x, ops.gemm(a, w, b, out_dtype=torch.float32), 0.5, p, seed, sid)
    got = T.gemm_dropout(a, w, b, p, seed, sid, resid=x, alpha=0.5)
    torch.testing.assert_close(got, want, atol=1e-5 * float(want.abs().max()), rtol=1e-5)            # same mask; the products may be contracted differently
    dropped = T.dropout_add(torch.zeros_like(x), torch.ones_like(x), 1.0, p, seed, sid) == 0
    assert torch.equal(got[dropped], x[dropped]) and 0.05 < float(dropped.float().mean()) < 0.15      # dropped elements leave the residual untouched
    wide = torch.zeros((M, 2 * N), device=DEV, dtype=BF)
    want16 = T.dropout_(ops.gemm(a, w, b), p, seed, sid)
    got16 = T.gemm_dropout(a, w, b, p, seed, sid, out=wide[:, :N])
    assert torch.equal(got16, want16) and float(wide[:, N:].float().abs().max()) == 0.0


def test_bgemm_modes():
    ops, T = _o()
    Z1, Z2, M, N, K = 3, 2, 70, 50, 90
    A = bfr(rnd(Z1, Z2, M, K, seed=1)); Bm = bfr(rnd(Z1, Z2, N, K, seed=2))
    want = torch.einsum("abmk,abnk->abmn", A, Bm)
    for ta in (False, True):
        for tb in (False, True):
            a = dev16(A.transpose(2, 3).contiguous()) if ta else dev16(A)
            b = dev16(Bm.transpose(2, 3).contiguous()) if tb else dev16(Bm)
            a_str = (a.stride(0), a.stride(1), 1, a.stride(2)) if ta else (a.stride(0), a.stride(1), a.stride(2), 1)
            b_str = (b.stride(0), b.stride(1), 1, b.stride(2)) if tb else (b.stride(0), b.stride(1), b.stride(2), 1)
            C = torch.zeros(Z1, Z2, M, N, device=DEV)
            T.bgemm(a, a_str, b, b_str, C, (C.stride(0), C.stride(1), C.stride(2)), Z1, Z2, M, N, K, alpha=0.5)
            torch.testing.assert_close(C.cpu(), 0.5 * want, atol=2e-3, rtol=1e-4)
            T.bgemm(a, a_str, b, b_str, C, (C.stride(0), C.stride(1), C.stride(2)), Z1, Z2, M, N, K, alpha=0.5, accumulate=True)
            torch.testing.assert_close(C.cpu(), want, atol=4e-3, rtol=1e-4)
    Cb = torch.zeros(Z1, Z2, M, N, device=DEV, dtype=BF)
    a, b = dev16(A), dev16(Bm)
    T.bgemm(a, (a.stride(0), a.stride(1), a.stride(2), 1), b, (b.stride(0), b.stride(1), b.stride(2), 1), Cb, (Cb.stride(0), Cb.stride(1), Cb.stride(2)), Z1, Z2, M, N, K)
    close(Cb, want, what="bgemm bf16 out")


@pytest.mark.parametrize("rel,causal,masked,Tq", [(True, False, True, 37), (False, True, False, 37), (False, False, True, 37),
                                                  (True, False, True, 250),      # the encoder's row length: four registers per lane
                                                  (True, False, True, 290), (False, False, True, 290)])    # rows beyond 256 keys: the loop-form kernels
def test_attn_softmax_fwd_bwd(rel, causal, masked, Tq):
    ops, T = _o()
    H, B = 2, 3
    Tk = Tq if (rel or causal) else Tq + 16
    P = 2 * Tq - 1
    scale = 0.25
    ac = rnd(H, B, Tq, Tk, seed=1, scale=2.0).requires_grad_(True)
    bd = rnd(H, B, Tq, P, seed=2, scale=2.0).requires_grad_(True) if rel else None
    lens = torch.tensor([Tk, Tk - 9, 5], dtype=torch.int32) if masked else None
    s = ac
    if rel:
        idx = (Tq - 1) - torch.arange(Tq)[:, None] + torch.arange(Tq)[None, :]
        s = s + torch.gather(bd, 3, idx[None, None].expand(H, B, Tq, Tq))
    s = s * scale
    fmin = torch.finfo(torch.float32).min
    if masked:
        km = torch.arange(Tk)[None, :] >= lens[:, None]
        s = s.masked_fill(km[None, :, None, :], fmin)
    if causal:
        s = s.masked_fill(torch.ones(Tq, Tk, dtype=torch.bool).triu(1)[None, None], fmin)
    prob = torch.softmax(s, -1)
    dp = rnd(H, B, Tq, Tk, seed=3)
    Ts, Ps = T.pad8(Tk), T.pad8(P)                      # padded row strides, as the trainer allocates them
    acp = torch.zeros(H, B, Tq, Ts); acp[..., :Tk] = ac.detach()
    bdp = None
    if rel:
        bdp = torch.zeros(H, B, Tq, Ps); bdp[..., :P] = bd.detach()
    got = T.attn_softmax_fwd(acp.to(DEV), bdp.to(DEV) if rel else None, lens.to(DEV) if masked else None, H, B, Tq, Tk, scale, causal)[..., :Tk]
    close(got, prob.detach(), floor=4e-3, what="prob")
    # backward through the bf16-rounded probabilities the kernel consumes
    pb = bfr(prob.detach())
    ds_want = pb * (dp - (pb * dp).sum(-1, keepdim=True)) * scale
    pbp = torch.zeros(H, B, Tq, Ts); pbp[..., :Tk] = pb
    dpp = torch.zeros(H, B, Tq, Ts); dpp[..., :Tk] = dp
    ds, dbd = T.attn_softmax_bwd(dev16(pbp), dpp.to(DEV), H, B, Tq, Tk, scale, want_dbd=rel)
    ds = ds[..., :Tk]
    if rel:
        dbd = dbd[..., :P]
    close(ds, ds_want, floor=4e-3, what="ds")
    prob.backward(dp)
    close(ds, ac.grad, floor=2e-2, what="ds vs autograd")
    if rel:
        close(dbd, bd.grad, floor=2e-2, what="dbd vs autograd")


@pytest.mark.parametrize("T_,H,hd,rel,causal,pdrop", [(250, 4, 128, True, False, 0.0), (97, 2, 64, True, False, 0.0), (75, 2, 64, False, False, 0.0), (130, 2, 128, False, True, 0.0),
                                                      (33, 1, 128, True, False, 0.0), (160, 2, 64, True, True, 0.0), (500, 2, 128, True, False, 0.0),
                                                      (250, 4, 128, True, False, 0.1), (97, 2, 64, True, False, 0.25), (75, 2, 64, False, True, 0.1)])
def test_attention_backward_recomputation(T_, H, hd, rel, causal, pdrop):
    """mi_attention_qkv_bwd_probs (P, dS and the un-shifted dBD from one walk over the keys, given the fused forward's context and log-sum-exp) against
    autograd of the oracle-style attention: scores (q+u)k^T + rel_shift((q+v)p^T), key-padding / causal mask, softmax, P V."""
    ops, T = _o()
    B, d, Tq = 3, H * hd, T_
    q, k, v = (bfr(rnd(B * Tq, d, seed=250 + i, scale=0.8)) for i in range(3))
    lengths = torch.tensor([Tq, max(1, Tq - 13), max(1, Tq // 2)], dtype=torch.int32)
    pos = bfr(rnd(2 * Tq - 1, d, seed=255, scale=0.8)) if rel else None
    u, vb = (0.2 * rnd(H, hd, seed=256), 0.2 * rnd(H, hd, seed=257)) if rel else (None, None)
    dctx = bfr(rnd(B * Tq, d, seed=258))
    scale = 1.0 / math.sqrt(hd)
    qh = q.view(B, Tq, H, hd).permute(2, 0, 1, 3)             # (H, B, T, hd)
    kh, vh = (t.view(B, Tq, H, hd).permute(2, 0, 1, 3) for t in (k, v))
    if rel:
        qu, qv = bfr(qh + u[:, None, None]), bfr(qh + vb[:, None, None])
        ph = pos.view(2 * Tq - 1, H, hd).permute(1, 0, 2)     # (H, P, hd)
        ac = (qu @ kh.transpose(-1, -2)).requires_grad_(True)
        bd = torch.einsum("hbtc,hpc->hbtp", qv, ph).requires_grad_(True)
        idx = (Tq - 1) - torch.arange(Tq)[:, None] + torch.arange(Tq)[None, :]
        s = ac + torch.gather(bd, 3, idx[None, None].expand(H, B, Tq, Tq))
    else:
        ac = (qh @ kh.transpose(-1, -2)).requires_grad_(True)
        s = ac
    s = s * scale
    dead = (torch.arange(Tq)[None, :] >= lengths[:, None])[None, :, None, :].expand(H, B, Tq, Tq)
    if causal:
        dead = dead | torch.ones(Tq, Tq, dtype=torch.bool).triu(1)[None, None]
    prob = torch.softmax(s.masked_fill(dead, float("-inf")), -1)
    drop = None
    if pdrop > 0:          # attention-probability dropout (e_branchformer.py:132) with the kernels' own counter-based mask, regenerated on the host
        from huggingface_asr_amd import synth
        drop = (pdrop, 4321, 55)
        keep = torch.from_numpy(synth.dropout_keep(drop[1], drop[2], H * B * Tq * Tq, pdrop).reshape(H, B, Tq, Tq)).float()
        prob = prob * keep / (1.0 - pdrop)          # what multiplies V (and what the backward kernel leaves as `prob`)
    ctx = prob @ vh
    ctx.backward(dctx.view(B, Tq, H, hd).permute(2, 0, 1, 3))
    ds_want = ac.grad / 1.0                                    # d loss / d(score before the scale) = P (dP - delta) scale
    # the query gradient and, with relative positions, its two terms' column sums (the pos_bias_u / pos_bias_v gradients)
    dqu_want = ac.grad @ kh                                    # (H, B, T, hd)
    dq_want = dqu_want
    if rel:
        dqv_want = torch.einsum("hbtp,hpc->hbtc", bd.grad, ph)
        dq_want = dqu_want + dqv_want
    qkv = torch.cat([q, k, v], 1).to(DEV, BF)
    kw = dict(pos=None if pos is None else dev16(pos), bias_u=None if u is None else u.to(DEV), bias_v=None if vb is None else vb.to(DEV),
              lengths=lengths.to(DEV), causal=causal)
    lse = torch.empty((B, H, Tq), device=DEV, dtype=torch.float32)
    ctx_k = ops.attention_qkv(qkv, B, Tq, H, lse=lse, drop=drop, **kw)
    close(ctx_k.view(B, Tq, H, hd).permute(2, 0, 1, 3), ctx.detach(), floor=1e-2, what="context (lse form)")
    lse_want = torch.logsumexp(s.detach().masked_fill(dead, float("-inf")), -1).permute(1, 0, 2) / math.log(2.0)
    assert (lse.cpu() - lse_want).abs().max() < 2e-2, "log-sum-exp (log2 domain)"
    # poisoned outputs: every element must be written
    Ts = (Tq + 31) // 32 * 32
    pad, Ps = T.band_geometry(Tq)
    assert (Tq - 32 + pad) % 32 == 0 and Ps % 32 == 0 and Ps >= pad + 2 * Tq - 1
    dqk = torch.full((B * Tq, 3 * d), float("nan"), device=DEV, dtype=BF)[:, :d]          # a strided row view, as the trainer's dqkv[:, :d]
    pk, dsk, dbdk, su, sv = T.attn_bwd_probs(qkv, B, Tq, H, ctx_k, dev16(dctx), lse, dqk, drop=drop, **kw)
    assert pk.shape == (H, B, Tq, Ts) and (dbdk is None) == (not rel)
    close(dqk.view(B, Tq, H, hd).permute(2, 0, 1, 3), dq_want, floor=2e-2, what="dQ")
    if rel:
        close(su.sum(0).view(H, hd), dqu_want.sum((1, 2)), rel=2e-2, floor=2e-2, what="sum dQu (pos_bias_u gradient)")
        close(sv.sum(0).view(H, hd), dqv_want.sum((1, 2)), rel=2e-2, floor=2e-2, what="sum dQv (pos_bias_v gradient)")
    close(pk[..., :Tq], prob.detach(), floor=4e-3, what="P")
    close(dsk[..., :Tq], ds_want, floor=2e-2, what="dS")
    assert float(pk[..., Tq:].float().abs().max() if Ts > Tq else 0.0) == 0.0 and float(dsk[..., Tq:].float().abs().max() if Ts > Tq else 0.0) == 0.0
    assert not bool(pk.float().isnan().any()) and not bool(dsk.float().isnan().any())
    if rel:
        assert dbdk.shape == (H, B, Tq, Ps)
        close(dbdk[..., pad:pad + 2 * Tq - 1], bd.grad, floor=2e-2, what="dBD")
        outside = torch.cat([dbdk[..., :pad], dbdk[..., pad + 2 * Tq - 1:]], -1).float()
        assert float(outside.abs().max()) == 0.0, "columns outside the relative positions must be zero"
        # the un-shift is exact: dBD[i][T-1-i+j + pad] is the very bf16 value of dS[i][j]
        idx = ((Tq - 1) - torch.arange(Tq)[:, None] + torch.arange(Tq)[None, :] + pad).to(DEV)
        assert torch.equal(torch.gather(dbdk, 3, idx[None, None].expand(H, B, Tq, Tq)), dsk[..., :Tq])
    # second call on poisoned buffers gives the same bits (nothing depends on what the outputs held)
    dqk2 = torch.empty_like(dqk)
    pk2, dsk2, dbdk2, su2, sv2 = T.attn_bwd_probs(qkv, B, Tq, H, ctx_k, dev16(dctx), lse, dqk2, drop=drop, **kw)
    assert torch.equal(pk, pk2) and torch.equal(dsk, dsk2) and (not rel or torch.equal(dbdk, dbdk2)) and torch.equal(dqk, dqk2)
    assert not rel or (torch.equal(su, su2) and torch.equal(sv, sv2))


def test_csgu_and_merge_dwconv_bwd():
    ops, T = _o()
    B, Tt, Cc, K = 3, 150, 128, 31
    M = B * Tt
    u = bfr(rnd(M, 2 * Cc, seed=1))
    g, be = 1 + 0.1 * rnd(Cc, seed=2), 0.1 * rnd(Cc, seed=3)
    w, bias = rnd(Cc, K, seed=4, scale=0.2), 0.1 * rnd(Cc, seed=5)
    ds = bfr(rnd(M, Cc, seed=6))
    ur, gr, ber, wr, br = [t.clone().requires_grad_(True) for t in (u, g, be, w, bias)]
    r_, g_ = ur[:, :Cc], ur[:, Cc:]
    gn = F.layer_norm(g_, (Cc,), gr, ber, 1e-5)
    gn.retain_grad()
    conv = R.dwconv1d(gn.view(B, Tt, Cc), wr.view(Cc, 1, K), br).reshape(M, Cc)
    (r_ * conv).backward(ds)
    ud = dev16(u)
    st = ops.row_stats(ud[:, Cc:])
    dr = torch.empty(M, Cc, device=DEV, dtype=BF); dgn = torch.empty(M, Cc, device=DEV, dtype=BF)
    dw = torch.zeros(Cc, K, device=DEV); db = torch.zeros(Cc, device=DEV)
    T.csgu_bwd(ud, st, g.to(DEV), be.to(DEV), w.to(DEV), bias.to(DEV), dev16(ds), dr, dgn, dw, db, B, Tt)
    close(dr, ur.grad[:, :Cc], what="dr")
    close(dgn, gn.grad, what="dgn")
    torch.testing.assert_close(dw.cpu(), wr.grad, atol=1e-2 * float(wr.grad.abs().max()), rtol=1e-2)
    torch.testing.assert_close(db.cpu(), br.grad, atol=1e-2 * float(br.grad.abs().max()), rtol=1e-2)
    # merge block: y = m + dwconv(m) + b
    m = bfr(rnd(M, Cc, seed=7))
    mr, wr2, br2 = m.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    dy = bfr(rnd(M, Cc, seed=8))
    (mr + R.dwconv1d(mr.view(B, Tt, Cc), wr2.view(Cc, 1, K), br2).reshape(M, Cc)).backward(dy)
    dm = torch.empty(M, Cc, device=DEV, dtype=BF)
    dw2 = torch.zeros(Cc, K, device=DEV); db2 = torch.zeros(Cc, device=DEV)
    T.dwconv_residual_bwd(dev16(m), w.to(DEV), dev16(dy), dm, dw2, db2, B, Tt)
    close(dm, mr.grad, what="dm")
    torch.testing.assert_close(dw2.cpu(), wr2.grad, atol=1e-2 * float(wr2.grad.abs().max()), rtol=1e-2)
    torch.testing.assert_close(db2.cpu(), br2.grad, atol=1e-2 * float(br2.grad.abs().max()), rtol=1e-2)


@pytest.mark.parametrize("B,Tt,Cc,K,dil", [(5, 150, 128, 31, 1), (40, 70, 64, 31, 1), (3, 90, 96, 15, 1), (2, 130, 64, 31, 15)])
def test_depthwise_conv_tap_gradient_sums_deferred_with_the_layernorm_reductions(B, Tt, Cc, K, dil):
    """csgu_bwd / dwconv_residual_bwd(..., defer=LnReduceBatch): the cross-utterance sums of the tap / bias gradients leave with the batch's next launch (together with
    LayerNorm entries); same data gradients bit for bit, same tap gradients up to the order of the B-term sums; twice into the same target = accumulated twice."""
    ops, T = _o()
    M = B * Tt
    u = dev16(rnd(M, 2 * Cc, seed=1))
    g, be = (1 + 0.1 * rnd(Cc, seed=2)).to(DEV), (0.1 * rnd(Cc, seed=3)).to(DEV)
    w, bias = rnd(Cc, K, seed=4, scale=0.2).to(DEV), (0.1 * rnd(Cc, seed=5)).to(DEV)
    ds, dy = dev16(rnd(M, Cc, seed=6)), dev16(rnd(M, Cc, seed=8))
    st = ops.row_stats(u[:, Cc:])
    pl = (K - 1) * dil if dil > 1 else None
    Z = lambda *s: torch.zeros(*s, device=DEV)
    E = lambda: torch.empty(M, Cc, device=DEV, dtype=BF)
    red = T.LnReduceBatch(DEV)
    # immediate
    dr0, dgn0, dm0 = E(), E(), E()
    dw0, db0, dw0m, db0m = Z(Cc, K), Z(Cc), Z(Cc, K), Z(Cc)
    for _ in range(2):
        T.csgu_bwd(u, st, g, be, w, bias, ds, dr0, dgn0, dw0, db0, B, Tt, pad_left=pl, dilation=dil)
    if dil == 1:
        T.dwconv_residual_bwd(u[:, :Cc], w, dy, dm0, dw0m, db0m, B, Tt)
    # deferred, with a LayerNorm entry in the same batch
    dr1, dgn1, dm1 = E(), E(), E()
    dw1, db1, dw1m, db1m = Z(Cc, K), Z(Cc), Z(Cc, K), Z(Cc)
    dgl, dbl, dgl0, dbl0 = Z(Cc), Z(Cc), Z(Cc), Z(Cc)
    x = rnd(M, Cc, seed=9).to(DEV)
    T.csgu_bwd(u, st, g, be, w, bias, ds, dr1, dgn1, dw1, db1, B, Tt, pad_left=pl, dilation=dil, defer=red)
    T.layernorm_bwd(x, g, dy, torch.empty(M, Cc, device=DEV), accumulate=False, dgamma=dgl, dbeta=dbl, defer=red)
    if dil == 1:
        T.dwconv_residual_bwd(u[:, :Cc], w, dy, dm1, dw1m, db1m, B, Tt, defer=red)
    T.csgu_bwd(u, st, g, be, w, bias, ds, dr1, dgn1, dw1, db1, B, Tt, pad_left=pl, dilation=dil, defer=red)      # same target again: flushes the pending entry first
    red.flush()
    T.layernorm_bwd(x, g, dy, torch.empty(M, Cc, device=DEV), accumulate=False, dgamma=dgl0, dbeta=dbl0)
    assert torch.equal(dr1, dr0) and torch.equal(dgn1, dgn0)
    tol = lambda t: dict(rtol=1e-5, atol=2e-6 * float(t.abs().max()))
    torch.testing.assert_close(dw1, dw0, **tol(dw0)); torch.testing.assert_close(db1, db0, **tol(db0))
    torch.testing.assert_close(dgl, dgl0, **tol(dgl0)); torch.testing.assert_close(dbl, dbl0, **tol(dbl0))
    if dil == 1:
        assert torch.equal(dm1, dm0)
        torch.testing.assert_close(dw1m, dw0m, **tol(dw0m)); torch.testing.assert_close(db1m, db0m, **tol(db0m))


@pytest.mark.parametrize("B,Tt,Cc", [(2, 100, 128), (3, 77, 96), (1, 520, 64)])
def test_causal_dilated_csgu_bwd(B, Tt, Cc):
    """the streaming encoder's CSGU: CausalConv1d with (K-1)//2 = 15 in its dilation slot (e_branchformer.py:153-160; left pad 450) — conv_bwd.hip's dilated kernel
    against autograd of the oracle's restatement, incl. a channel count that is not a multiple of the 64-channel block and T above the 450-frame reach"""
    ops, T = _o()
    K = 31
    dil = (K - 1) // 2
    M = B * Tt
    u = bfr(rnd(M, 2 * Cc, seed=11))
    g, be = 1 + 0.1 * rnd(Cc, seed=12), 0.1 * rnd(Cc, seed=13)
    w, bias = rnd(Cc, K, seed=14, scale=0.2), 0.1 * rnd(Cc, seed=15)
    ds = bfr(rnd(M, Cc, seed=16))
    ur, gr, ber, wr, br = [t.clone().requires_grad_(True) for t in (u, g, be, w, bias)]
    gn = F.layer_norm(ur[:, Cc:], (Cc,), gr, ber, 1e-5)
    gn.retain_grad()
    conv = R.dwconv1d(gn.view(B, Tt, Cc), wr.view(Cc, 1, K), br, True, dil).reshape(M, Cc)
    (ur[:, :Cc] * conv).backward(ds)
    ud = dev16(u)
    fwd = ops.csgu(ud, g.to(DEV), be.to(DEV), w.to(DEV), bias.to(DEV), B, Tt, pad_left=(K - 1) * dil, dilation=dil)
    close(fwd, (ur[:, :Cc] * conv).detach(), what="forward")
    st = ops.row_stats(ud[:, Cc:])
    dr = torch.empty(M, Cc, device=DEV, dtype=BF); dgn = torch.empty(M, Cc, device=DEV, dtype=BF)
    dw = torch.zeros(Cc, K, device=DEV); db = torch.zeros(Cc, device=DEV)
    T.csgu_bwd(ud, st, g.to(DEV), be.to(DEV), w.to(DEV), bias.to(DEV), dev16(ds), dr, dgn, dw, db, B, Tt, pad_left=(K - 1) * dil, dilation=dil)
    close(dr, ur.grad[:, :Cc], what="dr")
    close(dgn, gn.grad, what="dgn")
    torch.testing.assert_close(dw.cpu(), wr.grad, atol=1e-2 * float(wr.grad.abs().max()), rtol=1e-2)
    torch.testing.assert_close(db.cpu(), br.grad, atol=1e-2 * float(br.grad.abs().max()), rtol=1e-2)
    seen = (K - 1 - torch.arange(K)) * dil < Tt                     # taps whose reach stays inside the utterance; the others never meet data
    assert float(dw.cpu()[:, ~seen].abs().max() if (~seen).any() else 0.0) == 0.0


def test_layernorm_bwd_deferred_reductions_match_the_immediate_form():
    """ops_train.LnReduceBatch: the (dgamma | dbeta) reductions of several LayerNorm backward passes in one launch (d 512 and 1024 mixed, a repeated target, more
    entries than the batch holds) against the immediate form"""
    ops, T = _o()
    M = 1000
    batch = T.LnReduceBatch(DEV)
    want, got = [], []
    cases = [(512, 1), (1024, 2), (512, 3), (64, 4)] * 7                       # 28 entries > 24 slots: an automatic flush in between
    for d, seed in cases:
        x = rnd(M, d, seed=seed).to(DEV); dy = dev16(rnd(M, d, seed=seed + 50)); g = (1 + 0.1 * rnd(d, seed=seed + 90)).to(DEV)
        dg1, db1 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
        dg2, db2 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
        dx1, dx2 = torch.zeros(M, d, device=DEV), torch.zeros(M, d, device=DEV)
        T.layernorm_bwd(x, g, dy, dx1, accumulate=False, dgamma=dg1, dbeta=db1)
        T.layernorm_bwd(x, g, dy, dx2, accumulate=False, dgamma=dg2, dbeta=db2, defer=batch)
        T.layernorm_bwd(x, g, dy, dx1, accumulate=True, dgamma=dg1, dbeta=db1)                     # the same targets again: the deferred form must not race on them
        T.layernorm_bwd(x, g, dy, dx2, accumulate=True, dgamma=dg2, dbeta=db2, defer=batch)
        want.append((dx1, dg1, db1)); got.append((dx2, dg2, db2))
    batch.flush()
    for (a, b, c), (d2, e, f) in zip(want, got):
        assert torch.equal(a, d2)
        torch.testing.assert_close(e, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))
        torch.testing.assert_close(f, c, rtol=1e-5, atol=1e-5 * float(c.abs().max()))


@pytest.mark.parametrize("d,pdrop", [(512, 0.0), (512, 0.1), (256, 0.1), (1024, 0.0)])
def test_layernorm_bwd_writes_the_next_operand_itself(d, pdrop):
    """layernorm_bwd(..., cast=(alpha, drop)): the bf16 copy alpha * dropout(dx) from the same pass equals the separate dropout / cast kernel on the finished dx."""
    ops, T = _o()
    M = 777
    x, dy = rnd(M, d, seed=1).to(DEV), dev16(rnd(M, d, seed=2))
    g = (1 + 0.1 * rnd(d, seed=3)).to(DEV)
    base = rnd(M, d, seed=4).to(DEV)
    drop = (pdrop, 31, 977) if pdrop > 0 else None
    red = T.LnReduceBatch(DEV)
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    dx = base.clone()
    _, cast = T.layernorm_bwd(x, g, dy, dx, accumulate=True, dgamma=dg, dbeta=db, defer=red, cast=(0.5, drop))
    red.flush()
    dx2 = base.clone()
    dg2, db2 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    T.layernorm_bwd(x, g, dy, dx2, accumulate=True, dgamma=dg2, dbeta=db2, defer=red)
    red.flush()
    want = T.dropout_(dx2, pdrop, drop[1], drop[2], out=torch.empty((M, d), device=DEV, dtype=BF), alpha=0.5) if drop else T.add_cast(dx2, alpha=0.5)
    assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)
    assert torch.equal(cast, want)
    # frozen affine pair (no deferred reduction): the two-pass form behind the same call
    dx3 = base.clone()
    _, cast3 = T.layernorm_bwd(x, g, dy, dx3, accumulate=True, cast=(0.5, drop))
    assert torch.equal(dx3, dx2) and torch.equal(cast3, want)


@pytest.mark.parametrize("d,pdrop,dy2f32", [(256, 0.0, False), (512, 0.1, False), (320, 0.0, True), (64, 0.1, True)])
def test_layernorm_bwd_of_two_norms_on_the_same_rows_in_one_pass(d, pdrop, dy2f32):
    """layernorm_bwd_dual (the layer's two branch norms read the same x1, e_branchformer.py:273,292): dx against torch autograd of both norms in fp64 and against the
    two single passes; each affine gradient equal to its single pass's bit for bit (same rows per block, same order); the cast operand from the finished dx."""
    ops, T = _o()
    M = 1003
    x = rnd(M, d, seed=1)
    dy1, dy2 = rnd(M, d, seed=2).to(BF), rnd(M, d, seed=3)
    dy2 = dy2 if dy2f32 else dy2.to(BF).float()
    g1, g2 = 1 + 0.1 * rnd(d, seed=4), 1 + 0.1 * rnd(d, seed=5)
    base = rnd(M, d, seed=6)
    xr = x.double().requires_grad_(True)
    ga, gb = g1.double().requires_grad_(True), g2.double().requires_grad_(True)
    za, zb = torch.zeros(d, dtype=torch.float64, requires_grad=True), torch.zeros(d, dtype=torch.float64, requires_grad=True)
    y = (torch.nn.functional.layer_norm(xr, (d,), ga, za) * dy1.double()).sum() + (torch.nn.functional.layer_norm(xr, (d,), gb, zb) * dy2.double()).sum()
    y.backward()
    drop = (pdrop, 17, 4242) if pdrop > 0 else None
    red = T.LnReduceBatch(DEV)
    G = lambda: torch.zeros(d, device=DEV)
    dga, dba, dgb, dbb = G(), G(), G(), G()
    dx = base.to(DEV)
    xd, d1, d2 = x.to(DEV), dy1.to(DEV), (dy2.to(DEV) if dy2f32 else dy2.to(DEV, BF))
    _, cast = T.layernorm_bwd_dual(xd, g1.to(DEV), d1, g2.to(DEV), d2, dx, accumulate=True, dgamma=dga, dbeta=dba, dgamma2=dgb, dbeta2=dbb, defer=red, cast=(0.5, drop))
    red.flush()
    torch.testing.assert_close(dx.cpu().double(), base.double() + xr.grad, rtol=1e-4, atol=2e-5)
    # the two single passes
    dx2 = base.to(DEV)
    sga, sba, sgb, sbb = G(), G(), G(), G()
    T.layernorm_bwd(xd, g1.to(DEV), d1, dx2, accumulate=True, dgamma=sga, dbeta=sba, defer=red)
    T.layernorm_bwd(xd, g2.to(DEV), d2, dx2, accumulate=True, dgamma=sgb, dbeta=sbb, defer=red)
    red.flush()
    torch.testing.assert_close(dx, dx2, rtol=1e-5, atol=2e-6)
    assert torch.equal(dga, sga) and torch.equal(dba, sba) and torch.equal(dgb, sgb) and torch.equal(dbb, sbb)
    torch.testing.assert_close(dga.cpu().double(), ga.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dbb.cpu().double(), zb.grad, rtol=1e-4, atol=1e-4)
    want = T.dropout_(dx.clone(), pdrop, drop[1], drop[2], out=torch.empty((M, d), device=DEV, dtype=BF), alpha=0.5) if drop else T.add_cast(dx, alpha=0.5)
    assert torch.equal(cast, want)
    # without the cast operand, and overwriting
    dx3 = torch.full((M, d), 7.0, device=DEV)
    T.layernorm_bwd_dual(xd, g1.to(DEV), d1, g2.to(DEV), d2, dx3, accumulate=False, dgamma=G(), dbeta=G(), dgamma2=G(), dbeta2=G(), defer=red)
    red.flush()
    torch.testing.assert_close(dx3.cpu().double(), xr.grad, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("k,s,p", [(3, 2, 1), (3, 2, 0), (5, 3, 2)])
def test_subsampled_lengths_in_one_launch(k, s, p):
    """mi_subsampled_lengths_i32 against the reference's length arithmetic (_get_feat_extract_output_lengths: floor division, two conv layers), including inputs shorter
    than the kernel (negative numerators) and the clamp of the padded count to the batch's frame axis"""
    from huggingface_asr_amd import _lib
    fl = torch.tensor([1, 2, 3, 4, 7, 8, 9, 100, 999, 1000, 1001, 1998, 2000, 0], dtype=torch.int32)
    li, lo = fl.clone(), fl.clone()
    for _ in range(2):
        li = torch.div(li + 2 * p - k, s, rounding_mode="floor") + 1
        lo = torch.div(lo - k, s, rounding_mode="floor") + 1
    T2 = 250
    d = fl.to(DEV)
    inner, outer = torch.empty_like(d), torch.empty_like(d)
    _lib.check(_lib.lib().mi_subsampled_lengths_i32(d.data_ptr(), d.numel(), k, s, p, 2, T2, inner.data_ptr(), outer.data_ptr(), torch.cuda.current_stream().cuda_stream), "lengths")
    assert torch.equal(inner.cpu(), torch.clamp(li, max=T2).to(torch.int32)) and torch.equal(outer.cpu(), lo.to(torch.int32))


def test_position_bias_gradient_sums_deferred_with_the_layernorm_reductions():
    """attn_bwd_probs leaves the per-wave column sums of dS K / dBD P as rows of [u | v]; LnReduceBatch.add_rows2 sums them into the pos_bias_u / pos_bias_v gradients with
    its next launch — against colsum2_acc_ on the same rows (same values up to the order of the row sums), the buffer kept alive until the flush"""
    ops, T = _o()
    B, Tq, H, hd = 3, 200, 4, 64
    d = H * hd
    qkv = dev16(rnd(B * Tq, 3 * d, seed=1, scale=0.5))
    Pn = 2 * Tq - 1
    pos = dev16(rnd(Pn, d, seed=2, scale=0.5))
    bu, bv = (0.1 * rnd(H, hd, seed=3)).to(DEV), (0.1 * rnd(H, hd, seed=4)).to(DEV)
    lens = torch.tensor([200, 150, 77], dtype=torch.int32, device=DEV)
    lse = torch.empty((B, H, Tq), device=DEV)
    ctx = ops.attention_qkv(qkv, B, Tq, H, pos=pos, bias_u=bu, bias_v=bv, lengths=lens, lse=lse)
    dctx = dev16(rnd(B * Tq, d, seed=5))
    dq = torch.empty(B * Tq, d, device=DEV, dtype=BF)
    _, _, _, su, sv = T.attn_bwd_probs(qkv, B, Tq, H, ctx, dctx, lse, dq, pos=pos, bias_u=bu, bias_v=bv, lengths=lens)
    assert su.stride(0) == 2 * d and sv.data_ptr() == su.data_ptr() + 4 * d
    # the walk's own (q + u) / (q + v) operands against the pass that used to make them
    dq2 = torch.empty_like(dq)
    *_, qu, qv = T.attn_bwd_probs(qkv, B, Tq, H, ctx, dctx, lse, dq2, pos=pos, bias_u=bu, bias_v=bv, lengths=lens, qb=True)
    wu, wv = T.add_rowvec2(qkv[:, :d], bu, bv)
    assert torch.equal(qu, wu) and torch.equal(qv, wv) and torch.equal(dq2, dq)
    gu0, gv0 = torch.zeros(H, hd, device=DEV), torch.zeros(H, hd, device=DEV)
    T.colsum2_acc_(gu0, gv0, su, sv)
    gu1, gv1 = torch.zeros(H, hd, device=DEV), torch.zeros(H, hd, device=DEV)
    red = T.LnReduceBatch(DEV)
    red.add_rows2(su, gu1, gv1)
    keep_ptr = su.data_ptr()
    del su, sv
    junk = [torch.full((B * 8, 2 * d), 1e9, device=DEV) for _ in range(4)]          # would land in the freed buffer
    assert all(j.data_ptr() != keep_ptr for j in junk)
    red.flush()
    torch.testing.assert_close(gu1, gu0, rtol=1e-5, atol=2e-6 * float(gu0.abs().max()))
    torch.testing.assert_close(gv1, gv0, rtol=1e-5, atol=2e-6 * float(gv0.abs().max()))
    assert float(gu0.abs().max()) > 0 and float(gv0.abs().max()) > 0


def test_scale_by_device_scalar():
    """the autograd bridge's d(loss) factor: a device scalar, no host sync; exactly 1 leaves the buffer untouched, anything else scales it (odd length: the scalar tail)"""
    ops, T = _o()
    a = rnd(4099, seed=21).to(DEV)
    keep = a.clone()
    T.scale_by_device_scalar_(a, torch.ones((), device=DEV))
    assert torch.equal(a, keep)
    T.scale_by_device_scalar_(a, torch.tensor(0.25, device=DEV))
    assert torch.equal(a, keep * 0.25)


def test_conv_frontend_bwd():
    ops, T = _o()
    B, Tt, Fq, C1, K, s, pad = 2, 61, 40, 32, 3, 2, 1
    T1, F1 = (Tt + 2 * pad - K) // s + 1, (Fq + 2 * pad - K) // s + 1
    T2, F2 = (T1 + 2 * pad - K) // s + 1, (F1 + 2 * pad - K) // s + 1
    x = rnd(B, Tt, Fq, seed=1)
    w1, b1 = rnd(C1, 1, K, K, seed=2, scale=0.3), 0.1 * rnd(C1, seed=3)
    w1r, b1r = w1.clone().requires_grad_(True), b1.clone().requires_grad_(True)
    act1 = F.gelu(F.conv2d(x[:, None], w1r, b1r, stride=s, padding=pad))                  # (B,C1,T1,F1)
    col_ref = F.unfold(act1, K, padding=pad, stride=s)                                      # (B, C1*K*K, T2*F2), k = c*K*K + kh*K + kw
    col_ref = col_ref.view(B, C1, K * K, T2 * F2).permute(0, 3, 2, 1).reshape(B * T2 * F2, K * K * C1)   # -> k = (kh*K+kw)*C1 + c
    dcol = bfr(rnd(B * T2 * F2, K * K * C1, seed=4))
    col_ref.backward(dcol)
    # im2col of the channels-last bf16 activation
    a1 = ops.conv2d_first_gelu(x.to(DEV), w1.reshape(C1, K * K).to(DEV), b1.to(DEV), stride=s, pad=pad)
    col = T.im2col(a1, K, s, pad, T2, F2)
    close(col, col_ref.detach(), floor=4e-3, what="im2col")
    dw = torch.zeros(C1, K * K, device=DEV); db = torch.zeros(C1, device=DEV)
    T.conv2d_first_bwd(x.to(DEV), w1.reshape(C1, K * K).to(DEV), b1.to(DEV), dev16(dcol), dw, db, K, s, pad, T1, F1, K, s, pad, T2, F2)
    torch.testing.assert_close(dw.cpu(), w1r.grad.reshape(C1, K * K), atol=5e-3 * float(w1r.grad.abs().max()), rtol=5e-3)
    torch.testing.assert_close(db.cpu(), b1r.grad, atol=5e-3 * float(b1r.grad.abs().max()), rtol=5e-3)


@pytest.mark.parametrize("B,Tt,Fq", [(2, 61, 40), (3, 64, 80), (1, 37, 23)])
def test_conv_frontend_bwd_by_phases(B, Tt, Fq):
    """conv2's input gradient as four stride-1 implicit-GEMM convolutions (mi_conv2d_s2k3_dgrad_bf16: one per parity of the position) + conv1's backward reading the
    phase buffers, against autograd through gelu(conv1) -> conv2 (the reference's front end, extractors.py:82-89): the phase buffers hold d(act1) position for position
    (odd and even T1 / F1: the last row / column of a parity class reads past conv2's output and must see zeros), and dW1 / db1 follow."""
    ops, T = _o()
    C1 = C2 = 256
    K, s, pad = 3, 2, 1
    T1, F1 = (Tt + 2 * pad - K) // s + 1, (Fq + 2 * pad - K) // s + 1
    T2, F2 = (T1 + 2 * pad - K) // s + 1, (F1 + 2 * pad - K) // s + 1
    x = rnd(B, Tt, Fq, seed=1)
    w1, b1 = rnd(C1, 1, K, K, seed=2, scale=0.3), 0.1 * rnd(C1, seed=3)
    w2 = bfr(rnd(C2, C1, K, K, seed=5, scale=0.05))
    dy2 = bfr(rnd(B, T2, F2, C2, seed=4))
    w1r, b1r = w1.clone().requires_grad_(True), b1.clone().requires_grad_(True)
    act1 = F.gelu(F.conv2d(x[:, None], w1r, b1r, stride=s, padding=pad))                  # (B,C1,T1,F1)
    act1.retain_grad()
    F.conv2d(act1, w2, None, stride=s, padding=pad).backward(dy2.permute(0, 3, 1, 2))
    want_dact1 = act1.grad.permute(0, 2, 3, 1)                                              # (B,T1,F1,C1)
    wT2 = dev16(w2.permute(2, 3, 1, 0).reshape(K * K * C1, C2))                             # row (kh, kw, c), column co
    assert T.conv2d_s2k3_dgrad_supported(B, T1, F1, C1, T2, F2, pad) == B * T1 * F1 * C1
    assert T.conv2d_s2k3_dgrad_supported(B, T1, F1, C1, T2, F2, 2) == 0                     # the causal front end's leading pad: stays on the im2col-gradient path
    dw = torch.zeros(C1, K * K, device=DEV); db = torch.zeros(C1, device=DEV)
    phases = T.conv2d_first_bwd_phases(x.to(DEV), w1.reshape(C1, K * K).to(DEV), b1.to(DEV), dev16(dy2.reshape(B * T2 * F2, C2)), wT2, dw, db,
                                       K, s, pad, T1, F1, pad, T2, F2).float().cpu()
    got = torch.zeros(B, T1, F1, C1)
    off = 0
    for pt in (0, 1):
        t1s = [t for t in range(T1) if (t + pad) % 2 == pt]
        for pf in (0, 1):
            f1s = [f for f in range(F1) if (f + pad) % 2 == pf]
            n = B * len(t1s) * len(f1s) * C1
            blk = got[:, t1s]; blk[:, :, f1s] = phases[off:off + n].view(B, len(t1s), len(f1s), C1); got[:, t1s] = blk
            off += n
    assert off == phases.numel()
    close(got, want_dact1, floor=4e-3, what="d(act1) from the phase buffers")
    torch.testing.assert_close(dw.cpu(), w1r.grad.reshape(C1, K * K), atol=5e-3 * float(w1r.grad.abs().max()), rtol=5e-3)
    torch.testing.assert_close(db.cpu(), b1r.grad, atol=5e-3 * float(b1r.grad.abs().max()), rtol=5e-3)


@pytest.mark.parametrize("reduction", ["mean", "sum"])
def test_ctc_loss_bwd(reduction):
    ops, T = _o()
    B, Tt, V1, U = 4, 60, 51, 9
    logits = rnd(B, Tt, V1, seed=1, scale=1.5)
    labels = torch.randint(0, V1 - 1, (B, U), generator=torch.Generator().manual_seed(2))
    labels[1, 6:] = -100; labels[2, 1] = labels[2, 0]; labels[3, :] = labels[3, 0]          # padding, repeats
    in_len = torch.tensor([60, 47, 33, 12], dtype=torch.int32)                              # utterance 3: 9 repeats need 17 frames > 12: infeasible
    lg = logits.clone().requires_grad_(True)
    lp = torch.log_softmax(lg, -1).transpose(0, 1)
    tl = (labels >= 0).sum(-1)
    flat = labels[labels >= 0]
    loss = F.ctc_loss(lp, flat, in_len.long(), tl, blank=V1 - 1, reduction=reduction, zero_infinity=True)
    (0.3 * loss).backward()
    ld = logits.to(DEV)
    lse = ops.row_lse(ld.reshape(B * Tt, V1))
    got_loss, nll, _ = ops.ctc_loss(ld, labels.to(DEV), in_len.to(DEV), reduction=reduction, zero_infinity=True, lse=lse)
    torch.testing.assert_close(got_loss.cpu(), loss.detach(), atol=1e-3, rtol=1e-4)
    dl = T.ctc_loss_bwd(ld, lse, labels.to(DEV), in_len.to(DEV), nll, reduction=reduction, gscale=0.3)
    assert dl.shape == (B * Tt, 56)
    assert float(dl[:, V1:].float().abs().max()) == 0.0
    close(dl[:, :V1].reshape(B, Tt, V1), lg.grad, floor=5e-3, what="ctc dlogits")


@pytest.mark.parametrize("B,Tq,Tk,H,hd,causal,p", [(3, 60, 500, 4, 64, False, 0.1), (3, 60, 60, 4, 64, True, 0.1), (2, 37, 250, 2, 128, False, 0.0), (2, 33, 33, 2, 64, True, 0.0),
                                                    (2, 150, 97, 4, 64, False, 0.1)])
def test_fused_attention_for_separate_operands_against_the_materialised_path(B, Tq, Tk, H, hd, causal, p):
    """mi_attention_x_lse_bf16 / mi_attention_x_bwd_probs (round 4: the GPT-2 decoder's self- and cross-attention in training, Tq != Tk, ragged key lengths, probability dropout)
    against the path they replace — batched GEMM, generic soft-max (same counter-based mask), batched GEMM (train_aed.attention_fwd_plain / attention_bwd_plain): context,
    probabilities, score gradients and dq / dk / dv."""
    ops, T = _o()
    from huggingface_asr_amd import train_aed as TA
    d = H * hd
    q = dev16(rnd(B * Tq, d, seed=1, scale=0.7)); k = dev16(rnd(B * Tk, d, seed=2, scale=0.7)); v = dev16(rnd(B * Tk, d, seed=3, scale=0.7))
    dctx = dev16(rnd(B * Tq, d, seed=4, scale=0.5))
    lengths = None if causal else torch.tensor([max(1, Tk - 41 * b) for b in range(B)], dtype=torch.int32, device=DEV)
    drop = (p, 77, 5) if p > 0 else None
    ctx0, prob0, pdrop0 = TA.attention_fwd_plain(q, k, v, B, Tq, Tk, H, lengths=lengths, causal=causal, drop=drop)
    dq0, dk0, dv0 = (torch.zeros_like(t) for t in (q, k, v))
    TA.attention_bwd_plain(q, k, v, dctx, dq0, dk0, dv0, B, Tq, Tk, H, lengths=lengths, causal=causal, drop=drop, saved=(prob0, pdrop0))
    ctx1, lse = T.attention_x_lse(q, k, v, B, Tq, Tk, H, lengths=lengths, causal=causal, drop=drop)
    close(ctx1, ctx0.float().cpu(), floor=6e-3, what="context")
    dq1, dk1, dv1 = (torch.zeros_like(t) for t in (q, k, v))
    TA.attention_bwd_fused(q, k, v, ctx1, dctx, lse, dq1, dk1, dv1, B, Tq, Tk, H, lengths=lengths, causal=causal, drop=drop)
    close(dv1, dv0.float().cpu(), floor=8e-3, what="dv")
    close(dk1, dk0.float().cpu(), floor=8e-3, what="dk")
    close(dq1, dq0.float().cpu(), floor=8e-3, what="dq")
    # bit-reproducible
    dq2, dk2, dv2 = (torch.zeros_like(t) for t in (q, k, v))
    TA.attention_bwd_fused(q, k, v, ctx1, dctx, lse, dq2, dk2, dv2, B, Tq, Tk, H, lengths=lengths, causal=causal, drop=drop)
    assert torch.equal(dq1, dq2) and torch.equal(dk1, dk2) and torch.equal(dv1, dv2)


def test_bgemm_dead_row_tiles_are_stored_as_zeros():
    """mi_bgemm_sparse_bf16's m_valid: the tiles of rows at or beyond an entry's count are written as zeros without reading A (filled with NaN there: a multiplied tile
    would show it) — the dV / dK products at ragged key lengths; everything else equals the plain call bit for bit."""
    ops, T = _o()
    H, B, Tk, Tq, hd = 2, 3, 300, 70, 64
    d = H * hd
    Ts = (Tk + 31) // 32 * 32
    g = torch.Generator().manual_seed(9)
    lens = torch.tensor([300, 129, 64], dtype=torch.int32)
    prob = (torch.rand(H, B, Tq, Ts, generator=g)).to(torch.bfloat16)
    for b in range(B):
        prob[:, b, :, int(lens[b]):] = 0
    dctx = (torch.randn(B * Tq, d, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    sS = (B * Tq * Ts, Tq * Ts)
    pz = prob.to(DEV)
    want = torch.empty((B * Tk, d), device=DEV, dtype=torch.bfloat16)
    T.bgemm(pz, (*sS, 1, Ts), dctx, (hd, Tq * d, 1, d), want, (hd, Tk * d, d), H, B, Tk, hd, Tq)
    pn = prob.clone()
    for b in range(B):
        dead0 = (int(lens[b]) + 63) // 64 * 64                # first row of the first dead 64-row tile
        pn[:, b, :, dead0:] = float("nan")
    got = torch.full((B * Tk, d), 7.0, device=DEV, dtype=torch.bfloat16)
    T.bgemm(pn.to(DEV), (*sS, 1, Ts), dctx, (hd, Tq * d, 1, d), got, (hd, Tk * d, d), H, B, Tk, hd, Tq, m_valid=lens.to(DEV))
    assert torch.equal(got, want) and not bool(torch.isnan(got.float()).any())


@pytest.mark.parametrize("Tt,cg,hd", [(500, 4, 64), (250, 4, 128), (97, 2, 64), (33, 1, 64)])
def test_bgemm_band_skips_only_zero_tiles(Tt, cg, hd):
    """mi_bgemm_band_bf16 (round 4): the d(positions) product dBD^T (q + v) with the K loop limited, per tile of relative positions, to the query rows whose band of dBD
    can reach it — against the same product over the whole K range: identical bits (the skipped tiles are all zero), on a dBD with the real band geometry."""
    ops, T = _o()
    H, G = 2, 3                                                  # heads, utterance groups
    off, Kp = T.band_geometry(Tt)
    Ps = (off + 2 * Tt - 1 + 31) // 32 * 32
    d = H * hd
    g = torch.Generator().manual_seed(3)
    ds = (torch.randn(H, G * cg, Tt, Tt, generator=g) * 0.3).to(torch.bfloat16)
    dbd = torch.zeros(H, G * cg, Tt, Ps, dtype=torch.bfloat16)
    for i in range(Tt):
        dbd[:, :, i, Tt - 1 - i + off: 2 * Tt - 1 - i + off] = ds[:, :, i, :]
    qv = (torch.randn(G * cg * Tt, d, generator=g) * 0.5).to(torch.bfloat16)
    dbd_d, qv_d = dbd.to(DEV), qv.to(DEV)
    B = G * cg
    outs = []
    for band in (None, (Tt, Tt - 1 + off, cg)):
        dpp = torch.zeros((G, Kp * d), device=DEV, dtype=torch.float32)
        T.bgemm(dbd_d, (B * Tt * Ps, cg * Tt * Ps, 1, Ps), qv_d, (hd, cg * Tt * d, 1, d), dpp, (hd, Kp * d, d), H, G, Kp, hd, cg * Tt, band=band)
        outs.append(dpp)
    assert torch.equal(outs[0], outs[1])
    want = torch.einsum("hbip,bihc->bphc", dbd.float().view(H, G, cg * Tt, Ps)[..., :Kp].reshape(H, G, cg * Tt, Kp), qv.float().view(G, cg * Tt, H, hd)).reshape(G, Kp * d)
    close(outs[1], want, floor=4e-3, what="d(positions) partials")


@pytest.mark.parametrize("Tt,U", [(500, 60), (300, 20), (257, 63)])
def test_ctc_loss_bwd_long_inputs(Tt, U):
    """more than 256 frames with <= 128 states (BASELINE config 3: 20 s clips = 500 encoder frames, 60 labels): the wave form with the emissions in an L2-resident global
    table (round 4; before: the block form, alpha then beta with a barrier per frame) against torch's ctc_loss backward — ragged input lengths down to 1 frame past the
    256 boundary, repeated labels, a padded target, an infeasible utterance."""
    ops, T = _o()
    B, V1 = 5, 71
    logits = rnd(B, Tt, V1, seed=11, scale=1.5)
    labels = torch.randint(0, V1 - 1, (B, U), generator=torch.Generator().manual_seed(12))
    labels[1, U // 2:] = -100; labels[2, 1] = labels[2, 0]; labels[2, 5] = labels[2, 4]; labels[3, :] = labels[3, 0]
    in_len = torch.tensor([Tt, Tt - 37, 257, min(Tt, 2 * U - 2), 1], dtype=torch.int32)        # utterance 3: U repeats need 2U - 1 frames: infeasible; utterance 4: one frame
    labels[4, 1:] = -100
    lg = logits.clone().requires_grad_(True)
    lp = torch.log_softmax(lg, -1).transpose(0, 1)
    tl = (labels >= 0).sum(-1)
    flat = labels[labels >= 0]
    loss = F.ctc_loss(lp, flat, in_len.long(), tl, blank=V1 - 1, reduction="mean", zero_infinity=True)
    (0.3 * loss).backward()
    ld = logits.to(DEV)
    lse = ops.row_lse(ld.reshape(B * Tt, V1))
    got_loss, nll, _ = ops.ctc_loss(ld, labels.to(DEV), in_len.to(DEV), reduction="mean", zero_infinity=True, lse=lse)
    torch.testing.assert_close(got_loss.cpu(), loss.detach(), atol=1e-3, rtol=1e-4)
    dl = T.ctc_loss_bwd(ld, lse, labels.to(DEV), in_len.to(DEV), nll, reduction="mean", gscale=0.3)
    close(dl[:, :V1].reshape(B, Tt, V1), lg.grad, floor=5e-3, what="ctc dlogits, long inputs")
    assert torch.equal(dl, T.ctc_loss_bwd(ld, lse, labels.to(DEV), in_len.to(DEV), nll, reduction="mean", gscale=0.3))          # bit-reproducible


@pytest.mark.parametrize("Tt,U,red", [(120, 20, "mean"), (250, 40, "mean"), (500, 60, "sum"), (300, 33, "mean")])
def test_ctc_loss_and_gradient_from_one_pair_of_recursions(Tt, U, red):
    """mi_ctc_loss_bwd_nll (the training step: no forward loss kernel): loss, per-utterance nll and dlogits against torch's ctc_loss + backward, and against the two-call
    form (the forward kernel's nll comes from its bidirectional recursion: equal to 1e-5) — ragged lengths, repeats, a padded target, an infeasible utterance
    (nll = inf, zero gradient, counted as 0 with zero_infinity), a one-frame utterance; a target of more than 63 labels is refused (None: callers run the two calls)."""
    ops, T = _o()
    B, V1 = 5, 71
    logits = rnd(B, Tt, V1, seed=21, scale=1.5)
    labels = torch.randint(0, V1 - 1, (B, U), generator=torch.Generator().manual_seed(22))
    labels[1, U // 2:] = -100; labels[2, 1] = labels[2, 0]; labels[2, 5] = labels[2, 4]; labels[3, :] = labels[3, 0]
    in_len = torch.tensor([Tt, Tt - 37, max(Tt // 2, 2 * U + 1), min(Tt, 2 * U - 2), 1], dtype=torch.int32)
    labels[4, 1:] = -100
    lg = logits.clone().requires_grad_(True)
    lp = torch.log_softmax(lg, -1).transpose(0, 1)
    tl = (labels >= 0).sum(-1)
    loss = F.ctc_loss(lp, labels[labels >= 0], in_len.long(), tl, blank=V1 - 1, reduction=red, zero_infinity=True)
    (0.3 * loss).backward()
    ld, lab, il = logits.to(DEV), labels.to(DEV), in_len.to(DEV)
    lse = ops.row_lse(ld.reshape(B * Tt, V1))
    dl, got_loss, nll = T.ctc_loss_bwd_nll(ld, lse, lab, il, reduction=red, zero_infinity=True, gscale=0.3)
    torch.testing.assert_close(got_loss.cpu(), loss.detach(), atol=1e-3, rtol=1e-4)
    close(dl[:, :V1].reshape(B, Tt, V1), lg.grad, floor=5e-3, what="ctc dlogits, own nll")
    l2, nll2, _ = ops.ctc_loss(ld, lab, il, reduction=red, zero_infinity=True, lse=lse)
    assert torch.isinf(nll[3]) and torch.isinf(nll2[3])
    fin = torch.isfinite(nll2)
    torch.testing.assert_close(nll[fin], nll2[fin], rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(got_loss, l2, rtol=1e-5, atol=1e-5)
    dl2 = T.ctc_loss_bwd(ld, lse, lab, il, nll2, reduction=red, gscale=0.3)
    close(dl[:, :V1].float(), dl2[:, :V1].float(), floor=5e-3, what="own nll vs given nll")
    assert float(dl[3 * Tt:4 * Tt].abs().max()) == 0.0                       # the infeasible utterance
    r = T.ctc_loss_bwd_nll(ld, lse, lab, il, reduction=red, zero_infinity=True, gscale=0.3)
    assert torch.equal(r[0], dl) and torch.equal(r[2], nll)                   # bit-reproducible
    long_lab = torch.randint(0, V1 - 1, (B, 70), generator=torch.Generator().manual_seed(3)).to(DEV)
    assert T.ctc_loss_bwd_nll(ld, lse, long_lab, il, reduction=red, zero_infinity=True) is None


def test_ce_and_embed_bwd():
    ops, T = _o()
    B, U, V, d = 3, 11, 50, 64
    logits = rnd(B, U, V, seed=1, scale=2.0)
    labels = torch.randint(0, V, (B, U), generator=torch.Generator().manual_seed(2))
    labels[1, 7:] = -100
    lg = logits.clone().requires_grad_(True)
    loss = F.cross_entropy(lg[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), label_smoothing=0.1, ignore_index=-100)
    (0.6 * loss).backward()
    ld = logits.to(DEV)
    acc = torch.zeros(2, device=DEV)
    rc_loss = ops.ce_label_smoothing(ld, labels.to(DEV), shift=1, eps=0.1)
    torch.testing.assert_close(rc_loss.cpu(), loss.detach(), atol=1e-4, rtol=1e-4)
    # the backward kernel reads the [sum, count] pair of the forward
    from huggingface_asr_amd import _lib
    rows = torch.empty(B * (U - 1), device=DEV)
    _lib.check(_lib.lib().mi_ce_label_smoothing(ld.data_ptr(), ld.stride(1), labels.to(DEV).data_ptr(), B, U, 1, V, 0.1, acc.data_ptr(), rows.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), "ce")
    dl = T.ce_label_smoothing_bwd(ld, labels.to(DEV), acc, shift=1, eps=0.1, weight=0.6)
    close(dl[:, :V].reshape(B, U, V), lg.grad, floor=5e-3, what="ce dlogits")
    # embeddings
    ids = torch.randint(0, V, (B, U), generator=torch.Generator().manual_seed(3))
    wte, wpe = rnd(V, d, seed=4).requires_grad_(True), rnd(32, d, seed=5).requires_grad_(True)
    dx = rnd(B * U, d, seed=6)
    (wte[ids] * 2.0 + wpe[torch.arange(U)][None]).reshape(B * U, d).backward(dx)
    dwte, dwpe = torch.zeros(V, d, device=DEV), torch.zeros(32, d, device=DEV)
    T.embed_tokens_bwd(ids.to(DEV), dx.to(DEV), dwte, dwpe, scale=2.0)
    torch.testing.assert_close(dwte.cpu(), wte.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(dwpe.cpu(), wpe.grad, atol=1e-4, rtol=1e-4)


def test_adamw_matches_torch():
    ops, T = _o()
    n = 5000
    p0, g1, g2 = rnd(n, seed=1), rnd(n, seed=2, scale=3.0), rnd(n, seed=3, scale=0.1)
    decay = (torch.arange(n) % 3 != 0)
    pa, pb = p0[decay].clone().requires_grad_(True), p0[~decay].clone().requires_grad_(True)
    opt = torch.optim.AdamW([{"params": [pa], "weight_decay": 0.01}, {"params": [pb], "weight_decay": 0.0}], lr=2e-3, betas=(0.9, 0.98), eps=1e-8)
    p = p0.clone().to(DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    mirror = torch.empty(n, device=DEV, dtype=BF)
    dmask = decay.to(DEV, torch.uint8)
    for step, g in enumerate((g1, g2), start=1):
        full = torch.cat([pa.detach(), pb.detach()])
        gg = g.clone()
        norm = gg.norm()
        coef = min(1.0, 1.0 / (float(norm) + 1e-6))
        pa.grad, pb.grad = (gg[decay] * coef), (gg[~decay] * coef)
        opt.step()
        ss = torch.zeros(1, device=DEV); nc = torch.zeros(3, device=DEV)
        gd = g.to(DEV)
        T.sumsq_(ss, gd); T.clip_coef(ss, 1.0, nc)
        torch.testing.assert_close(nc.cpu(), torch.tensor([float(norm), coef, 0.0]), atol=1e-4, rtol=1e-4)
        T.adamw_step_(p, gd, m, v, dmask, lr=2e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01, step=step, norm_coef=nc, mirror=mirror)
    want = torch.empty(n); want[decay] = pa.detach(); want[~decay] = pb.detach()
    torch.testing.assert_close(p.cpu(), want, atol=1e-6, rtol=1e-5)
    assert torch.equal(mirror.cpu(), p.cpu().to(BF))
    # a step whose norm is above the skip threshold (GradAwareTrainer, training_utils.py:81,101-115) or not finite leaves everything untouched
    p0, m0, v0 = p.clone(), m.clone(), v.clone()
    for bad, thr in ((gd * 1e4, 100.0), (torch.full_like(gd, float("nan")), 0.0)):
        ss.zero_(); T.sumsq_(ss, bad); T.clip_coef(ss, 1.0, nc, skip_above=thr)
        assert float(nc[2]) == 1.0 and float(nc[1]) == 0.0
        T.adamw_step_(p, bad, m, v, dmask, lr=2e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01, step=9, norm_coef=nc, mirror=mirror)
        assert torch.equal(p, p0) and torch.equal(m, m0) and torch.equal(v, v0)



def test_layer_mixing_ops_match_torch():
    """the four entries behind the fine-tuning head's layer mixing (bestrq.py:239-245): softmax of the weights, device-coefficient axpy,
    deterministic dot product, softmax backward — against torch fp32/fp64 on the same inputs."""
    _, T = _o()
    torch.manual_seed(5)
    n, L1 = 2 * 50 * 64 + 3, 13                      # odd length: the scalar tail of the 16-B loops runs too
    w = torch.randn(L1, device=DEV)
    s = T.softmax_vec(w)
    assert torch.allclose(s, torch.softmax(w, 0), rtol=1e-6, atol=1e-7) and abs(float(s.sum()) - 1.0) < 1e-6
    hs = [torch.randn(n, device=DEV) for _ in range(L1)]
    mixed = torch.empty(n, device=DEV)
    for i, h in enumerate(hs):
        T.axpy_dev_(mixed, h, s[i:i + 1], overwrite=(i == 0))
    want = (torch.stack(hs).double() * s.double()[:, None]).sum(0)
    assert float((mixed.double() - want).abs().max()) < 1e-5
    d = torch.randn(n, device=DEV)
    g = torch.zeros(L1, device=DEV)
    for i, h in enumerate(hs):
        T.dot_(g[i:i + 1], d, h)
    gw = torch.stack([(d.double() * h.double()).sum() for h in hs])
    assert float((g.double() - gw).abs().max()) < 1e-3 * float(gw.abs().max())
    g2 = torch.zeros(L1, device=DEV)
    for i, h in enumerate(hs):
        T.dot_(g2[i:i + 1], d, h)
    assert torch.equal(g, g2)                         # bit-reproducible
    dw = torch.full((L1,), 0.25, device=DEV)
    T.softmax_vec_bwd_(dw, s, g)
    wr = w.double().clone().requires_grad_(True)
    (torch.softmax(wr, 0) * gw).sum().backward()
    assert float((dw.double() - 0.25 - wr.grad).abs().max()) < 1e-3 * float(wr.grad.abs().max()) + 1e-6
    # unaligned views take the scalar path
    a = torch.zeros(n + 1, device=DEV)[1:]
    b = torch.randn(n + 1, device=DEV)[1:]
    T.axpy_dev_(a, b.contiguous() if not b.is_contiguous() else b, s[0:1], overwrite=True)
    assert torch.allclose(a, b * s[0], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("kind", ["gelu", "gelu_new"])
def test_activation_with_fused_dropout_is_bit_identical_to_two_passes(kind):
    """mi_act_dropout_fwd/bwd_bf16 = activation + mi_dropout in one pass: same mask (logical index m*N + n), same bf16 rounding points"""
    _, T = _o()
    if kind not in T.KIND:
        pytest.skip(kind)
    M, N = 333, 256
    pre = bfr(rnd(M, N + 8, seed=3)).to(BF).to(DEV)[:, :N]           # a row stride different from N: the mask must follow the logical index
    dy = bfr(rnd(M, N, seed=4)).to(BF).to(DEV)
    drop = (0.2, 99, 1234)
    two = T.act_fwd(pre, kind)
    T.dropout_(two, *drop)
    one = T.act_fwd(pre, kind, drop=drop)
    assert torch.equal(one, two)
    kept = float((one != 0).float().mean())
    assert 0.75 < kept < 0.85
    d2 = dy.clone()
    T.dropout_(d2, *drop)
    two_b = T.act_bwd(d2, pre, kind)
    one_b = T.act_bwd(dy, pre, kind, drop=drop)
    assert torch.equal(one_b, two_b)


def test_dropout_kernels_match_the_host_twin_masks():
    """mi_dropout (8-per-thread form and the scalar fall-back), mi_dropout_add_f32 and the attention softmax's probability dropout (register
    form, Tk <= 256, and loop form) against masks regenerated on the host by synth.dropout_keep for the same (seed, stream, logical index)."""
    import numpy as np
    from huggingface_asr_amd import synth
    _, T = _o()
    p, seed, sid = 0.25, 4321, 77
    for (M, N, ld) in ((61, 256, 256), (61, 256, 264), (61, 250, 250)):         # aligned; padded rows (mask follows m*N + n); odd width -> scalar kernel
        keep = torch.from_numpy(synth.dropout_keep(seed, sid, M * N, p).reshape(M, N))
        for dt in (torch.float32, BF):
            x = bfr(rnd(M, ld, seed=9)).to(dt)
            xd = x.to(DEV)[:, :N]
            got = T.dropout_(xd.clone() if ld == N else xd, p, seed, sid, out=torch.empty((M, N), device=DEV, dtype=dt), alpha=0.5)
            want = (x[:, :N].float() * 0.5 * keep.float() * (1.0 / (1.0 - p))).to(dt)
            assert torch.equal(got.cpu(), want), (M, N, ld, dt)
    r, t = rnd(40, 128, seed=1), rnd(40, 128, seed=2)
    keep = torch.from_numpy(synth.dropout_keep(seed, sid, 40 * 128, p).reshape(40, 128)).float()
    y = T.dropout_add(r.to(DEV), t.to(DEV), 0.5, p, seed, sid)
    assert torch.allclose(y.cpu(), r + 0.5 * t * keep / (1 - p), rtol=1e-6, atol=1e-6)
    for Tq in (37, 290):
        H, B, Tk = 2, 2, Tq
        lds = T.pad8(Tk)
        ac = torch.zeros(H, B, Tq, lds); ac[..., :Tk] = rnd(H, B, Tq, Tk, seed=3)
        prob, pdrop = T.attn_softmax_fwd(ac.to(DEV), None, None, H, B, Tq, Tk, 0.3, False, drop=(p, seed, sid))
        keep = torch.from_numpy(synth.dropout_keep(seed, sid, H * B * Tq * Tk, p).reshape(H, B, Tq, Tk)).float()
        got = pdrop[..., :Tk].float().cpu()
        assert torch.equal(got == 0, (keep == 0) | (prob[..., :Tk].float().cpu() == 0)), Tq          # exactly the host twin's mask
        want = prob[..., :Tk].float().cpu() * keep / (1 - p)                                          # (the kernel scales the un-rounded probability: 1 bf16 ulp)
        assert float((got - want).abs().max()) <= 2 ** -7 * float(want.abs().max()), Tq
        dp = torch.zeros(H, B, Tq, lds); dp[..., :Tk] = rnd(H, B, Tq, Tk, seed=4)
        ds, _ = T.attn_softmax_bwd(prob, dp.to(DEV), H, B, Tq, Tk, 0.3, drop=(p, seed, sid))
        pf = prob[..., :Tk].float().cpu()
        g = dp[..., :Tk] * keep / (1 - p)
        wds = pf * (g - (pf * g).sum(-1, keepdim=True)) * 0.3
        err = (ds[..., :Tk].float().cpu() - wds).abs().max()
        assert float(err) < 2e-2 * float(wds.abs().max()) + 1e-4, (Tq, float(err))


@pytest.mark.parametrize("share", [1, 4])
def test_context_aware_front_end_backward_pieces(share):
    """backward pieces of the gated Conv2d front ends (extractors.py:23-65) against torch autograd: d GELU(z * sigmoid(g)) with shared gate rows, col2im (the
    transpose of im2col, both geometries, accumulating), and the weight / bias gradient of the 1-channel first conv for the (3,3) and (12,3) kernels."""
    ops, T = _o()
    B, Tt, Fq, C = 2, 16, 10, 32
    z = bfr(rnd(B, Tt, Fq, C, seed=31)).requires_grad_()
    g = bfr(rnd(B, Tt // share, Fq, C, seed=32)).requires_grad_()
    dout = bfr(rnd(B, Tt, Fq, C, seed=33))
    y = F.gelu(z.view(B, Tt // share, share, Fq, C) * torch.sigmoid(g)[:, :, None]).reshape(B, Tt, Fq, C)
    y.backward(dout)
    dz, dg = T.gated_act_bwd(dev16(dout).view(-1, C), dev16(z.detach()).view(-1, C), dev16(g.detach()).view(-1, C), B, Tt, Fq, C, share)
    close(dz.view(B, Tt, Fq, C), z.grad, what="dz"); close(dg.view(B, Tt // share, Fq, C), g.grad, what="dg")
    # col2im == autograd of im2col (= F.unfold in channels-last order)
    K, S, P = ((3, 3), (2, 2), (1, 1)) if share == 1 else ((12, 3), (8, 2), (4, 1))
    Cin, Tin, Fin = 16, 32, 20
    T1, F1 = (Tin + 2 * P[0] - K[0]) // S[0] + 1, (Fin + 2 * P[1] - K[1]) // S[1] + 1
    a = bfr(rnd(B, Tin, Fin, Cin, seed=34))
    col = T.im2col_geo(dev16(a), K, S, P, T1, F1)
    unf = F.unfold(a.permute(0, 3, 1, 2), K, padding=P, stride=S)                                  # (B, Cin*KH*KW, L), k = (c, kh, kw)
    want_col = unf.view(B, Cin, K[0] * K[1], T1 * F1).permute(0, 3, 2, 1).reshape(B * T1 * F1, -1)  # -> (b, to, fo), k = (kh, kw, c)
    torch.testing.assert_close(col.float().cpu(), want_col, atol=0, rtol=0)
    dcol = bfr(rnd(B * T1 * F1, K[0] * K[1] * Cin, seed=35))
    fold_in = dcol.view(B, T1 * F1, K[0] * K[1], Cin).permute(0, 3, 2, 1).reshape(B, Cin * K[0] * K[1], T1 * F1)
    want_in = F.fold(fold_in, (Tin, Fin), K, padding=P, stride=S).permute(0, 2, 3, 1)
    din = T.col2im(dev16(dcol), (B, Tin, Fin, Cin), K, S, P, T1, F1)
    close(din, want_in, what="col2im")
    din2 = T.col2im(dev16(dcol), (B, Tin, Fin, Cin), K, S, P, T1, F1, out=din.clone())
    close(din2, 2 * want_in, rel=2.5e-2, what="col2im accumulate")
    # first-layer weight gradient
    x = rnd(B, 64, 80, seed=36)
    w = rnd(C, 1, *K, seed=37, scale=0.2).requires_grad_()
    b = rnd(C, seed=38, scale=0.1).requires_grad_()
    o = F.conv2d(x[:, None], w, b, stride=S, padding=P)
    dy = bfr(rnd(*o.shape, seed=39)).permute(0, 2, 3, 1).contiguous()                              # channels-last
    o.backward(dy.permute(0, 3, 1, 2))
    dw = torch.zeros(C, K[0] * K[1], device=DEV); db = torch.zeros(C, device=DEV)
    T.conv2d_first_wgrad(x.to(DEV), dev16(dy), dw, db, K, S, P)
    torch.testing.assert_close(dw.cpu(), w.grad.reshape(C, -1), atol=2e-3 * float(w.grad.abs().max()), rtol=1e-3)
    torch.testing.assert_close(db.cpu(), b.grad, atol=2e-3 * float(b.grad.abs().max()), rtol=1e-3)


def test_small_fused_passes_of_the_attention_backward():
    """(q + u, q + v) from one read of q == the element-wise op twice, bit for bit (also on a strided view and on an un-aligned width, which takes the element-wise form);
    the in-order column sum of a few partial rows straight to bf16 == torch's sum rounded once."""
    from huggingface_asr_amd import ops_train as T
    g = torch.Generator().manual_seed(3)
    for M, d, ld in ((777, 512, 1536), (50, 36, 36)):
        qkv = (torch.randn(M, ld, generator=g)).to(DEV, torch.bfloat16)
        u, v = torch.randn(d, generator=g).to(DEV), torch.randn(d, generator=g).to(DEV)
        q = qkv[:, :d]
        qu, qv = T.add_rowvec2(q, u, v)
        assert torch.equal(qu, T.add_rowvec(q, u)) and torch.equal(qv, T.add_rowvec(q, v))
        assert torch.equal(qu, (q.float() + u).to(torch.bfloat16))
    for M, N in ((8, 511 * 64), (24, 4096), (1, 64)):
        x = torch.randn(M, N + 128, generator=g).to(DEV)
        got = T.colsum_cast(x[:, 64:64 + N])
        want = x[:, 64:64 + N].double().sum(0)
        assert got.dtype == torch.bfloat16 and got.shape == (N,)
        err = (got.double() - want).abs()
        assert float((err / (want.abs() + 1.0)).max()) < 8e-3                       # one bf16 rounding of an fp32 sum
        seq = x[0, 64:64 + N].clone()
        for m in range(1, M):
            seq = seq + x[m, 64:64 + N]
        assert torch.equal(got, seq.to(torch.bfloat16))                             # rows added in order, rounded once
    a, b = torch.randn(264, 512, generator=g).to(DEV), torch.randn(264, 512, generator=g).to(DEV)
    oa, ob = torch.randn(512, generator=g).to(DEV), torch.randn(512, generator=g).to(DEV)
    wa, wb = oa.double() + a.double().sum(0), ob.double() + b.double().sum(0)
    oa2, ob2 = oa.clone(), ob.clone()
    T.colsum2_acc_(oa, ob, a, b)
    T.colsum2_acc_(oa2, ob2, a, b)
    torch.testing.assert_close(oa.double(), wa, atol=2e-4, rtol=1e-5)
    torch.testing.assert_close(ob.double(), wb, atol=2e-4, rtol=1e-5)
    assert torch.equal(oa, oa2) and torch.equal(ob, ob2)                            # no atomics: the same bits every time


@pytest.mark.parametrize("T_,H,hd", [(250, 4, 128), (500, 2, 64), (97, 2, 64)])
def test_attention_backward_sparse_writes_leave_the_products_unchanged(T_, H, hd):
    """Round 5: with `sparse=True` the fused walk does not write the zeros nobody reads — its dBD buffer is zero-filled once per shape and re-used by every call, P / dS hold
    garbage from the key length rounded up to 128 on.  What the trainer computes from them — dV = P^T dctx and dK = dS^T (q + u) through `bgemm(..., m_valid=lengths)`, the
    position gradient dBD^T (q + v) through the banded product — must be the bits of the dense-write walk, also when the SAME buffer was written by an earlier call with LONGER
    utterances (whatever that call left beyond the new lengths must have been zeroed again), and dQ / the bias sums come from the same walk either way."""
    ops, T = _o()
    B, d, Tq = 4, H * hd, T_
    qkv = torch.cat([bfr(rnd(B * Tq, d, seed=700 + i, scale=0.8)) for i in range(3)], 1).to(DEV, BF)
    pos = dev16(bfr(rnd(2 * Tq - 1, d, seed=705, scale=0.8)))
    u, vb = (0.2 * rnd(H, hd, seed=706)).to(DEV), (0.2 * rnd(H, hd, seed=707)).to(DEV)
    dctx = dev16(bfr(rnd(B * Tq, d, seed=708)))
    pad, Ps = T.band_geometry(Tq)
    off = pad
    long_len = torch.tensor([Tq, Tq - 3, Tq, Tq - 1], dtype=torch.int32, device=DEV)
    short_len = torch.tensor([max(1, Tq // 3), max(1, Tq // 2 + 5), 7, max(1, Tq - 40)], dtype=torch.int32, device=DEV)

    def run(lengths, sparse):
        lse = torch.empty((B, H, Tq), device=DEV, dtype=torch.float32)
        ctx = ops.attention_qkv(qkv, B, Tq, H, lse=lse, pos=pos, bias_u=u, bias_v=vb, lengths=lengths)
        dq = torch.full((B * Tq, d), float("nan"), device=DEV, dtype=BF)
        prob, ds, dbd, su, sv, qu, qv = T.attn_bwd_probs(qkv, B, Tq, H, ctx, dctx, lse, dq, pos=pos, bias_u=u, bias_v=vb, lengths=lengths, qb=True, sparse=sparse)
        Ts = prob.shape[-1]
        sTT = (B * Tq * Ts, Tq * Ts)
        dv = torch.full((B * Tq, d), float("nan"), device=DEV, dtype=BF)
        dk = torch.full((B * Tq, d), float("nan"), device=DEV, dtype=BF)
        T.bgemm(prob, (*sTT, 1, Ts), dctx, (hd, Tq * d, 1, d), dv, (hd, Tq * d, d), H, B, Tq, hd, Tq, m_valid=lengths)
        T.bgemm(ds, (*sTT, 1, Ts), qu, (hd, Tq * d, 1, d), dk, (hd, Tq * d, d), H, B, Tq, hd, Tq, m_valid=lengths)
        Kp = Ps
        dpp = torch.full((B, Kp * d), float("nan"), device=DEV, dtype=torch.float32)
        T.bgemm(dbd, (B * Tq * Ps, Tq * Ps, 1, Ps), qv, (hd, Tq * d, 1, d), dpp, (hd, Kp * d, d), H, B, Kp, hd, Tq, band=(Tq, Tq - 1 + off, 1))
        torch.cuda.synchronize()
        return dq.clone(), su.clone(), sv.clone(), dv, dk, dpp

    want_long, want_short = run(long_len, False), run(short_len, False)
    got_long = run(long_len, True)                       # fills the static buffer's maximal band
    got_short = run(short_len, True)                     # ... and must clear what the long call left beyond the short lengths
    got_long2 = run(long_len, True)
    for name, want, got in (("long", want_long, got_long), ("short after long", want_short, got_short), ("long again", want_long, got_long2)):
        for w, g_, what in zip(want, got, ("dQ", "su", "sv", "dV", "dK", "d(posp) partials")):
            assert torch.equal(w, g_), (name, what, float((w.float() - g_.float()).abs().max()))
