"""GPU parity, op by op: every C-ABI entry point (through huggingface_asr_amd.ops -> ctypes -> libhfasr_hip.so)
against the oracle's restatement of the same reference op, on identical bf16-rounded inputs.

Tolerances: outputs stored in bf16 are compared at ~1 bf16 ulp (2^-8 relative) of the fp32 oracle value;
fp32 outputs at 1e-3 abs (accumulation order only)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden
from oracle import ebranchformer_ref as R
from oracle import fbank_ref

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from huggingface_asr_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bfr(x):
    return x.to(torch.bfloat16).float()


def assert_close_bf16(got, want, atol=2e-2, rtol=1.2e-2, what=""):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} off, max err {float(err.max()):.4g}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (250, 512, 512), (300, 5001, 64), (1000, 192, 1024), (64, 64, 128), (129, 130, 72)])
def test_gemm_plain(M, N, K):
    ops = _ops()
    a, w, b = bfr(rnd(M, K, seed=1)), bfr(rnd(N, K, seed=2, scale=1 / math.sqrt(K))), rnd(N, seed=3)
    want = a @ w.t() + b
    got32 = ops.gemm(a.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), b.to(DEV), out_dtype=torch.float32)
    torch.testing.assert_close(got32.cpu(), want, atol=2e-3, rtol=1e-4)
    got16 = ops.gemm(a.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), b.to(DEV))
    assert_close_bf16(got16, want, what="gemm bf16")


@pytest.mark.parametrize("M,N,K,ld", [(777, 5001, 512, 5056), (8000, 5001, 512, 5004), (300, 300, 256, 304), (256, 256, 128, 256), (100, 5001, 64, 5004), (513, 1030, 192, 1032)])
def test_ctc_head_gemm_leaves_the_row_log_sum_exp(M, N, K, ld):
    """mi_gemm_lse_f32 (the CTC head, e_branchformer.py:456-457 + the log_softmax of :472-488): the logits are bit-identical to the plain fp32-out GEMM's, the row
    log-sum-exp from the epilogue's per-64-column (max, sum exp) pairs agrees with fp64 log-sum-exp of those logits to 2e-6 relative and with mi_row_lse to the order of the
    sums — ragged N (5001 = 19 tiles + 137 columns), ragged M, large logits (no overflow: the pairs carry their own maximum), a K outside the kernel (falls back)."""
    ops = _ops()
    a = rnd(M, K, seed=1, scale=3.0).to(DEV, torch.bfloat16)
    w = rnd(N, K, seed=2, scale=2.0 / math.sqrt(K)).to(DEV, torch.bfloat16)
    b = (5.0 * rnd(N, seed=3)).to(DEV)
    out = torch.full((M, ld), float("nan"), device=DEV)
    lse = ops.gemm_lse(a, w, b, out)
    ref = torch.empty((M, ld), device=DEV)
    ops.gemm(a, w, b, out=ref)
    assert torch.equal(out[:, :N], ref[:, :N])
    want = torch.logsumexp(ref[:, :N].double(), dim=1)
    torch.testing.assert_close(lse.double(), want, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(lse, ops.row_lse(ref[:, :N]), rtol=2e-6, atol=2e-6)
    assert float(ref[:, :N].abs().max()) > 20.0                    # the logits are large enough that a sum of exp without the running maximum would lose digits


def test_gemm_identity_asymmetric():
    """A = I with an asymmetric W catches a transposed C write (guide §3)."""
    ops = _ops()
    n = 128
    a = torch.eye(n)
    w = bfr(torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 100.0)
    got = ops.gemm(a.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), None, out_dtype=torch.float32)
    torch.testing.assert_close(got.cpu(), w.t().contiguous(), atol=0, rtol=0)


def test_gemm_epilogues():
    ops = _ops()
    M, N, K = 257, 384, 256
    a, w, b = bfr(rnd(M, K, seed=4)), bfr(rnd(N, K, seed=5, scale=1 / 16)), rnd(N, seed=6)
    res = rnd(M, N, seed=7)
    ad, wd, bd = a.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), b.to(DEV)
    lin = a @ w.t() + b
    assert_close_bf16(ops.gemm(ad, wd, bd, act="gelu"), F.gelu(lin), what="gelu")
    x = res.to(DEV).clone()
    ops.gemm(ad, wd, bd, out=x, resid=x, alpha=0.5)            # in-place residual update
    torch.testing.assert_close(x.cpu(), res + 0.5 * lin, atol=2e-3, rtol=1e-4)
    # strided output (writes one half of a concat buffer) leaves the other half untouched
    cat = torch.full((M, 2 * N), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.gemm(ad, wd, bd, out=cat[:, N:])
    assert_close_bf16(cat[:, N:], lin, what="strided out")
    assert bool((cat[:, :N] == 7.0).all())
    # transposed product with per-row bias and time-padded column remap (the V^T projection)
    T, Tp, B = 50, 64, 4
    a2 = bfr(rnd(B * T, K, seed=8))
    wv, bv = bfr(rnd(96, K, seed=9, scale=1 / 16)), rnd(96, seed=10)
    vt = torch.zeros((96, B * Tp), device=DEV, dtype=torch.bfloat16)
    ops.gemm(wv.to(DEV, torch.bfloat16), a2.to(DEV, torch.bfloat16), bv.to(DEV), out=vt, bias_per_row=True, col_remap=(T, Tp))
    want = (a2 @ wv.t() + bv).t().reshape(96, B, T)
    got = vt.float().cpu().reshape(96, B, Tp)
    assert_close_bf16(got[:, :, :T], want, what="V^T")
    assert bool((got[:, :, T:] == 0).all())


@pytest.mark.parametrize("M,N,K,f32", [(24000, 768, 768, True), (24000, 768, 3072, True), (23987, 2304, 768, False), (24000, 3072, 768, False)])
def test_gemm_tail_round_on_quarter_tiles_same_bits(M, N, K, f32):
    """Wave quantization (gemm_glds.hip, round 4): when the 256 x 256 tiles of a GEMM leave a last round less than half full (Whisper-small's shapes at 16 x 30 s: 282 /
    846 / 1128 tiles on 256 CUs), the product dispatch gives that round's rows to the 128 x 128 phase kernel.  Same K order and MFMA shape: the result must equal, bit
    for bit, the one launch on 256 x 256 tiles (variant 40) — fp32 out with residual (attention / FFN out) and bf16 out with GELU (FFN in), a ragged last tile included."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(5)
    a = (torch.randn(M, K, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=DEV, generator=g)
    if f32:
        res = torch.randn(M, N, device=DEV, generator=g)
        got = ops.gemm(a, w, b, out=res.clone(), resid=res, alpha=1.0)
        one = ops.gemm(a, w, b, out=res.clone(), resid=res, alpha=1.0, variant=40)
    else:
        got = ops.gemm(a, w, b, act="gelu")
        one = ops.gemm(a, w, b, act="gelu", variant=40)
    assert torch.equal(got, one)
    rows = torch.tensor([0, 255, 21759, 21760, 21761, M - 1])               # both sides of the split (85 M tiles x 256 rows) against fp32 torch
    want = a[rows].float() @ w.float().t() + b
    want = (res[rows] + want) if f32 else F.gelu(want)
    torch.testing.assert_close(got[rows].float().cpu(), want.cpu(), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("causal", [False, True])
def test_conv_subsampling(causal):
    ops = _ops()
    B, T, Fd, C1, C2 = 2, 61, 80, 32, 64
    x = rnd(B, T, Fd, seed=11)
    w1, b1 = rnd(C1, 1, 3, 3, seed=12, scale=0.3), rnd(C1, seed=13, scale=0.1)
    w2, b2 = bfr(rnd(C2, C1, 3, 3, seed=14, scale=0.08)), rnd(C2, seed=15, scale=0.1)
    xin = x[:, None]
    if causal:
        h1 = F.gelu(F.conv2d(F.pad(xin, (2, 0, 2, 0)), w1, b1, stride=2))
    else:
        h1 = F.gelu(F.conv2d(xin, w1, b1, stride=2, padding=1))
    g1 = ops.conv2d_first_gelu(x.to(DEV), w1.reshape(C1, 9).to(DEV), b1.to(DEV), causal=causal)
    assert_close_bf16(g1.permute(0, 3, 1, 2), h1, what="conv1")
    h1q = g1.float().cpu().permute(0, 3, 1, 2)
    if causal:
        h2 = F.gelu(F.conv2d(F.pad(h1q, (2, 0, 2, 0)), w2, b2, stride=2))
    else:
        h2 = F.gelu(F.conv2d(h1q, w2, b2, stride=2, padding=1))
    w2p = w2.permute(0, 2, 3, 1).reshape(C2, 9 * C1).to(DEV, torch.bfloat16)
    g2 = ops.conv2d_cl(g1, w2p, b2.to(DEV), causal=causal)
    assert_close_bf16(g2.permute(0, 3, 1, 2), h2, what="conv2")


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("B,T,Fd", [(2, 61, 80), (3, 130, 23)])
def test_conv2d_first_on_the_matrix_cores(B, T, Fd, causal):
    """C = 256 (every model of the recipes): Conv2d(1 -> 256, 3x3, stride 2) + GELU as two K = 16 MFMAs per 32 positions x 32 channels with x, w, b each split into two
    bf16 parts (conv.hip conv2d_first3_mfma_kernel) — against fp32 torch: the result before its bf16 rounding must agree to ~2^-15 (a plain bf16 product would be 2^-8 off),
    checked as: the bf16 outputs equal torch's GELU(conv) rounded to bf16 except for last-bit ties; position count not a multiple of the 32-row tile; causal (leading) padding."""
    ops = _ops()
    C1 = 256
    x = rnd(B, T, Fd, seed=41) * 3.0 + 0.7                      # log-mel-like magnitudes
    w1, b1 = rnd(C1, 1, 3, 3, seed=42, scale=0.3), rnd(C1, seed=43, scale=0.2)
    xin = x[:, None]
    h1 = F.gelu(F.conv2d(F.pad(xin, (2, 0, 2, 0)), w1, b1, stride=2)) if causal else F.gelu(F.conv2d(xin, w1, b1, stride=2, padding=1))
    g1 = ops.conv2d_first_gelu(x.to(DEV), w1.reshape(C1, 9).to(DEV), b1.to(DEV), causal=causal)
    got = g1.permute(0, 3, 1, 2).float().cpu()
    assert got.shape == h1.shape and (B * h1.shape[2] * h1.shape[3]) % 32 != 0
    want = h1.to(torch.bfloat16).float()
    ulp = torch.maximum(want.abs(), torch.tensor(2.0 ** -126)) * 2.0 ** -7            # one bf16 step at |want|
    d = (got - want).abs()
    assert bool((d <= ulp * 1.01 + 1e-4).all()), float((d / ulp).max())              # never more than one step (GELU fit: 2.6e-5 absolute) ...
    big = want.abs() > 0.05                                                         # (below it the GELU fit's 2.6e-5 is comparable to a bf16 step: last-bit flips)
    assert float((d[big] > 0).float().mean()) < 0.03 and int(big.sum()) > 1000      # ... and almost always the same bf16 value: the split-precision product is fp32-like
    assert torch.equal(g1, ops.conv2d_first_gelu(x.to(DEV), w1.reshape(C1, 9).to(DEV), b1.to(DEV), causal=causal))


@pytest.mark.parametrize("d", [64, 512, 1024])
def test_layernorm_chain(d):
    ops = _ops()
    M, T = 37, 10
    x = rnd(M, d, seed=20) * 2 + 0.3
    gs = [1 + 0.1 * rnd(d, seed=21 + i) for i in range(3)]
    bs = [0.1 * rnd(d, seed=31 + i) for i in range(3)]
    dev = lambda t: t.to(DEV)
    # (a) plain LN -> bf16 ; (b) dual ; (c) chain with fp32 store + masked rows
    xa = dev(x)
    oa = torch.empty((M, d), device=DEV, dtype=torch.bfloat16)
    ob = torch.empty_like(oa)
    ops.layernorm_chain(xa, lna=(dev(gs[0]), dev(bs[0])), outa=oa, lnb=(dev(gs[1]), dev(bs[1])), outb=ob)
    assert_close_bf16(oa, F.layer_norm(x, (d,), gs[0], bs[0]), what="ln a")
    assert_close_bf16(ob, F.layer_norm(x, (d,), gs[1], bs[1]), what="ln b")
    lens = torch.tensor([7, 10, 3, 9], dtype=torch.int32)
    M2 = 4 * T
    x2 = rnd(M2, d, seed=40)
    xm = x2.clone().reshape(4, T, d)
    for b in range(4):
        xm[b, lens[b]:] = 0
    xm = xm.reshape(M2, d)
    xd = dev(x2)
    o32 = torch.empty((M2, d), device=DEV)
    ops.layernorm_chain(xd, lengths=dev(lens), T=T, ln1=(dev(gs[0]), dev(bs[0])), store_y=xd,
                        lna=(dev(gs[2]), dev(bs[2])), outa32=o32)
    y = F.layer_norm(xm, (d,), gs[0], bs[0])
    torch.testing.assert_close(xd.cpu(), y, atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(o32.cpu(), F.layer_norm(y, (d,), gs[2], bs[2]), atol=5e-5, rtol=1e-5)
    # mask only (no stage 1): padded rows become exactly zero in the residual stream
    xd2 = dev(x2)
    ops.layernorm_chain(xd2, lengths=dev(lens), T=T, store_y=xd2, lna=(dev(gs[0]), dev(bs[0])), outa=torch.empty((M2, d), device=DEV, dtype=torch.bfloat16))
    torch.testing.assert_close(xd2.cpu(), xm, atol=0, rtol=0)


def _attn_ref(q, k, v, B, T, H, pos=None, u=None, vb=None, lengths=None, causal=False):
    d = q.shape[1]
    hd = d // H
    qh = q.view(B, T, H, hd).transpose(1, 2)
    kh = k.view(B, T, H, hd).transpose(1, 2)
    vh = v.view(B, T, H, hd).transpose(1, 2)
    if pos is not None:
        pp = pos.view(-1, H, hd).transpose(0, 1)
        scores = R.rel_attention_scores(qh, kh, pp, u, vb, R.bf16_round)
    else:
        scores = qh @ kh.transpose(-2, -1) / math.sqrt(hd)
    if lengths is not None:
        keymask = torch.arange(T)[None, :] >= lengths[:, None]
        scores = scores.masked_fill(keymask[:, None, None, :], float("-inf"))
    if causal:
        scores = scores.masked_fill(torch.ones(T, T, dtype=torch.bool).triu(1), float("-inf"))
    return (torch.softmax(scores, -1) @ vh).transpose(1, 2).reshape(B * T, d)


@pytest.mark.parametrize("T,H,hd,rel,causal", [(50, 4, 16, True, False), (250, 4, 128, True, False), (75, 2, 64, False, False),
                                               (97, 4, 32, True, True), (33, 4, 16, False, True), (300, 4, 128, True, False)])
def test_attention(T, H, hd, rel, causal):
    ops = _ops()
    B, d = 3, H * hd
    Tp = (T + 31) // 32 * 32
    q, k, v = (bfr(rnd(B * T, d, seed=50 + i, scale=0.8)) for i in range(3))
    lengths = torch.tensor([T, max(1, T - 13), max(1, T // 2)], dtype=torch.int32)
    pos = bfr(rnd(2 * T - 1, d, seed=55, scale=0.8)) if rel else None
    u, vb = (0.2 * rnd(H, hd, seed=56), 0.2 * rnd(H, hd, seed=57)) if rel else (None, None)
    want = _attn_ref(q, k, v, B, T, H, pos, u, vb, lengths, causal)
    qk = torch.cat([q, k], 1).to(DEV, torch.bfloat16)
    vt = torch.zeros((d, B * Tp), dtype=torch.bfloat16, device=DEV)
    vt.view(d, B, Tp)[:, :, :T] = v.t().reshape(d, B, T).to(DEV, torch.bfloat16)
    got = ops.attention(qk[:, :d], qk[:, d:], vt, Tp, B, T, H, pos=None if pos is None else pos.to(DEV, torch.bfloat16),
                        bias_u=None if u is None else u.to(DEV), bias_v=None if vb is None else vb.to(DEV),
                        lengths=lengths.to(DEV), causal=causal)
    assert_close_bf16(got, want, atol=1.5e-2, rtol=2e-2, what="attention")


@pytest.mark.parametrize("T,H,hd,rel,causal", [(250, 4, 128, True, False), (75, 2, 64, False, False), (97, 4, 64, True, True),
                                               (33, 2, 128, False, True), (300, 4, 128, True, False), (129, 1, 128, True, False),
                                               (500, 2, 64, True, False), (1500, 12, 64, False, False), (1500, 2, 128, True, False)])
def test_attention_lds_staged(T, H, hd, rel, causal):
    """LDS-staged kernel (fused QKV input, tr-read V, carried G tile) against the same oracle."""
    ops = _ops()
    B, d = 3, H * hd
    q, k, v = (bfr(rnd(B * T, d, seed=150 + i, scale=0.8)) for i in range(3))
    lengths = torch.tensor([T, max(1, T - 13), max(1, T // 2)], dtype=torch.int32)
    pos = bfr(rnd(2 * T - 1, d, seed=155, scale=0.8)) if rel else None
    u, vb = (0.2 * rnd(H, hd, seed=156), 0.2 * rnd(H, hd, seed=157)) if rel else (None, None)
    want = _attn_ref(q, k, v, B, T, H, pos, u, vb, lengths, causal)
    qkv = torch.cat([q, k, v], 1).to(DEV, torch.bfloat16)
    got = ops.attention_qkv(qkv, B, T, H, pos=None if pos is None else pos.to(DEV, torch.bfloat16),
                            bias_u=None if u is None else u.to(DEV), bias_v=None if vb is None else vb.to(DEV),
                            lengths=lengths.to(DEV), causal=causal)
    assert_close_bf16(got, want, atol=1.5e-2, rtol=2e-2, what="attention (lds)")


@pytest.mark.parametrize("B,T,H,hd,rel,causal", [(3, 250, 4, 128, True, False), (2, 97, 4, 64, True, True), (8, 500, 2, 64, True, False), (3, 33, 2, 128, False, True),
                                                 (5, 129, 1, 128, True, False), (4, 1500, 2, 64, False, False), (1, 1, 2, 64, False, False)])
def test_attention_eight_wave_form_against_the_oracle_and_the_four_wave_kernel(B, T, H, hd, rel, causal):
    """Round 4's eight-wave forward FORCED on every shape (variant 2: also head 64 with relative positions, which the product routes to the four-wave kernel; block counts with
    and without the XCD-aware order: N % 8 == 0 or not; ragged key lengths; a lone query) against the oracle, and against the four-wave kernel of rounds 1-3 (variant 1) —
    they differ by the summation order over the keys only (the wave pair of a query group splits them)."""
    ops = _ops()
    d = H * hd
    q, k, v = (bfr(rnd(B * T, d, seed=250 + i, scale=0.8)) for i in range(3))
    lengths = torch.tensor([max(1, T - 7 * b) if b % 2 == 0 else max(1, T // (b + 1)) for b in range(B)], dtype=torch.int32)
    pos = bfr(rnd(2 * T - 1, d, seed=255, scale=0.8)) if rel else None
    u, vb = (0.2 * rnd(H, hd, seed=256), 0.2 * rnd(H, hd, seed=257)) if rel else (None, None)
    want = _attn_ref(q, k, v, B, T, H, pos, u, vb, lengths, causal)
    qkv = torch.cat([q, k, v], 1).to(DEV, torch.bfloat16)
    kw = dict(pos=None if pos is None else pos.to(DEV, torch.bfloat16), bias_u=None if u is None else u.to(DEV), bias_v=None if vb is None else vb.to(DEV),
              lengths=lengths.to(DEV), causal=causal)
    got8 = ops.attention_qkv(qkv, B, T, H, variant=2, **kw)
    got4 = ops.attention_qkv(qkv, B, T, H, variant=1, **kw)
    assert_close_bf16(got8, want, atol=1.5e-2, rtol=2e-2, what="attention (eight-wave)")
    assert_close_bf16(got4, want, atol=1.5e-2, rtol=2e-2, what="attention (four-wave)")
    dd = (got8.float() - got4.float()).abs()
    assert float(dd.max()) < 1.6e-2 and float(dd.mean()) < 3e-4, (float(dd.max()), float(dd.mean()))
    assert torch.equal(got8, ops.attention_qkv(qkv, B, T, H, variant=2, **kw))            # bit-reproducible launch to launch


@pytest.mark.parametrize("causal", [False, True])
def test_csgu_and_merge(causal):
    ops = _ops()
    B, T, Cc, K = 2, 150, 128, 31
    u = bfr(rnd(B * T, 2 * Cc, seed=60))
    g, b = 1 + 0.1 * rnd(Cc, seed=61), 0.1 * rnd(Cc, seed=62)
    w, wb = rnd(Cc, 1, K, seed=63, scale=0.2), 0.1 * rnd(Cc, seed=64)
    r_, g_ = u.view(B, T, 2 * Cc).chunk(2, -1)
    gn = F.layer_norm(g_, (Cc,), g, b)
    dil = (K - 1) // 2 if causal else 1
    conv = R.dwconv1d(gn, w, wb, causal, dil)
    want = (r_ * conv).reshape(B * T, Cc)
    got = ops.csgu(u.to(DEV, torch.bfloat16), g.to(DEV), b.to(DEV), w.reshape(Cc, K).to(DEV), wb.to(DEV), B, T,
                   pad_left=(K - 1) * dil if causal else None, dilation=dil)
    assert_close_bf16(got, want, what="csgu")
    m = bfr(rnd(B * T, 2 * Cc, seed=65))
    w2, wb2 = rnd(2 * Cc, 1, K, seed=66, scale=0.2), 0.1 * rnd(2 * Cc, seed=67)
    want2 = (m.view(B, T, -1) + R.dwconv1d(m.view(B, T, -1), w2, wb2)).reshape(B * T, -1)
    got2 = ops.dwconv_residual(m.to(DEV, torch.bfloat16), w2.reshape(2 * Cc, K).to(DEV), wb2.to(DEV), B, T)
    assert_close_bf16(got2, want2, what="merge dwconv")


def test_rotary():
    ops = _ops()
    B, T, H, hd = 2, 40, 4, 32
    x = bfr(rnd(B * T, H * hd, seed=70))
    cos, sin = R.rotary_table(T, hd)
    want = R.apply_rotary(x.view(B, T, -1), cos, sin, H).reshape(B * T, -1)
    got = ops.rotary(x.to(DEV, torch.bfloat16), cos.contiguous().to(DEV), sin.contiguous().to(DEV), T, H)
    assert_close_bf16(got, want, what="rotary")


@pytest.mark.parametrize("wave", ["sweep", "noise", "silence_padded"])
def test_fbank_golden(wave):
    """HIP log-mel (+CMVN) against the REFERENCE's own outputs (tests/golden/fbank.npz)."""
    from huggingface_asr_amd import fbank as FB
    g = load_golden("fbank")
    tb = FB.FbankTables(80)
    w = torch.from_numpy(g[f"{wave}_wave"])[None].to(DEV)
    raw, frames = FB.fbank_gpu(w, tb, normalize=None)
    assert int(frames[0]) == g[f"{wave}_raw"].shape[0]
    np.testing.assert_allclose(raw[0].cpu().numpy(), g[f"{wave}_raw"], atol=2e-5, rtol=0)
    cm, _ = FB.fbank_gpu(w, tb, normalize="utterance")
    np.testing.assert_allclose(cm[0].cpu().numpy(), g[f"{wave}_cmvn"], atol=2e-5, rtol=0)


def test_fbank_ragged_batch_and_padding():
    from huggingface_asr_amd import fbank as FB
    from huggingface_asr_amd import synth
    tb = FB.FbankTables(80)
    w = synth.waveforms(3, 3, 16000)
    ns = torch.tensor([16000, 9000, 12345], dtype=torch.int32)
    feats, frames = FB.fbank_gpu(torch.from_numpy(w).to(DEV), tb, num_samples=ns.to(DEV), pad_frames_to=100)
    assert feats.shape == (3, 100, 80)
    for b in range(3):
        ref = fbank_ref.extract(w[b, : int(ns[b])])
        n = ref.shape[0]
        assert int(frames[b]) == n
        np.testing.assert_allclose(feats[b, :n].cpu().numpy(), ref, atol=3e-5, rtol=0)
        assert bool((feats[b, n:] == 0).all())
    # global normalisation
    means, stds = torch.linspace(5, 9, 80), torch.linspace(2, 4, 80)
    fg, _ = FB.fbank_gpu(torch.from_numpy(w[:1]).to(DEV), tb, normalize="global", global_means=means.to(DEV), global_stds=stds.to(DEV))
    np.testing.assert_allclose(fg[0].cpu().numpy(), fbank_ref.extract(w[0], "global", means.numpy(), stds.numpy()), atol=3e-5, rtol=0)


@pytest.mark.parametrize("case", ["basic", "repeat", "infeasible", "empty_target"])
def test_ctc_known_answers(case):
    ops = _ops()
    g = load_golden("ctc_known")
    logits = torch.from_numpy(g[f"{case}/logits"]).to(DEV)
    labels = torch.from_numpy(g[f"{case}/labels"]).to(DEV)
    in_len = torch.from_numpy(g[f"{case}/in_len"]).to(DEV, torch.int32)
    for zi in (0, 1):
        for red in ("mean", "sum", "none"):
            got, _, _ = ops.ctc_loss(logits, labels, in_len, reduction=red, zero_infinity=bool(zi))
            np.testing.assert_allclose(got.cpu().numpy(), g[f"{case}/{red}/{zi}"], rtol=2e-5, atol=2e-5)


def test_ctc_large_vs_oracle():
    ops = _ops()
    B, T, V, U = 4, 250, 5000, 40
    logits = rnd(B, T, V + 1, seed=80)
    labels = torch.randint(0, V, (B, U), generator=torch.Generator().manual_seed(81))
    labels[1, 30:] = -100
    labels[2, 5] = -100          # hole inside: masked_select semantics keep the remaining ids in order
    in_len = torch.tensor([248, 200, 248, 100], dtype=torch.int32)
    tl = (labels >= 0).sum(-1)
    want = R.ctc_loss_ref(torch.log_softmax(logits, -1), labels, in_len, tl, blank=V, reduction="none")
    got, _, tlg = ops.ctc_loss(logits.to(DEV), labels.to(DEV), in_len.to(DEV), reduction="none")
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-5)
    assert tlg.cpu().tolist() == tl.tolist()
    got_bf, _, _ = ops.ctc_loss(logits.to(DEV, torch.bfloat16), labels.to(DEV), in_len.to(DEV), reduction="none")
    want_bf = R.ctc_loss_ref(torch.log_softmax(bfr(logits), -1), labels, in_len, tl, blank=V, reduction="none")
    np.testing.assert_allclose(got_bf.cpu().numpy(), want_bf.numpy(), rtol=2e-5)


def test_gelu_is_total_over_the_bf16_range():
    """The forward GELU (x * logistic(odd quintic), common.hpp) against erf-GELU over [-30, 30] and at +-inf: the quintic's exponent turns around at |x| ~ 11
    unless its argument is clamped — un-clamped, GELU(12) came out as 9e-12 and GELU(-12) as -12."""
    from huggingface_asr_amd import ops_train as T
    x = torch.cat([torch.linspace(-30, 30, 8192 - 8), torch.tensor([float("inf"), float("-inf"), 10.5, -10.5, 11.2, -11.2, 12.0, -12.0])]).to(torch.bfloat16).reshape(8, 1024)
    got = T.act_fwd(x.to(DEV)).float().cpu()
    want = F.gelu(x.float())
    fin = torch.isfinite(want)
    err = (got[fin] - want[fin]).abs()
    assert float((err - (2e-3 + 2 ** -8 * want[fin].abs())).max()) <= 0, float(err.max())
    assert got.reshape(-1)[-8].item() == float("inf") and got.reshape(-1)[-7].item() == 0.0 and not torch.isnan(got).any()


def test_context_aware_front_end_pieces():
    """The kernels of the gated Conv2d front ends (extractors.py:23-65) one by one against torch on the same operands: the general-geometry first conv (raw / GELU; the
    shared gate's (12,3) / (8,2) / (4,1)), the fused two-bank first layer, the general-geometry implicit GEMM, and GELU(z * sigmoid(g)) with one gate row per four rows."""
    ops = _ops()
    B, T, Fd, C1, C2 = 2, 96, 80, 32, 64
    x = rnd(B, T, Fd, seed=21)
    w1, b1 = rnd(C1, 1, 3, 3, seed=22, scale=0.3), rnd(C1, seed=23, scale=0.1)
    wg, bg = rnd(C1, 1, 12, 3, seed=24, scale=0.15), rnd(C1, seed=25, scale=0.1)
    wg3 = rnd(C1, 1, 3, 3, seed=26, scale=0.3)
    xin = x[:, None]
    z = F.conv2d(xin, w1, b1, stride=2, padding=1)                               # (B, C, 48, 40)
    g = F.conv2d(xin, wg, bg, stride=(8, 2), padding=(4, 1))                     # (B, C, 12, 40)
    cl = lambda t: t.permute(0, 2, 3, 1)
    gz = ops.conv2d_first_geo(x.to(DEV), w1.reshape(C1, 9).to(DEV), b1.to(DEV), act="none")
    gg = ops.conv2d_first_geo(x.to(DEV), wg.reshape(C1, 36).to(DEV), bg.to(DEV), K=(12, 3), stride=(8, 2), pad=(4, 1), act="none")
    assert gg.shape == (B, 12, 40, C1)
    assert_close_bf16(gz, cl(z), what="conv1 raw"); assert_close_bf16(gg, cl(g), what="shared gate conv1 raw")
    assert_close_bf16(ops.conv2d_first_geo(x.to(DEV), w1.reshape(C1, 9).to(DEV), b1.to(DEV), act="gelu"), cl(F.gelu(z)), what="conv1 gelu (generic kernel)")
    # fused two-bank layer 1
    g3 = F.conv2d(xin, wg3, bg, stride=2, padding=1)
    fused = ops.conv2d_first_gated_gelu(x.to(DEV), w1.reshape(C1, 9).to(DEV), b1.to(DEV), wg3.reshape(C1, 9).to(DEV), bg.to(DEV))
    assert_close_bf16(fused, cl(F.gelu(z * torch.sigmoid(g3))), what="fused gated conv1")
    # GELU(z * sigmoid(g)), shared rows: what the reference's view / unsqueeze computes (extractors.py:52-53)
    zr, gr = bfr(cl(z)), bfr(cl(g))
    want = F.gelu(zr.view(B, 12, 4, 40, C1) * torch.sigmoid(gr)[:, :, None])
    got = ops.gated_act(zr.to(DEV, torch.bfloat16), gr.to(DEV, torch.bfloat16), B, 48, 40, C1, share=4)
    assert_close_bf16(got.view(B, 12, 4, 40, C1), want, what="gated_act share=4")
    # layer 2: the shared gate's geometry through the implicit GEMM (K = 36 * C1)
    a1 = bfr(rnd(B, 48, 40, C1, seed=27))
    w2g, b2g = bfr(rnd(C2, C1, 12, 3, seed=28, scale=0.03)), rnd(C2, seed=29, scale=0.1)
    want2 = cl(F.conv2d(a1.permute(0, 3, 1, 2), w2g, b2g, stride=(8, 2), padding=(4, 1)))
    got2 = ops.conv2d_cl_geo(a1.to(DEV, torch.bfloat16), w2g.permute(0, 2, 3, 1).reshape(C2, -1).contiguous().to(DEV, torch.bfloat16), b2g.to(DEV),
                             K=(12, 3), stride=(8, 2), pad=(4, 1), act="none")
    assert got2.shape == (B, 6, 20, C2)
    assert_close_bf16(got2, want2, what="shared gate conv2")


def _lib_call_inplace(ops, a_in, w_p, b_p, x, alpha):
    from huggingface_asr_amd import _lib
    a, w, b = a_in.to(DEV, torch.bfloat16), w_p.to(DEV, torch.bfloat16), b_p.to(DEV)
    M, K = a.shape
    N = w.shape[0]
    c2 = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    st = torch.zeros((M, 32), device=DEV, dtype=torch.float32)
    _lib.check(_lib.lib().mi_gemm_resid_stats_f32_v(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), x.data_ptr(), N, x.data_ptr(), N, float(alpha), c2.data_ptr(), N, st.data_ptr(),
                                                    M, N, K, 40, torch.cuda.current_stream().cuda_stream), "mi_gemm_resid_stats_f32_v")
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,d,I,wide", [(1000, 512, 2048, False), (700, 256, 1024, False), (1000, 512, 2048, True), (700, 256, 1024, True)])
def test_layernorm_folded_into_the_gemms(M, d, I, wide):
    """LN(x) W^T + b evaluated as rstd (bf16(x) W'^T) - rstd mu colsum(W') + (W beta + b) (csrc/gemm_args.hpp): the producer GEMM's epilogue (fp32 rows + bf16 copy + per-row
    partial statistics), the LayerNorm kernel's producer mode, and the consumer GEMM with both statistic forms (1 pair / one pair per 32 columns), against torch.
    `wide`: the producer on the 256 x 256 tile of the throughput mode (one pair per 64 columns)."""
    ops = _ops()
    x0 = rnd(M, d, seed=41, scale=2.0) + 0.7                                        # a residual stream with a mean
    a_in, w_p, b_p = bfr(rnd(M, I, seed=42)), bfr(rnd(d, I, seed=43, scale=I ** -0.5)), rnd(d, seed=44, scale=0.1)
    gam, bet = 1.0 + 0.2 * rnd(d, seed=45), 0.1 * rnd(d, seed=46)
    W, b = rnd(I, d, seed=47, scale=d ** -0.5), rnd(I, seed=48, scale=0.1)
    # producer: x = x0 + 0.5 (a W_p^T + b_p)
    c, c2, st = ops.gemm_resid_stats(a_in.to(DEV, torch.bfloat16), w_p.to(DEV, torch.bfloat16), b_p.to(DEV), x0.to(DEV), alpha=0.5, wide=wide)
    x = x0 + 0.5 * (a_in @ w_p.t() + b_p)
    torch.testing.assert_close(c.cpu(), x, atol=2e-3, rtol=1e-4)
    assert torch.equal(c2.cpu(), c.cpu().to(torch.bfloat16))                        # the bf16 copy IS the rounded stored row
    npart = d // (64 if wide else 32)
    pairs = st.cpu().view(M, 16, 2)[:, :npart]
    xs = c.cpu().view(M, npart, d // npart)
    if wide:                                                                        # same rows as the product's tile, up to the summation order over K
        c0 = ops.gemm_resid_stats(a_in.to(DEV, torch.bfloat16), w_p.to(DEV, torch.bfloat16), b_p.to(DEV), x0.to(DEV), alpha=0.5)[0]
        torch.testing.assert_close(c, c0, atol=2e-5, rtol=1e-5)
        xin = x0.to(DEV).clone()                                                    # in place over the residual, as the layer calls it
        _lib_call_inplace(ops, a_in, w_p, b_p, xin, 0.5)
        assert torch.equal(xin, c)
    torch.testing.assert_close(pairs[..., 0], xs.sum(-1), atol=1e-3, rtol=1e-5)
    torch.testing.assert_close(pairs[..., 1], (xs * xs).sum(-1), atol=1e-2, rtol=1e-5)
    # consumer on the producer's outputs
    wf = (W * gam[None]).to(torch.bfloat16)
    colsum, cbias = wf.float().sum(-1), W @ bet + b
    want = F.gelu(F.linear(F.layer_norm(c.cpu(), (d,), gam, bet, 1e-5), W, b))
    got = ops.gemm_lnfold(c2, wf.to(DEV), colsum.to(DEV), cbias.to(DEV), st, npart, act="gelu")
    err = (got.float().cpu() - want).abs()
    assert float(err.max()) < 0.06 and float(err.mean()) < 0.006, (float(err.max()), float(err.mean()))        # bf16 operands: the same budget as LayerNorm kernel -> bf16 -> GEMM
    ref_path = F.gelu(F.linear(bfr(F.layer_norm(c.cpu(), (d,), gam, bet, 1e-5)), bfr(W), b))                  # what the un-folded kernels compute
    assert float(err.mean()) < 2.0 * float((ref_path - want).abs().mean()) + 1e-4
    # LayerNorm kernel's producer mode (masking + LN), one statistics pair, and the consumer without activation
    T, B = 100, M // 100
    lens = torch.tensor([100, 37] + [100] * (B - 2), dtype=torch.int32)
    x1 = c.cpu()[: B * T]
    y, yb, st1 = ops.layernorm_fold(c[: B * T].contiguous(), ln=(gam.to(DEV), bet.to(DEV)), lengths=lens.to(DEV), T=T)
    xm = x1.clone().view(B, T, d); xm[1, 37:] = 0
    want_y = F.layer_norm(xm.view(-1, d), (d,), gam, bet, 1e-5)
    torch.testing.assert_close(y.cpu(), want_y, atol=2e-5, rtol=1e-5)
    assert torch.equal(yb.cpu(), y.cpu().to(torch.bfloat16))
    torch.testing.assert_close(st1.cpu()[:, 0], want_y.sum(-1), atol=2e-3, rtol=1e-5)
    torch.testing.assert_close(st1.cpu()[:, 1], (want_y * want_y).sum(-1), atol=2e-2, rtol=1e-5)
    got2 = ops.gemm_lnfold(yb, wf.to(DEV), colsum.to(DEV), cbias.to(DEV), st1, 1, act="none")
    want2 = F.linear(F.layer_norm(want_y, (d,), gam, bet, 1e-5), W, b)
    e2 = (got2.float().cpu() - want2).abs()
    assert float(e2.max()) < 0.08 and float(e2.mean()) < 0.008, (float(e2.max()), float(e2.mean()))


@pytest.mark.parametrize("ratio", [0.0, 3.0, 10.0, 30.0])
def test_layernorm_fold_error_grows_with_the_row_mean_as_modelled(ratio):
    """the folded form feeds the UN-normalised bf16(x) to the GEMM and subtracts rstd mu colsum(W') afterwards, so the bf16 rounding of x (relative 2^-9 of |x| ~ |mu|)
    is amplified by |mu| / sigma against the un-folded bf16(LN(x)) — invisible on the parity fixtures (row means ~ 0), real for residual streams with a large common offset.
    Rows with mean = ratio x sigma: the folded output's error against fp32 LayerNorm -> Linear must stay within the model  e_unfolded x sqrt(1 + (c ratio)^2)  (c ~ 1: both
    errors are bf16 roundings of an operand of the same GEMM), i.e. ~ 3x at |mu| / sigma = 3, ~ 10x at 10, ~ 30x at 30 — the documented price; `HFASR_LN_FOLD=0`
    forces the un-folded form for checkpoints whose streams carry such offsets.  (The E[x^2] - mu^2 variance from fp32 partials is NOT the problem: its relative error is
    ~ 6e-8 (mu / sigma)^2, 5e-5 at ratio 30 — asserted through rstd below.)"""
    ops = _ops()
    M, d, I = 512, 512, 2048
    z = rnd(M, d, seed=61, scale=1.0)
    x = (z + ratio).contiguous()                                                    # sigma ~ 1, row mean ~ ratio
    gam, bet = 1.0 + 0.2 * rnd(d, seed=62), 0.1 * rnd(d, seed=63)
    W, b = rnd(I, d, seed=64, scale=d ** -0.5), rnd(I, seed=65, scale=0.1)
    # statistics exactly as a producer would leave them: one (sum, sum of squares) pair per row over the stored fp32 row
    st = torch.zeros(M, 32, device=DEV)
    st[:, 0] = x.sum(-1).to(DEV); st[:, 1] = (x * x).sum(-1).to(DEV)
    wf = (W * gam[None]).to(torch.bfloat16)
    colsum, cbias = wf.float().sum(-1), W @ bet + b
    want = F.linear(F.layer_norm(x, (d,), gam, bet, 1e-5), W, b)
    got = ops.gemm_lnfold(x.to(DEV, torch.bfloat16), wf.to(DEV), colsum.to(DEV), cbias.to(DEV), st, 1, act="none").float().cpu()
    unfolded = F.linear(bfr(F.layer_norm(x, (d,), gam, bet, 1e-5)), bfr(W * 1.0), b)                     # what LayerNorm kernel -> bf16 -> GEMM computes
    e_fold, e_unf = float((got - want).abs().mean()), float((unfolded - want).abs().mean())
    bound = 1.6 * e_unf * (1.0 + ratio * ratio) ** 0.5 + 2e-4
    assert e_fold < bound, (ratio, e_fold, e_unf, bound)
    if ratio >= 10.0:
        assert e_fold > 2.0 * e_unf, (ratio, e_fold, e_unf)                         # the amplification is real: this test documents it, it does not hide it
    var = (x * x).mean(-1) - x.mean(-1) ** 2
    assert float(((var - x.var(-1, unbiased=False)).abs() / x.var(-1, unbiased=False)).max()) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(8000, 512, 2048), (777, 256, 384), (130, 128, 768), (1000, 512, 4096)])
def test_gemm_128_loader_consumer_form_same_bits_as_the_pipelined_form(M, N, K):
    """gemm8p128l_kernel (round 5: four MFMA waves + four LDS-DMA waves; variant 43 — measured faster in isolation, slower inside the step, so not the product's choice) against gemm8p128p_kernel: same ring, K order and MFMA shape, so every
    output must be bit-identical — fp32 + residual, bf16, the LayerNorm-fold producer's bf16 copy and partial statistics, ragged last row tile — and right against fp32 torch."""
    from huggingface_asr_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(DEV)
    b, r = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    outs = {}
    for v in (42, 43):
        o32 = torch.full((M, N), float("nan"), device=DEV)
        ops.gemm(a, w, b, out=o32, resid=r, alpha=0.5, variant=v)
        o16 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
        ops.gemm(a, w, b, out=o16, variant=v)
        outs[v] = (o32, o16) + tuple(ops.gemm_resid_stats(a, w, b, r, alpha=0.5, variant=v))
    for x, y in zip(outs[42], outs[43]):
        assert torch.equal(x, y)
    ref = r + 0.5 * (a.float() @ w.float().t() + b)
    torch.testing.assert_close(outs[43][0], ref, atol=2e-3, rtol=1e-3)
    c, c2, st = outs[43][2:]
    assert torch.equal(c2, c.to(torch.bfloat16))
    torch.testing.assert_close(st[:, 0:2 * (N // 32):2].sum(1), c.sum(1), atol=1e-2, rtol=1e-4)
