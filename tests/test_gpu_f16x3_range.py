"""Unit tests of the two matrix cores over their real operand range, against fp64 -- in BOTH arithmetic modes, through the
stand-alone C-ABI entries (mg_test_linear / mg_test_attention honour mg_set_precision).

f16x3 holds an operand as x = hi + lo (two fp16 halves) and drops the lo.lo product.  Proven representation error
(round-to-nearest halves, fp16 has 11 significant bits and subnormal spacing 2^-24):

    |x - (hi + lo)| <= max(2^-22 |x|, 2^-25)            for |x| <= 65504  (lo is subnormal for |x| < 0.125: absolute floor 3e-8)

so one product a.w is off by at most  |w| da + |a| dw + 2^-22 |a w|  (+ second-order terms), and a dot product adds the
fp32 accumulation error of the MFMA chain.  The tests assert the measured error against exactly that bound, element by
element, for operands spanning 1e-7 .. 1e3, weights of sigma 1e-3, and soft-max rows whose logits span +-40 over 2048 keys.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

U22, U24, FLOOR = 2.0 ** -22, 2.0 ** -24, 2.0 ** -25


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module", params=["f32", "f16x3"])
def engine(request, dev):
    from gluefactory_amd import _native

    e = _native.Engine(dev)
    e.set_precision(request.param)
    e.mode = request.param
    return e


def rep_err(x, mode):
    """Bound on |x - stored(x)| of one operand (fp64 tensor in, fp64 tensor out)."""
    if mode == "f32":
        return torch.zeros_like(x)
    return torch.maximum(x.abs() * U22, torch.full_like(x, FLOOR))


def log_uniform(shape, lo, hi, g):
    """sign * 10^U(lo, hi)"""
    mag = 10.0 ** (lo + (hi - lo) * torch.rand(shape, generator=g))
    return mag * (torch.randint(0, 2, shape, generator=g) * 2 - 1)


def linear_bound(A, W, mode):
    """Element-wise bound on |C_hip - C_fp64| for C = A W^T (A, W fp32 tensors as handed to the library)."""
    A64, W64 = A.double(), W.double()
    K = A.shape[1]
    absprod = A64.abs() @ W64.abs().t()
    rep = rep_err(A64, mode) @ W64.abs().t() + A64.abs() @ rep_err(W64, mode).t()
    dropped = U22 * absprod if mode == "f16x3" else 0.0
    acc = (K / 16 + 8) * U24 * absprod if mode == "f16x3" else (K + 2) * U24 * absprod  # fp32 accumulation chain
    return rep + dropped + acc + 1e-30


CASES = {
    "unit": lambda g, M, N, K: (torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)),
    "wide_1e-7_1e3": lambda g, M, N, K: (log_uniform((M, K), -7, 3, g), log_uniform((N, K), -7, 0, g)),
    "small_weights_sigma_1e-3": lambda g, M, N, K: (torch.rand(M, K, generator=g), 1e-3 * torch.randn(N, K, generator=g)),
    "tiny_activations_1e-6": lambda g, M, N, K: (1e-6 * torch.randn(M, K, generator=g), 0.05 * torch.randn(N, K, generator=g)),
    "large_1e3": lambda g, M, N, K: (1e3 * torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)),
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (513, 256, 512)])
def test_linear_against_fp64_bound(engine, dev, case, M, N, K):
    g = torch.Generator().manual_seed(sum(map(ord, case)) + M)
    A, W = CASES[case](g, M, N, K)
    A, W = A.float(), W.float()
    C = engine.test_linear(A.to(dev), W).cpu().double()
    ref = A.double() @ W.double().t()
    err = (C - ref).abs()
    bound = linear_bound(A, W, engine.mode)
    ratio = float((err / bound).max())
    rel = float(err.max() / ref.abs().max())
    print(f"[{engine.mode}] {case} {M}x{N}x{K}: max err {float(err.max()):.3e} (|C| max {float(ref.abs().max()):.3e}, rel {rel:.2e}), err/bound max {ratio:.3f}")
    assert bool(torch.isfinite(C).all())
    assert ratio <= 1.0, f"error exceeds the representation + accumulation bound by x{ratio:.2f}"


def test_linear_fp16_range_limit(engine, dev):
    """|x| > 65504 cannot be held in the fp16 halves: f16x3 documents that range -- the f32 mode has no such limit."""
    A = torch.full((128, 32), 1.0e5)
    W = torch.full((32, 32), 1.0e-3)
    C = engine.test_linear(A.to(dev), W).cpu()
    if engine.mode == "f32":
        torch.testing.assert_close(C, torch.full((128, 32), 3200.0), rtol=1e-6, atol=0)
    else:
        assert not bool(torch.isfinite(C).all())


def attention_ref(q, k, v, B, M, N, cross):
    T0 = B * M
    out = torch.empty_like(q, dtype=torch.float64)
    bound_terms = []
    for b in range(B):
        seqs = [(slice(b * M, (b + 1) * M), slice(b * M, (b + 1) * M)), (slice(T0 + b * N, T0 + (b + 1) * N), slice(T0 + b * N, T0 + (b + 1) * N))]
        if cross:
            seqs = [(seqs[0][0], seqs[1][0]), (seqs[1][0], seqs[0][0])]
        for qs, ks in seqs:
            for h in range(4):
                c = slice(64 * h, 64 * h + 64)
                l = q[qs, c].double() @ k[ks, c].double().t()
                p = torch.softmax(l, -1)
                out[qs, c] = p @ v[ks, c].double()
                bound_terms.append((qs, ks, c, p))
    return out, bound_terms


def attention_bound(q, k, v, out, terms, mode):
    """First-order bound on |o_hip - o_fp64|: logit errors perturb the soft-max (do = sum_j p_j (v_j - o) dl_j), plus the
    representation of P (2^-21), of V, of the output row and the fp32 accumulation of both contractions."""
    bound = torch.zeros_like(out)
    LOG2E = 1.4426950408889634
    for qs, ks, c, p in terms:
        q64, k64, v64 = q[qs, c].double(), k[ks, c].double(), v[ks, c].double()
        if mode == "f16x3":
            q2 = q64 * LOG2E
            dl = (rep_err(q2, mode) + q2.abs() * U24) @ k64.abs().t() + q2.abs() @ rep_err(k64, mode).t() + (U22 + 12 * U24) * (q2.abs() @ k64.abs().t())
            dl = dl / LOG2E + 2 * U24 * (q64 @ k64.t()).abs()  # back to natural units; the exp2 argument itself is rounded to fp32
            p_rep, v_rep = 2.0 ** -21, rep_err(v64, mode)
        else:
            dl = (64 + 2) * U24 * (q64.abs() @ k64.abs().t()) + 2 * U24 * (q64 @ k64.t()).abs()
            p_rep, v_rep = U24, torch.zeros_like(v64)
        o = out[qs, c]
        dl_max = dl.max(1, keepdim=True).values
        spread = p @ v64.abs() + o.abs()                        # sum_j p_j |v_j - o| <= sum p|v| + |o|
        b = 2 * dl_max * spread + (p_rep + 40 * U24) * (p @ v64.abs()) + p @ v_rep + 2048 * 2.0 ** -31 * v64.abs().max()
        if mode == "f16x3":
            b = b + rep_err(o, mode)                             # the kernel's output rows are SH8
        bound[qs, c] = b
    return bound + 1e-30


def make_qkv(kind, B, M, N, g):
    T = B * (M + N)
    v = torch.randn(T, 256, generator=g)
    if kind == "diffuse":  # logits O(1): every key contributes
        q, k = 0.35 * torch.randn(T, 256, generator=g), torch.randn(T, 256, generator=g)
    elif kind == "peaked_pm40":  # each query aligned with one key: logits reach +40, the bulk sits at 0 +- 13 -> span ~ +-40 per row
        k = torch.randn(T, 256, generator=g)
        pick = torch.randint(0, T, (T,), generator=g)
        q = torch.zeros(T, 256)
        for h in range(4):
            c = slice(64 * h, 64 * h + 64)
            kk = k[pick][:, c]
            q[:, c] = kk * (40.0 / kk.pow(2).sum(-1, keepdim=True)) + 1.6 * torch.randn(T, 64, generator=g)
    elif kind.startswith("flat_tail"):  # two dominant keys per row, every other key `depth` (natural units) below them: a long flat
        depth = float(kind.split("_")[-1])  # tail inside the subnormal zone of the fp16 halves of P (2^-14 .. 2^-24 without the 2^7 shift)
        k = torch.randn(T, 256, generator=g) * 0.01
        q = torch.randn(T, 256, generator=g) * 0.01
        k[:, 0::64] = 1.0
        q[:, 0::64] = -depth
        k[0::max(M, N) // 2, 0::64] = 0.0   # the dominant keys (logit ~0, the rest ~ -depth)
    elif kind == "wide_v":
        q, k = 0.35 * torch.randn(T, 256, generator=g), torch.randn(T, 256, generator=g)
        v = log_uniform((T, 256), -6, 3, g)
    else:
        raise KeyError(kind)
    return q.float(), k.float(), v.float()


@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("kind,B,M,N", [("diffuse", 2, 300, 517), ("peaked_pm40", 1, 2048, 2048), ("flat_tail_11", 1, 2048, 2048), ("flat_tail_17", 1, 2048, 2048), ("wide_v", 1, 1024, 999),
                                        # round 4 (attention_pw_f16.hip: software-pipelined key loop with a prologue, a peeled last iteration and a masked partial
                                        # tile): one to four key tiles, full and partial, fewer queries than a wave / a workgroup holds
                                        ("diffuse", 3, 5, 31), ("diffuse", 2, 32, 33), ("diffuse", 2, 70, 64), ("diffuse", 1, 257, 65), ("peaked_pm40", 2, 96, 97), ("peaked_pm40", 1, 128, 130)])
def test_attention_against_fp64_bound(engine, dev, kind, B, M, N, cross):
    g = torch.Generator().manual_seed(len(kind) + M)
    q, k, v = make_qkv(kind, B, M, N, g)
    got = engine.test_attention(q.to(dev), k.to(dev), v.to(dev), B, M, N, cross).cpu().double()
    ref, terms = attention_ref(q, k, v, B, M, N, cross)
    span = max(float((q[qs, c].double() @ k[ks, c].double().t()).abs().max()) for qs, ks, c, _ in terms[:4])
    err = (got - ref).abs()
    bound = attention_bound(q, k, v, ref, terms, engine.mode)
    ratio = float((err / bound).max())
    print(f"[{engine.mode}] {kind} cross={cross}: max |logit| {span:.1f}, max err {float(err.max()):.3e} (|o| max {float(ref.abs().max()):.2e}), err/bound max {ratio:.3f}")
    assert bool(torch.isfinite(got).all())
    assert ratio <= 1.0, f"error exceeds the first-order bound by x{ratio:.2f}"
    if kind != "wide_v":  # unit-scale values: the north-star tolerance itself
        assert float(err.max()) < 1e-4
