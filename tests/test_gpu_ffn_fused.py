"""The fused feed-forward kernel (csrc/ffn_f16.hip: ffn.0 -> LayerNorm -> GELU -> ffn.3 -> residual in one launch) against the
three-launch form it replaces (GEMM, ln_gelu, GEMM -- still in the library behind a debug request) and against the oracle.
Same arithmetic class (split-fp16 products, fp32 accumulation); the summation order inside LayerNorm and the erf
evaluation differ, so floats agree to ~1e-6 and the assignments must be identical."""
import pytest
import torch

import tie_aware as TA
from oracle import anchor as AN
from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _data(B, M, N, seed):
    g = torch.Generator().manual_seed(seed)
    size = torch.tensor([[640.0, 480.0]] * B)
    k0 = torch.rand(B, M, 2, generator=g) * size[:, None] * 0.98 + 0.5
    d0 = torch.nn.functional.normalize(torch.randn(B, M, 256, generator=g), dim=-1)
    perm = torch.stack([torch.randperm(max(M, N), generator=g)[:N] % M for _ in range(B)])
    k1 = torch.gather(k0, 1, perm[..., None].expand(-1, -1, 2)) * 0.9 + 2.0 * torch.randn(B, N, 2, generator=g)
    d1 = torch.nn.functional.normalize(torch.gather(d0, 1, perm[..., None].expand(-1, -1, 256)) + 0.03 * torch.randn(B, N, 256, generator=g), dim=-1)
    return {"keypoints0": k0, "keypoints1": k1, "descriptors0": d0, "descriptors1": d1, "view0": {"image_size": size}, "view1": {"image_size": size}}


@pytest.mark.parametrize("B,M,N", [(2, 1024, 1024), (1, 130, 61), (3, 129, 517), (1, 1, 5)])
def test_fused_ffn_equals_three_launch_form(B, M, N):
    from gluefactory_amd import get_model

    dev = torch.device("cuda:0")
    W = O.make_lg_weights(5, structured=True, final_gain=3.0, match_bias=2.0)
    data = _data(B, M, N, B + M)
    to = lambda x: {k: (to(v) if isinstance(v, dict) else v.to(dev)) for k, v in x.items()}
    outs = []
    for unfused in (False, True):
        m = get_model("gluefactory_amd.lightglue")({"filter_threshold": 0.1, "precision": "f16x3"}).eval()
        m.load_state_dict(W)
        eng = m.engine(dev)
        for tap in ("lg.self0", "lg.layer0", "lg.layer8"):
            eng.debug_request(tap)
        if unfused:
            eng.debug_request("lg.unfused_ffn")
        p = m(to(data))
        T = B * (M + N)
        taps = {t: eng.debug_read(t, T * 256).cpu() for t in ("lg.self0", "lg.layer0", "lg.layer8")}
        outs.append(({k: v.cpu() for k, v in p.items()}, taps))
    (pf, tf), (pu, tu) = outs
    for t in tf:
        err = (tf[t] - tu[t]).abs().max().item()
        print(f"B{B} M{M} N{N} {t}: max |fused - unfused| {err:.2e} (|x| max {tu[t].abs().max().item():.2f})")
        torch.testing.assert_close(tf[t], tu[t], rtol=0, atol=2e-5)
    assert torch.equal(pf["matches0"], pu["matches0"]) and torch.equal(pf["matches1"], pu["matches1"])
    ref = O.lightglue_forward(data, W, filter_threshold=0.1)
    # both forms against the float64 oracle: 1e-4, or 2 x the fp32 CPU path's own error where that exceeds 1e-4 (oracle/anchor.py)
    _, r64 = AN.matcher_floats(O.lightglue_forward, pf, ref, data, W, f"fused B{B} M{M} N{N}", ("ref_descriptors0", "log_assignment"), dict(filter_threshold=0.1))
    AN.matcher_floats(O.lightglue_forward, pu, ref, data, W, f"three-launch B{B} M{M} N{N}", ("ref_descriptors0", "log_assignment"), r64=r64)
    for b in range(B):
        eps = TA.la_eps(pf["log_assignment"][b], ref["log_assignment"][b])
        r = TA.check_matches(pf["matches0"][b], ref["matches0"][b], ref["log_assignment"][b], 0.1, eps)
        print(f"  pair {b}: vs oracle {r['differing']}/{r['compared']} matches differ (near-ties), desc err "
              f"{(pf['ref_descriptors0'][b, 0] - ref['ref_descriptors0'][b, 0]).abs().max().item():.2e}")
