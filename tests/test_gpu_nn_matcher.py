"""GPU parity of the native nearest-neighbour matcher (mg_nn_forward) -- SURVEY 8f rank 4.

Goldens come from gluefactory/models/matchers/nearest_neighbor_matcher.py itself.  Decisions that hinge on the last
bits of a dot product (a threshold met with equality, two neighbours equally close) may legitimately differ; every
mismatch is therefore checked to be such a knife-edge case on the reference's own similarity matrix.
"""
import pytest
import torch

from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
NN_CONFS = ({}, {"ratio_thresh": 0.9}, {"distance_thresh": 0.8}, {"ratio_thresh": 0.95, "distance_thresh": 0.9, "mutual_check": False})
EDGE = 2e-5


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _module(conf, precision):
    from gluefactory_amd import get_model

    return get_model("gluefactory_amd.nearest_neighbor_matcher")({**conf, "precision": precision}).eval()


def _knife_edge(sim, conf):
    """(B,M) bool: rows of `sim` whose NN decision is within EDGE of flipping."""
    v, _ = sim.topk(2, dim=-1)
    d0, d1 = 2 * (1 - v[..., 0]), 2 * (1 - v[..., 1])
    edge = (v[..., 0] - v[..., 1]) < EDGE
    if conf.get("ratio_thresh"):
        edge |= (d0 - conf["ratio_thresh"] ** 2 * d1).abs() < EDGE
    if conf.get("distance_thresh"):
        edge |= (d0 - conf["distance_thresh"] ** 2).abs() < EDGE
    return edge


def _check_matches(got0, got1, ref0, ref1, sim, conf):
    e0, e1 = _knife_edge(sim, conf), _knife_edge(sim.transpose(1, 2), conf)
    bad0, bad1 = got0 != ref0, got1 != ref1
    if conf.get("mutual_check", True):  # a flipped candidate on one side also flips its partner
        frac = (bad0.float().mean() + bad1.float().mean()).item() / 2
        assert frac <= 0.02 and (bad0.sum() + bad1.sum()) <= 4 * (e0.sum() + e1.sum()), (frac, int(e0.sum()), int(e1.sum()))
    else:
        assert not (bad0 & ~e0).any() and not (bad1 & ~e1).any()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_nn_matcher_golden(dev, golden, precision):
    g = golden("nn_matcher")
    for tag in ("a", "b", "c"):
        d0, d1 = g[tag + ".descriptors0"], g[tag + ".descriptors1"]
        sim_ref = g[tag + ".similarity"]
        for ci, conf in enumerate(NN_CONFS):
            p = _module(conf, precision)({"descriptors0": d0.to(dev), "descriptors1": d1.to(dev)})
            _check_matches(p["matches0"].cpu(), p["matches1"].cpu(), g[f"{tag}.c{ci}.matches0"], g[f"{tag}.c{ci}.matches1"], sim_ref, conf)
            assert p["matches0"].dtype == torch.int64
            assert torch.equal(p["matching_scores0"].cpu(), (p["matches0"].cpu() > -1).float())
            assert torch.equal(p["matching_scores1"].cpu(), (p["matches1"].cpu() > -1).float())
            torch.testing.assert_close(p["similarity"].cpu(), sim_ref, rtol=0, atol=2e-6)
            torch.testing.assert_close(p["log_assignment"].cpu(), g[tag + ".log_assignment"], rtol=0, atol=2e-5)
            if ci == 0:
                assert torch.equal(p["matches0"].cpu(), g[f"{tag}.c0.matches0"]) and torch.equal(p["matches1"].cpu(), g[f"{tag}.c0.matches1"])


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("B,M,N,D", [(2, 1024, 1024, 256), (1, 2048, 1999, 256), (3, 129, 517, 128)])
def test_nn_matcher_vs_oracle_fullsize(dev, B, M, N, D, precision):
    g = torch.Generator().manual_seed(B + M + N)
    d0 = torch.nn.functional.normalize(torch.randn(B, M, D, generator=g), dim=-1)
    perm = torch.stack([torch.randperm(max(M, N), generator=g)[:N] % M for _ in range(B)])
    d1 = torch.gather(d0, 1, perm[..., None].expand(-1, -1, D)) + (0.7 / D ** 0.5) * torch.randn(B, N, D, generator=g)
    d1 = torch.nn.functional.normalize(d1, dim=-1)
    conf = {"ratio_thresh": 0.9, "distance_thresh": 1.0}
    ref = O.nn_forward({"descriptors0": d0, "descriptors1": d1}, **conf)
    p = _module(conf, precision)({"descriptors0": d0.to(dev), "descriptors1": d1.to(dev)})
    torch.testing.assert_close(p["similarity"].cpu(), ref["similarity"], rtol=0, atol=3e-6)
    torch.testing.assert_close(p["log_assignment"].cpu(), ref["log_assignment"], rtol=0, atol=5e-5)
    _check_matches(p["matches0"].cpu(), p["matches1"].cpu(), ref["matches0"], ref["matches1"], ref["similarity"], conf)
    agree = (p["matches0"].cpu() == ref["matches0"]).float().mean().item()
    print(f"[{precision}] B{B} M{M} N{N} D{D}: matches0 agreement {agree:.5f}, matched {(ref['matches0'] > -1).float().mean().item():.3f}")
    # oracle-independent: mutual consistency and the thresholds hold on the returned similarity
    m0, m1, sim = p["matches0"], p["matches1"], p["similarity"]
    ok = m0 > -1
    idx = torch.arange(M, device=dev)[None].expand(B, -1)
    assert torch.all(torch.gather(m1, 1, m0.clamp(min=0))[ok] == idx[ok])
    best = torch.gather(sim, 2, m0.clamp(min=0)[..., None])[..., 0]
    assert torch.all(best[ok] == sim.max(dim=2).values[ok])
    assert torch.all((2 * (1 - best))[ok] <= 1.0 + 1e-6)


def test_nn_matcher_contract(dev):
    d = torch.nn.functional.normalize(torch.randn(1, 8, 32), dim=-1).to(dev)
    m = _module({"ratio_thresh": 0.8}, "f16x3")
    with pytest.raises(AssertionError):
        m({"descriptors0": d})
    with pytest.raises(RuntimeError):
        m({"descriptors0": d[:, :1], "descriptors1": d})  # ratio test with a single candidate: the reference's topk(2) raises too
    # any descriptor width: rows are zero-padded to the GEMM's 16-step internally
    d20 = torch.nn.functional.normalize(d[..., :20], dim=-1).contiguous()
    ref = O.nn_forward({"descriptors0": d20.cpu(), "descriptors1": d20.cpu().flip(1)})
    got = _module({}, "f16x3")({"descriptors0": d20, "descriptors1": d20.flip(1).contiguous()})
    assert torch.equal(got["matches0"].cpu(), ref["matches0"])
    torch.testing.assert_close(got["similarity"].cpu(), ref["similarity"], rtol=0, atol=2e-6)
    p = _module({}, "f16x3")({"descriptors0": d, "descriptors1": d})  # identical sets: identity matching
    assert torch.equal(p["matches0"].cpu(), torch.arange(8)[None])
