"""Adaptive depth / width LightGlue on the GPU (mg_lg_forward_adaptive).

Point pruning (`width_confidence`) is pinned by tests/golden/lightglue_adaptive.npz, produced by the reference's own LightGlue;
early stopping (`depth_confidence`) raises AttributeError in the reference fork (lightglue.py:556 reads an undefined attribute), so it
is checked against the oracle's restatement of the evident intent (thresholds of `confidence_threshold()`, = upstream LightGlue)."""
import pytest
import torch

import tie_aware as TA
from oracle import anchor as AN
from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _data(g, tag):
    return {"keypoints0": g[tag + ".keypoints0"], "keypoints1": g[tag + ".keypoints1"], "descriptors0": g[tag + ".descriptors0"],
            "descriptors1": g[tag + ".descriptors1"], "view0": {"image_size": g[tag + ".size"]}, "view1": {"image_size": g[tag + ".size"]}}


def _to(x, dev):
    return {k: (_to(v, dev) if isinstance(v, dict) else v.to(dev)) for k, v in x.items()}


def _module(W, precision, **conf):
    from gluefactory_amd import get_model

    m = get_model("gluefactory_amd.lightglue")({"filter_threshold": 0.1, "precision": precision, **conf}).eval()
    m.load_state_dict(W)
    return m


def _compare(p, ref, tag, data, W, conf, decisions):
    """pruning decisions and surviving sets identical; assignments identical up to proven near-ties; floats within 1e-4 of the
    float64 oracle replaying the same decisions (or 2 x the fp32 CPU path's own error where that exceeds 1e-4, oracle/anchor.py)"""
    assert torch.equal(p["prune0"].cpu(), ref["prune0"]) and torch.equal(p["prune1"].cpu(), ref["prune1"]), f"{tag}: pruning decisions differ"
    assert p["log_assignment"].shape == ref["log_assignment"].shape
    la, S = p["log_assignment"][0].cpu(), ref["log_assignment"][0]
    AN.matcher_floats(O.lightglue_forward, p, ref, data, W, tag, ("log_assignment", "ref_descriptors0", "ref_descriptors1"),
                      dict(filter_threshold=0.1, **conf), replay=decisions)
    eps = TA.la_eps(la, S)
    # matches live in the ORIGINAL index space; compare there directly (pruned points are -1 on both sides)
    for k in ("matches0", "matches1"):
        got, want = p[k][0].cpu(), ref[k][0]
        diff = torch.nonzero(got != want).flatten().tolist()
        assert len(diff) <= 2, (tag, k, len(diff))  # any difference must be a near-tie of the oracle's decision:
        if diff:
            surv0 = torch.nonzero(ref["prune0"][0] == ref["prune0"][0].max()).flatten()
            surv1 = torch.nonzero(ref["prune1"][0] == ref["prune1"][0].max()).flatten()
            inv = {int(v): i for i, v in enumerate((surv0 if k == "matches0" else surv1).tolist())}
            other = {int(v): i for i, v in enumerate((surv1 if k == "matches0" else surv0).tolist())}
            m_row, m_col, m_th = TA.match_margins(S if k == "matches0" else S.t(), 0.1)
            for i in diff:
                for j in {int(got[i]), int(want[i])} - {-1}:
                    assert min(float(m_row[inv[i], other[j]]), float(m_col[inv[i], other[j]])) < eps or float(m_th[inv[i], other[j]]) < eps, (tag, k, i, j)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_point_pruning_golden(golden, precision):
    g = golden("lightglue_adaptive")
    dev = torch.device("cuda:0")
    for tag, wseed in (("a", 11), ("b", 12), ("c", 13)):
        W = O.make_lg_weights(wseed, structured=True, final_gain=3.0, match_bias=0.0)
        wc = float(g[tag + ".width_confidence"])
        m = _module(W, precision, width_confidence=wc)
        p = m(_to(_data(g, tag), dev))
        ref = {k: g[f"{tag}.{k}"] for k in ("matches0", "matches1", "matching_scores0", "matching_scores1", "log_assignment", "prune0", "prune1",
                                            "ref_descriptors0", "ref_descriptors1")}
        assert p["prune0"].dtype == torch.int64 and p["matches0"].shape == ref["matches0"].shape
        # the fp32 oracle (== the golden, tests/test_oracle_golden.py) supplies the pruning decisions the float64 run replays
        orc = O.lightglue_forward(_data(g, tag), W, filter_threshold=0.1, width_confidence=wc)
        assert torch.equal(orc["prune0"], ref["prune0"])
        _compare(p, ref, f"{tag}/{precision}", _data(g, tag), W, dict(width_confidence=wc), orc["_decisions"])
        print(f"[{precision}] {tag}: survivors {p['log_assignment'].shape[1] - 1} x {p['log_assignment'].shape[2] - 1} of "
              f"{p['matches0'].shape[1]} x {p['matches1'].shape[1]}, matches {(p['matches0'] > -1).sum().item()}, layers {m.last_layers_run}")
        assert m.last_layers_run == 9


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("mode", ["depth", "depth+width"])
def test_early_stopping_vs_oracle(golden, precision, mode):
    g = golden("lightglue_adaptive")
    dev = torch.device("cuda:0")
    data = _data(g, "a")
    W = dict(O.make_lg_weights(11, structured=True, final_gain=3.0, match_bias=0.0))
    # seeded token-confidence heads sit near 0.5 (never confident): make layers 0-2 half confident and layer 3 fully confident
    for i in range(3):
        W[f"token_confidence.{i}.token.0.weight"] = W[f"token_confidence.{i}.token.0.weight"] * 40.0
        W[f"token_confidence.{i}.token.0.bias"] = W[f"token_confidence.{i}.token.0.bias"] + 1.5
    W["token_confidence.3.token.0.bias"] = W["token_confidence.3.token.0.bias"] + 6.0
    conf = {"depth_confidence": 0.95, "width_confidence": 0.57 if mode == "depth+width" else -1}
    ref = O.lightglue_forward(data, W, filter_threshold=0.1, **conf)
    assert ref["_layers_run"] == 4  # stops after layer index 3
    m = _module(W, precision, **conf)
    p = m(_to(data, dev))
    assert m.last_layers_run == 4
    if mode == "depth":  # no pruning: prune* = n_layers, as the reference reports them (:509-511)
        assert torch.equal(p["prune0"].cpu(), torch.full((1, 200), 9.0)) and p["log_assignment"].shape == (1, 201, 181)
        la, S = p["log_assignment"][0].cpu(), ref["log_assignment"][0]
        AN.matcher_floats(O.lightglue_forward, p, ref, data, W, f"depth/{precision}", ("log_assignment", "ref_descriptors0"), dict(filter_threshold=0.1, **conf),
                          replay=ref["_decisions"])
        r = TA.check_matches(p["matches0"][0].cpu(), ref["matches0"][0], S, 0.1, TA.la_eps(la, S))
        assert r["compared"] == 200
    else:
        _compare(p, ref, f"depth+width/{precision}", data, W, conf, ref["_decisions"])
        assert p["log_assignment"].shape[1] - 1 < 200
    # with both switches off the adaptive entry is never taken and the result equals the plain forward
    plain = _module(W, precision)(_to(data, dev))
    assert plain["log_assignment"].shape == (1, 201, 181) and torch.equal(plain["prune0"].cpu(), torch.full((1, 200), 9.0))
