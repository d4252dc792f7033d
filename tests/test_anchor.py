"""The fp64 anchor (oracle/anchor.py) on the CPU: the float64 run of the oracle is the same algorithm (it reproduces the fp32 run to
rounding on a well-conditioned case and replays its integer decisions), the anchored rule has teeth, and the fp32 CPU path's own
distance from the exact answer under the stress weights is what the widened tolerances of rounds 1-2 were hiding."""
import pytest
import torch

from oracle import anchor as AN
from oracle import sp_lg_oracle as O
from oracle import tie_aware as TA

torch.set_grad_enabled(False)


def _pair(h=120, w=160, seed=3):
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.avg_pool2d(torch.rand(1, 1, h + 56, w + 56, generator=g), 5, 1, 2)
    base = (base - base.min()) / (base.max() - base.min())
    size = torch.tensor([[float(w), float(h)]])
    return {"view0": {"image": base[:, :, :h, :w].contiguous(), "image_size": size},
            "view1": {"image": base[:, :, 16:16 + h, 24:24 + w].contiguous(), "image_size": size}}


@pytest.fixture(scope="module")
def case():
    data = _pair()
    W_sp, W_lg = O.make_sp_weights(0), O.make_lg_weights(7, final_gain=8.0, match_bias=2.0)
    spc, lgc = dict(nms_radius=3, detection_threshold=0.0, max_num_keypoints=128), dict(filter_threshold=0.1)
    ref = O.two_view_forward(data, W_sp, W_lg, spc, lgc, with_maps=True)
    return data, W_sp, W_lg, spc, lgc, ref, AN.TwoViewAnchor(data, W_sp, W_lg, spc, lgc)


def test_fp64_run_is_the_same_algorithm(case):
    data, W_sp, W_lg, spc, lgc, ref, anchor = case
    r64 = anchor.at(ref["keypoints0"][0], ref["keypoints1"][0])
    assert r64["log_assignment"].dtype == torch.float64 and r64["descriptors0"].dtype == torch.float64
    assert torch.equal(r64["keypoints0"].float(), ref["keypoints0"])  # the forced decisions, in the forced order
    assert AN.max_err(anchor.heatmap(0), ref["_heatmap0"]) < 1e-5
    for k in ("keypoint_scores0", "descriptors0", "descriptors1", "ref_descriptors0"):
        assert AN.max_err(ref[k], r64[k]) < 5e-5, k
    # an unforced float64 run takes (up to near-ties) the same decisions
    free = O.two_view_forward(O.cast(data, torch.float64), O.cast(W_sp, torch.float64), O.cast(W_lg, torch.float64), spc, lgc)
    assert (free["keypoints0"].float() == ref["keypoints0"]).all(-1).float().mean() > 0.9


def test_stress_weights_put_the_fp32_cpu_path_itself_outside_1e_4(case):
    """|sim| ~ 3e2 with final_proj = 8 I: the fp32 CPU log_assignment is further than 1e-4 from the exact value, its exp() is not."""
    data, W_sp, W_lg, spc, lgc, ref, anchor = case
    r64 = anchor.at(ref["keypoints0"][0], ref["keypoints1"][0])
    e_la = AN.max_err(ref["log_assignment"], r64["log_assignment"])
    e_sc = AN.score_rows(ref["log_assignment"][0], ref["matching_scores0"][0], r64["log_assignment"][0])
    print(f"|sim| max {float(ref['_sim'].abs().max()):.0f}: CPU-fp32 log_assignment is {e_la:.2e} from fp64, matching scores {e_sc:.2e}")
    assert e_la > 1e-4 and e_sc < 1e-4


def test_rule_has_teeth(case):
    data, W_sp, W_lg, spc, lgc, ref, anchor = case
    r = TA.check_two_view({k: v for k, v in ref.items() if not k.startswith("_")}, ref, 0, 128, 0.1, tag="oracle as HIP", anchor=anchor)
    assert r["floats"]["descriptors0"]["bound"] == 1e-4 and r["floats"]["descriptors0"]["err_hip"] == r["floats"]["descriptors0"]["err_cpu"]
    row = int(torch.nonzero(ref["matching_scores0"][0] > 0)[0])
    for key, at in (("descriptors0", 5), ("matching_scores0", row)):
        bad = {k: v.clone() for k, v in ref.items() if not k.startswith("_")}
        bad[key][0, at] += 3e-4
        with pytest.raises(AssertionError):
            TA.check_two_view(bad, ref, 0, 128, 0.1, tag="perturbed " + key, anchor=anchor)
    # log_assignment: the CPU path is e_cpu > 1e-4 off, so the bound is 2 e_cpu -- and 3 e_cpu must fail
    e_cpu = r["floats"]["log_assignment"]["err_cpu"]
    bad = {k: v.clone() for k, v in ref.items() if not k.startswith("_")}
    bad["log_assignment"][0, 3, 4] += 3.2 * e_cpu
    with pytest.raises(AssertionError):
        TA.check_two_view(bad, ref, 0, 128, 0.1, tag="perturbed log_assignment", anchor=anchor)
    # without an anchor the plain 1e-4 against the fp32 oracle applies
    bad = {k: v.clone() for k, v in ref.items() if not k.startswith("_")}
    bad["log_assignment"][0, 3, 4] += 2e-4
    with pytest.raises(AssertionError):
        TA.check_two_view(bad, ref, 0, 128, 0.1, tag="no anchor")


def test_adaptive_decisions_replay():
    g = torch.Generator().manual_seed(5)
    M, N = 90, 80
    size = torch.tensor([[640.0, 480.0]])
    k0 = torch.rand(1, M, 2, generator=g) * size[:, None]
    d0 = torch.nn.functional.normalize(torch.randn(1, M, 256, generator=g), dim=-1)
    data = {"keypoints0": k0, "keypoints1": k0[:, :N] * 0.9 + 3.0, "descriptors0": d0,
            "descriptors1": torch.nn.functional.normalize(d0[:, :N] + 0.03 * torch.randn(1, N, 256, generator=g), dim=-1),
            "view0": {"image_size": size}, "view1": {"image_size": size}}
    W = O.make_lg_weights(11, structured=True, final_gain=3.0)
    ref = O.lightglue_forward(data, W, filter_threshold=0.1, width_confidence=0.55)
    assert ref["log_assignment"].shape[1] - 1 < M, "the case must prune"
    res, r64 = AN.matcher_floats(O.lightglue_forward, ref, ref, data, W, "replay", ("log_assignment", "ref_descriptors0"),
                                 dict(filter_threshold=0.1, width_confidence=0.55), replay=ref["_decisions"])
    assert r64["log_assignment"].shape == ref["log_assignment"].shape and torch.equal(r64["_ind0"], ref["_ind0"])
    assert res["log_assignment"]["err_hip"] < 1e-4
