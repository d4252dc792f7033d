"""The benchmarked batch shapes, end to end, in both arithmetic modes.

BASELINE.json configs[1]: 32 pairs of 640x480 images, 1024 keypoints -- what bench.py times.
BASELINE.json configs[2]: the per-GPU shard of the 256-pair job = 32 pairs of 1024x1024 square-padded images whose valid
region is 1024x768 (`image_size`), 2048 keypoints.

Two properties, no thresholds:
  * batch invariance: pair i inside the batch is BIT-IDENTICAL (every output tensor) to pair i run alone;
  * pairs {0, 15, 31} (configs[1]) / pair 7 (configs[2]) agree with the CPU oracle under the tie-aware exact rule of
    oracle/tie_aware.py (indices may differ only at proven near-ties of the oracle's own decisions), floats within 1e-4 of the
    oracle run in float64 (oracle/anchor.py).
Inputs: SURVEY 8(d)'s tiling of the boat pair with per-item circular shifts (tests/golden/two_view_boat.npz holds the images).
"""
import os

import numpy as np
import pytest
import torch

import tie_aware as TA
from oracle import anchor as AN
from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("keypoints0", "keypoints1", "keypoint_scores0", "keypoint_scores1", "descriptors0", "descriptors1", "matches0", "matches1",
        "matching_scores0", "matching_scores1", "log_assignment")


def boat_pairs(B, H, W, Hv, seed):
    """(B,1,H,W) x 2: boat1 / boat2 resized to (Hv, W), circularly shifted per item, zero-padded to H rows."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "two_view_boat.npz"))
    g = torch.Generator().manual_seed(seed)
    views = []
    for key in ("gray0", "gray1"):
        base = torch.nn.functional.interpolate(torch.from_numpy(z[key]).float()[None, None] / 255.0, size=(Hv, W), mode="bilinear", align_corners=False)[0, 0]
        out = torch.zeros(B, 1, H, W)
        for i in range(B):
            dy, dx = (int(v) for v in torch.randint(0, 10 ** 6, (2,), generator=g))
            out[i, 0, :Hv] = torch.roll(base, shifts=(dy % Hv, dx % W), dims=(0, 1))
        views.append(out)
    return views


def run_case(dev, precision, B, H, W, Hv, K, oracle_pairs):
    from gluefactory_amd.two_view_pipeline import TwoViewPipeline

    W_sp, W_lg = O.make_sp_weights(0), O.make_lg_weights(7, structured=True, final_gain=3.0, match_bias=2.0)
    spc = dict(nms_radius=3, detection_threshold=0.0, max_num_keypoints=K)
    pipe = TwoViewPipeline({"extractor": {"name": "gluefactory_nonfree.superpoint", "remove_borders": 4, "force_num_keypoints": True, "precision": precision, **spc},
                            "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1, "precision": precision}}).eval()
    pipe.extractor.load_state_dict(W_sp)
    pipe.matcher.load_state_dict(W_lg)
    im0, im1 = boat_pairs(B, H, W, Hv, 77)
    size = torch.tensor([[float(W), float(Hv)]] * B)
    d0, d1, ds = im0.to(dev), im1.to(dev), size.to(dev)
    full = pipe({"view0": {"image": d0, "image_size": ds}, "view1": {"image": d1, "image_size": ds}})
    full = {k: full[k] for k in KEYS}
    # batch invariance, every pair, every output, bit for bit
    for i in range(B):
        one = pipe({"view0": {"image": d0[i:i + 1], "image_size": ds[i:i + 1]}, "view1": {"image": d1[i:i + 1], "image_size": ds[i:i + 1]}})
        for k in KEYS:
            assert torch.equal(one[k][0], full[k][i]), f"pair {i}: '{k}' differs between the batch of {B} and the pair alone"
    p = {k: v.cpu() for k, v in full.items()}
    assert bool((p["keypoint_scores0"] > 0).all()) and bool((p["keypoint_scores1"] > 0).all()), "an image produced fewer than K keypoints"
    for i in oracle_pairs:
        data = {"view0": {"image": im0[i:i + 1], "image_size": size[:1]}, "view1": {"image": im1[i:i + 1], "image_size": size[:1]}}
        # (oracle and float64 anchor of a pair are shared by the two arithmetic modes: same seeded inputs)
        ref = AN.memo(("b32.ref", H, W, Hv, K, i), lambda: O.two_view_forward(data, W_sp, W_lg, spc, dict(filter_threshold=0.1), with_maps=True))
        anchor = AN.memo(("b32.anchor", H, W, Hv, K, i), lambda: AN.TwoViewAnchor(data, W_sp, W_lg, spc, dict(filter_threshold=0.1)))  # float64: what 1e-4 is measured against
        r = TA.check_two_view(p, ref, i, K, 0.1, tag=f"{W}x{H} K{K} B{B} {precision}", anchor=anchor)
        # moderate |sim| (final_proj = 3 I): the fp32 CPU path is itself within 1e-4 of fp64, so the bound applied was north_star's 1e-4
        assert all(v["bound"] == 1e-4 for k, v in r["floats"].items() if k != "log_assignment"), r["floats"]
        for s in ("0", "1"):
            rr = TA.check_keypoints(p["keypoints" + s][i], ref["keypoints" + s][0], ref["keypoint_scores" + s][0], ref["_nms_scores" + s][0],
                                    got_scores=p["keypoint_scores" + s][i], k_max=K)
            ok = rr["perm"] >= 0
            torch.testing.assert_close(p["descriptors" + s][i][ok], ref["descriptors" + s][0][rr["perm"][ok]], rtol=0, atol=1e-4)
        assert r["m0"]["compared"] > K // 2
    return p


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_config1_batch32(precision):
    """configs[1]: the batch bench.py times (32 pairs, 640x480, 1024 keypoints)."""
    run_case(torch.device("cuda:0"), precision, 32, 480, 640, 480, 1024, (0, 15, 31))


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_config2_shard_batch32(precision):
    """configs[2]: one rank's 32-pair shard (1024x1024 square-padded, valid 1024x768, 2048 keypoints)."""
    run_case(torch.device("cuda:0"), precision, 32, 1024, 1024, 768, 2048, (7,))
