"""Seeded random shapes through the whole native path against the oracle: odd image sizes (not multiples of the conv
tiles), batch sizes 1-3, ragged keypoint counts, both SuperPoint layouts.  Floats within the north-star tolerance of the float64 oracle (oracle/anchor.py);
keypoint indices / matches compared with the tie-aware exact rule of oracle/tie_aware.py (no percentage thresholds)."""
import pytest
import torch

import tie_aware as TA
from oracle import anchor as AN
from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
ATOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _to(x, dev):
    return {k: (_to(v, dev) if isinstance(v, dict) else v.to(dev)) for k, v in x.items()}


@pytest.mark.parametrize("seed", range(6))
def test_two_view_random_shapes(dev, seed):
    from gluefactory_amd.two_view_pipeline import TwoViewPipeline

    g = torch.Generator().manual_seed(1000 + seed)
    B = int(torch.randint(1, 4, (1,), generator=g))
    H = int(torch.randint(9, 40, (1,), generator=g)) * 8 + int(torch.randint(0, 8, (1,), generator=g))   # 72 .. 319, any remainder mod 8
    W = int(torch.randint(9, 48, (1,), generator=g)) * 8 + int(torch.randint(0, 8, (1,), generator=g))
    K = int(torch.randint(32, 300, (1,), generator=g))
    variant = "open" if seed % 2 else "magicleap"
    C = 3 if seed % 3 == 0 else 1
    img = torch.nn.functional.avg_pool2d(torch.rand(2 * B, C, H + 4, W + 4, generator=g), 5, 1)
    img = (img - img.amin()) / (img.amax() - img.amin())
    size = torch.tensor([[float(W), float(H)]] * B)
    W_sp = O.make_sp_weights(seed, variant)
    W_lg = O.make_lg_weights(20 + seed, structured=True, final_gain=8.0, match_bias=2.0)
    spc = dict(variant=variant, nms_radius=int(torch.randint(1, 5, (1,), generator=g)), detection_threshold=0.0, max_num_keypoints=K)
    data = {"view0": {"image": img[:B], "image_size": size}, "view1": {"image": img[B:], "image_size": size}}
    name = "extractors.superpoint_open" if variant == "open" else "gluefactory_nonfree.superpoint"
    pipe = TwoViewPipeline({"extractor": {"name": name, "force_num_keypoints": True, **{k: v for k, v in spc.items() if k != "variant"}},
                            "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1}}).eval()
    pipe.extractor.load_state_dict(W_sp)
    pipe.matcher.load_state_dict(W_lg)
    p = {k: v.cpu() for k, v in pipe(_to(data, dev)).items()}
    solo = TwoViewPipeline({"extractor": {"name": name, **{k: v for k, v in spc.items() if k != "variant"}}, "fuse_views": False,
                            "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1}}).eval()
    solo.extractor.load_state_dict(W_sp)
    solo.matcher.load_state_dict(W_lg)
    # oracle per pair (the reference batches only equal keypoint counts)
    n_full = 0
    for b in range(B):
        one = {"view0": {"image": img[b:b + 1], "image_size": size[:1]}, "view1": {"image": img[B + b:B + b + 1], "image_size": size[:1]}}
        ref = O.two_view_forward(one, W_sp, W_lg, spc, dict(filter_threshold=0.1), with_maps=True)
        anchor = AN.TwoViewAnchor(one, W_sp, W_lg, spc, dict(filter_threshold=0.1))
        counts = [ref["keypoints" + i].shape[1] for i in ("0", "1")]
        tag = f"seed {seed} B{B} {W}x{H} C{C} K{K} r{spc['nms_radius']} {variant}"
        if counts == [K, K]:
            TA.check_two_view(p, ref, b, K, 0.1, tag=tag, anchor=anchor)
            n_full += 1
        else:
            # fewer than K detections: the batched rows are zero-padded (a documented difference from the reference's random
            # padding, superpoint.py:304-317) and the matcher sees the padding, so the batch is checked up to the extractor ...
            for i in ("0", "1"):
                n = counts[int(i)]
                r = TA.check_keypoints(p["keypoints" + i][b, :n], ref["keypoints" + i][0], ref["keypoint_scores" + i][0], ref["_nms_scores" + i][0], k_max=K)
                assert bool((p["keypoint_scores" + i][b, n:] == 0).all()) and bool((p["keypoints" + i][b, n:] == 0).all())
                ok = r["perm"] >= 0
                torch.testing.assert_close(p["descriptors" + i][b, :n][ok], ref["descriptors" + i][0][r["perm"][ok]], rtol=0, atol=ATOL)
            # ... and the pair alone (rows trimmed to the detected count, as the reference returns them) end to end
            q = {k: v.cpu() for k, v in solo(_to(one, dev)).items()}
            TA.check_two_view(q, ref, 0, K, 0.1, tag=tag + " (alone)", anchor=anchor)
        for i in ("0", "1"):
            n = counts[int(i)]
            torch.testing.assert_close(p["keypoint_scores" + i][b, :n].sort().values, ref["keypoint_scores" + i][0].sort().values, rtol=0, atol=ATOL)
    print(f"seed {seed}: B{B} {W}x{H} C{C} K{K} r{spc['nms_radius']} {variant}: {n_full}/{B} pairs with K keypoints in both images")
