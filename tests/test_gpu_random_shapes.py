"""Seeded random shapes through the whole native path against the oracle: odd image sizes (not multiples of the conv
tiles), batch sizes 1-3, ragged keypoint counts, both SuperPoint layouts.  Floats within the north-star tolerance;
keypoint sets / matches compared with the same near-tie allowance as the fixed-size tests."""
import pytest
import torch

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
    p = pipe(_to(data, dev))
    # oracle per pair (the reference batches only equal keypoint counts)
    kp_same, m_same, n = 0.0, 0.0, 0
    for b in range(B):
        one = {"view0": {"image": img[b:b + 1], "image_size": size[:1]}, "view1": {"image": img[B + b:B + b + 1], "image_size": size[:1]}}
        ref = O.two_view_forward(one, W_sp, W_lg, spc, dict(filter_threshold=0.1))
        for i in ("0", "1"):
            nk = ref["keypoints" + i].shape[1]
            got = p["keypoints" + i][b, :nk].cpu()
            a, c = set(map(tuple, got.tolist())), set(map(tuple, ref["keypoints" + i][0].tolist()))
            assert len(a & c) >= nk - max(2, nk // 50), (seed, b, i, len(a & c), nk, H, W, K, variant)
            kp_same += float((got == ref["keypoints" + i][0]).all(-1).float().mean())
            n += 1
        nk0 = ref["matches0"].shape[1]
        if all((p["keypoints" + i][b, : ref["keypoints" + i].shape[1]].cpu() == ref["keypoints" + i][0]).all() for i in ("0", "1")):
            assert torch.equal(p["matches0"][b, :nk0].cpu(), ref["matches0"][0]), (seed, b, H, W, K, variant)
            torch.testing.assert_close(p["matching_scores0"][b, :nk0].cpu(), ref["matching_scores0"][0], rtol=0, atol=2e-4)
            m_same += 1
    print(f"seed {seed}: B{B} {W}x{H} C{C} K{K} r{spc['nms_radius']} {variant}: same-rank keypoints {kp_same / n:.4f}, pairs with identical keypoints {int(m_same)}/{B}")
    assert kp_same / n > 0.95
