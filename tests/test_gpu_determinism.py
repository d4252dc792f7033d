"""Run-to-run determinism of the whole hot path: the same batch three times, every output bit-identical.

Nothing on this path uses atomics on floats or an order-dependent reduction, so any difference is a hardware-level
hazard (round 2 found one: packed-fp32 FMAs with an op_sel broadcast issued between a wave's MFMAs returned wrong low
halves now and then -- conv_f16.hip, conv1ab_dma_kernel).  configs[1] geometry, 8 pairs, both precisions."""
import pytest
import torch
import torch.nn.functional as F

from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_two_view_pipeline_is_deterministic(precision):
    from gluefactory_amd.two_view_pipeline import TwoViewPipeline

    dev = torch.device("cuda:0")
    B, H, W, K = 8, 480, 640, 1024
    g = torch.Generator().manual_seed(11)
    img = F.avg_pool2d(torch.rand(2 * B, 1, H + 8, W + 8, generator=g), 9, 1)
    img = ((img - img.amin()) / (img.amax() - img.amin())).to(dev)
    pipe = TwoViewPipeline({"extractor": {"name": "gluefactory_nonfree.superpoint", "max_num_keypoints": K, "nms_radius": 3, "detection_threshold": 0.0,
                                          "force_num_keypoints": True, "precision": precision},
                            "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1, "precision": precision}}).eval()
    pipe.extractor.load_state_dict(O.make_sp_weights(0))
    pipe.matcher.load_state_dict(O.make_lg_weights(7, structured=True, final_gain=3.0, match_bias=2.0))
    data = {"view0": {"image": img[:B]}, "view1": {"image": img[B:]}}
    runs = []
    for _ in range(3):
        out = pipe(data)
        torch.cuda.synchronize()
        runs.append({k: v.clone() for k, v in out.items() if torch.is_tensor(v)})
    keys = sorted(runs[0])
    assert {"keypoints0", "descriptors0", "matches0", "matching_scores0", "log_assignment"} <= set(keys)
    for r in (1, 2):
        for k in keys:
            a, b = runs[0][k], runs[r][k]
            same = torch.equal(a.view(torch.uint8) if a.dtype.is_floating_point else a, b.view(torch.uint8) if b.dtype.is_floating_point else b) \
                if a.dtype != torch.bool else torch.equal(a, b)
            if not same:
                d = (a.float() - b.float()).abs()
                print(k, "run", r, "differs: max", d.max().item(), "count", int((d > 0).sum()), "of", d.numel())
            assert same, (k, r)
