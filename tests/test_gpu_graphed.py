"""hipGraph replay (gluefactory_amd/graphed.py) returns what the eager forward returns, bit for bit: on inputs the graph was not captured
on (this is what caught ROCm's unordered memset node, sp_post.hip: zero_ints_kernel), for more than one input signature, and across a
re-allocation of the native workspace (mg_workspace_generation)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _pair(seed, B, H, W, dev):
    g = torch.Generator().manual_seed(seed)
    img = F.avg_pool2d(torch.rand(2 * B, 1, H + 8, W + 8, generator=g), 9, 1)
    img = ((img - img.amin()) / (img.amax() - img.amin())).to(dev)
    size = torch.tensor([[float(W), float(H)]] * B, device=dev)
    return {"view0": {"image": img[:B], "image_size": size}, "view1": {"image": img[B:], "image_size": size}}


def _same(a, b):
    if a.dtype.is_floating_point:
        return torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8))
    return torch.equal(a, b)


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_graphed_pipeline_equals_eager(precision):
    from gluefactory_amd import _native
    from gluefactory_amd.graphed import GraphedModule
    from gluefactory_amd.two_view_pipeline import TwoViewPipeline

    dev = torch.device("cuda:0")
    pipe = TwoViewPipeline({"extractor": {"name": "gluefactory_nonfree.superpoint", "max_num_keypoints": 512, "nms_radius": 3, "detection_threshold": 0.0,
                                          "force_num_keypoints": True, "precision": precision},
                            "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1, "precision": precision}}).eval()
    pipe.extractor.load_state_dict(O.make_sp_weights(0))
    pipe.matcher.load_state_dict(O.make_lg_weights(7, structured=True, final_gain=3.0, match_bias=2.0))
    g = GraphedModule(pipe)
    gen0 = None
    # (seed, B, H, W): the first signature three times with different images, a second, larger one in between (its workspace grows: the
    # first graph must be re-captured afterwards, not replayed into freed buffers)
    for seed, B, H, W in [(1, 1, 240, 320), (2, 1, 240, 320), (3, 2, 256, 384), (4, 1, 240, 320)]:
        data = _pair(seed, B, H, W, dev)
        eager = {k: v.clone() for k, v in pipe(data).items() if torch.is_tensor(v)}
        out = g(data)
        torch.cuda.synchronize()
        assert set(out) == set(eager), (sorted(out), sorted(eager))
        for k in sorted(eager):
            assert _same(out[k], eager[k]), (k, seed)
        assert bool(torch.isfinite(out["descriptors0"]).all()) and out["keypoints0"].shape == (B, 512, 2)
        gen0 = gen0 if gen0 is not None else int(_native.lib().mg_workspace_generation())
    assert len(g._graphs) == 2 and int(_native.lib().mg_workspace_generation()) > gen0


def test_graphed_follows_load_state_dict_and_precision():
    """New weights (or another precision) between two graphed calls: the replay must not run the launches recorded for the old weight
    buffers -- neither before the eager path has uploaded the new set (stale result) nor after (freed buffers)."""
    from gluefactory_amd.graphed import GraphedModule
    from gluefactory_amd.two_view_pipeline import TwoViewPipeline

    dev = torch.device("cuda:0")
    pipe = TwoViewPipeline({"extractor": {"name": "gluefactory_nonfree.superpoint", "max_num_keypoints": 256, "nms_radius": 3, "detection_threshold": 0.0,
                                          "force_num_keypoints": True},
                            "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1}}).eval()
    pipe.extractor.load_state_dict(O.make_sp_weights(0))
    pipe.matcher.load_state_dict(O.make_lg_weights(7, structured=True, final_gain=3.0, match_bias=2.0))
    g = GraphedModule(pipe)
    data = _pair(5, 1, 240, 320, dev)
    first = {k: v.clone() for k, v in g(data).items() if torch.is_tensor(v)}
    # (a) new weights, replay BEFORE any eager forward has uploaded them
    pipe.extractor.load_state_dict(O.make_sp_weights(1))
    out = {k: v.clone() for k, v in g(data).items() if torch.is_tensor(v)}
    eager = {k: v.clone() for k, v in pipe(data).items() if torch.is_tensor(v)}
    assert all(_same(out[k], eager[k]) for k in eager)
    assert not _same(out["keypoint_scores0"], first["keypoint_scores0"])
    # (b) new weights, an eager forward re-tiles them (the old device buffers are freed), then the graphed call
    pipe.matcher.load_state_dict(O.make_lg_weights(8, structured=True, final_gain=3.0, match_bias=2.0))
    eager = {k: v.clone() for k, v in pipe(data).items() if torch.is_tensor(v)}
    out = g(data)
    torch.cuda.synchronize()
    assert all(_same(out[k], eager[k]) for k in eager)


def test_graphed_needs_device_inputs():
    from gluefactory_amd import get_model
    from gluefactory_amd.graphed import GraphedModule

    m = get_model("gluefactory_amd.superpoint")({"max_num_keypoints": 16, "force_num_keypoints": True}).eval()
    with pytest.raises(RuntimeError, match="device-resident"):
        GraphedModule(m)({"image": torch.zeros(1, 1, 64, 64)})
