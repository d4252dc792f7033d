"""SURVEY 8f rank 2 on real GPU output: a batched SuperPoint + LightGlue result goes through the batched exporter
(gluefactory_amd.export, the reference's h5 layout), is read back by its inverse (gluefactory_amd.export.read_predictions, checked on
the CPU against what the reference's own CacheLoader reads), and the cached features drive the pipeline's `cache` path
(two_view_pipeline.py:62-70).  h5py is not installed in this image: the h5 calls go to an in-memory stand-in."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import sp_lg_oracle as O

sys.path.insert(0, os.path.dirname(__file__))
from test_export_h5 import _H5  # noqa: E402

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_export_read_back_and_cached_features(tmp_path):
    from gluefactory_amd.export import export_predictions, read_predictions
    from gluefactory_amd.two_view_pipeline import TwoViewPipeline

    dev = torch.device("cuda:0")
    z = np.load(os.path.join(ROOT, "tests", "golden", "two_view_boat.npz"))
    g0, g1 = (torch.from_numpy(z[k]).float() / 255.0 for k in ("gray0", "gray1"))
    B, H, W, K = 4, 240, 320, 256
    im0 = torch.stack([g0[40 * i: 40 * i + H, 60 * i: 60 * i + W] for i in range(B)])[:, None].contiguous()
    im1 = torch.stack([g1[30 * i: 30 * i + H, 50 * i: 50 * i + W] for i in range(B)])[:, None].contiguous()
    conf = {"extractor": {"name": "gluefactory_nonfree.superpoint", "max_num_keypoints": K, "nms_radius": 3, "detection_threshold": 0.0, "force_num_keypoints": True},
            "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1}}
    pipe = TwoViewPipeline(conf).eval()
    pipe.extractor.load_state_dict(O.make_sp_weights(0))
    pipe.matcher.load_state_dict(O.make_lg_weights(7, structured=True, final_gain=3.0, match_bias=2.0))
    size = torch.tensor([[float(W), float(H)]] * B)
    scales0 = torch.tensor([[0.5, 2.0]] * B)   # powers of two: the 1/scales -> x scales round trip is exact
    scales1 = torch.tensor([[4.0, 0.25]] * B)
    names = [f"seq/pair{i}.ppm" for i in range(B)]
    batch = {"name": names, "view0": {"image": im0, "image_size": size, "scales": scales0}, "view1": {"image": im1, "image_size": size, "scales": scales1}}
    keys = ["keypoints0", "keypoints1", "keypoint_scores0", "keypoint_scores1", "descriptors0", "descriptors1", "matches0", "matches1",
            "matching_scores0", "matching_scores1"]
    path = tmp_path / "pred.h5"
    export_predictions([batch], pipe, path, keys=keys, h5=_H5)
    assert sorted(_H5.files[str(path)]) == sorted(names)
    direct = {k: v.cpu() for k, v in pipe({k: ({kk: vv.to(dev) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in batch.items()}).items()}

    for b, name in enumerate(names):
        got = read_predictions(path, [name], {"view0": {"scales": scales0[b:b + 1]}, "view1": {"scales": scales1[b:b + 1]}}, h5=_H5)[0]
        assert sorted(got) == sorted(keys)
        for k in keys:
            want = direct[k][b]
            n = got[k].shape[0]
            assert n == K  # every image of the boat crops yields K keypoints at threshold 0
            assert got[k].dtype == want.dtype and torch.equal(got[k], want), (name, k)
    # cached features instead of the extractor (the reference's `cache` + allow_no_extract path), one pair
    cached = TwoViewPipeline({**conf, "allow_no_extract": True}).eval()
    cached.extractor.load_state_dict(O.make_sp_weights(0))
    cached.matcher.load_state_dict(O.make_lg_weights(7, structured=True, final_gain=3.0, match_bias=2.0))
    b = 2
    feats = read_predictions(path, [names[b]], {"view0": {"scales": scales0[b:b + 1]}, "view1": {"scales": scales1[b:b + 1]}}, h5=_H5)[0]
    view = lambda i: {"image_size": size[b:b + 1].to(dev),
                      "cache": {k: feats[k + str(i)][None].to(dev) for k in ("keypoints", "keypoint_scores", "descriptors")}}
    p = cached({"view0": view(0), "view1": view(1)})
    for k in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
        assert torch.equal(p[k][0].cpu(), direct[k][b]), k
