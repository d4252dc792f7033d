"""The LDS-DMA convolutions (conv1ab_dma_kernel, conv3x3_dma_kernel) against the register-staged kernels they replace: same
accumulation orders, so every encoder layer must be BIT-identical (SH8 words), on interior, border and ragged tiles, pooled
and not, gray and RGB input, one and several tiles per workgroup."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

NAMES = ["conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convDa"]


def run_layers(img, dev, staged):
    from gluefactory_amd import get_model

    if staged:
        os.environ["MG_CONV_STAGED"] = "1"
    else:
        os.environ.pop("MG_CONV_STAGED", None)
    try:
        m = get_model("gluefactory_amd.superpoint")({"max_num_keypoints": 16, "nms_radius": 3, "detection_threshold": 0.0, "precision": "f16x3",
                                                      "force_num_keypoints": True, "dense_outputs": True}).eval()
        m.load_state_dict(O.make_sp_weights(0))
        eng = m.engine(dev)
        for n in NAMES:
            eng.debug_request("sp." + n)
        m({"image": img.to(dev)})
        B, _, H, W = img.shape
        out = {}
        h, w = H, W
        for n in NAMES:
            cout = {"1": 64, "2": 64, "3": 128, "4": 128, "P": 256, "D": 256}[n[4]]
            if n in ("conv1b", "conv2b", "conv3b"):
                h, w = h // 2, w // 2
            out[n] = eng.debug_read("sp." + n, B * h * w * cout).cpu().view(torch.int32).clone()
        return out
    finally:
        os.environ.pop("MG_CONV_STAGED", None)


@pytest.mark.parametrize("shape", [(2, 480, 640, 1), (3, 64, 96, 3), (1, 136, 200, 1), (2, 248, 72, 3), (5, 120, 168, 1), (9, 240, 320, 3)])
def test_dma_conv_bit_identical_to_staged(shape):
    dev = torch.device("cuda:0")
    B, H, W, C = shape
    g = torch.Generator().manual_seed(H * 7 + W)
    img = F.avg_pool2d(torch.rand(B, C, H + 4, W + 4, generator=g), 5, 1)
    img = (img - img.amin()) / (img.amax() - img.amin())
    a = run_layers(img, dev, staged=True)
    b = run_layers(img, dev, staged=False)
    for n in NAMES:
        same = torch.equal(a[n], b[n])
        if not same:
            d = (a[n] != b[n]).nonzero().flatten()
            print(n, "differing words", d.numel(), "of", a[n].numel(), "first", d[:8].tolist())
        assert same, n
