"""The LDS-DMA convolutions against the register-staged kernels they replace.

conv3x3_dma_kernel (debug request "sp.conv_dma32" since the tap-pair kernel ships) keeps the accumulation order of
conv3x3_f16x3_kernel, so given the same input every encoder layer must be BIT-identical (SH8 words), on interior, border and ragged
tiles, pooled and not, one and several tiles per workgroup: both runs take conv1b from the register-staged conv1ab kernel (debug
request "sp.conv1ab_staged"), so all later layers see identical inputs.

conv3x3_pair_kernel (16x16x32 MFMA, tap pairs as the contraction index: the shipped 3x3 kernel) sums the taps in another order, so it
is held to the staged kernels within fp32 rounding of the sums (2e-5 of the layer's largest activation, through all eight layers) and
must be bit-identical from run to run.

conv1ab_mfma_kernel (round 3) evaluates conv1a on the matrix pipe with split-fp16 operands where the staged kernel uses fp32 FMAs: its
conv1b output is held to 1e-4 ABSOLUTE of the float64 oracle (oracle/anchor.py; activations reach |x| ~ 10) on the same shapes, gray
and RGB, and must be bit-identical from run to run."""
import pytest
import torch
import torch.nn.functional as F

from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

NAMES = ["conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convDa"]


def run_layers(img, dev, request, names=NAMES):
    """SH8 words of the requested layers; `request`: the debug request that selects the kernels (None = the shipped ones)."""
    from gluefactory_amd import get_model

    m = get_model("gluefactory_amd.superpoint")({"max_num_keypoints": 16, "nms_radius": 3, "detection_threshold": 0.0, "precision": "f16x3",
                                                  "force_num_keypoints": True, "dense_outputs": True}).eval()
    m.load_state_dict(O.make_sp_weights(0))
    eng = m.engine(dev)
    for r in ([request] if isinstance(request, str) else request or []):
        eng.debug_request(r)
    for n in names:
        eng.debug_request("sp." + n)
    m({"image": img.to(dev)})
    B, _, H, W = img.shape
    out = {}
    h, w = H, W
    for n in NAMES:
        cout = {"1": 64, "2": 64, "3": 128, "4": 128, "P": 256, "D": 256}[n[4]]
        if n in ("conv1b", "conv2b", "conv3b"):
            h, w = h // 2, w // 2
        if n in names:
            out[n] = eng.debug_read("sp." + n, B * h * w * cout).cpu().view(torch.int32).clone()
    return out


def _image(shape):
    B, H, W, C = shape
    g = torch.Generator().manual_seed(H * 7 + W)
    img = F.avg_pool2d(torch.rand(B, C, H + 4, W + 4, generator=g), 5, 1)
    return (img - img.amin()) / (img.amax() - img.amin())


SHAPES = [(2, 480, 640, 1), (3, 64, 96, 3), (1, 136, 200, 1), (2, 248, 72, 3), (5, 120, 168, 1), (9, 240, 320, 3)]


@pytest.mark.parametrize("shape", SHAPES)
def test_dma_conv_bit_identical_to_staged(shape):
    dev = torch.device("cuda:0")
    img = _image(shape)
    a = run_layers(img, dev, "sp.conv_staged")      # every layer on the register-staged kernels
    b = run_layers(img, dev, ("sp.conv1ab_staged", "sp.conv_dma32"))   # conv1ab staged (identical conv1b), every 3x3 layer on conv3x3_dma_kernel
    for n in NAMES:
        same = torch.equal(a[n], b[n])
        if not same:
            d = (a[n] != b[n]).nonzero().flatten()
            print(n, "differing words", d.numel(), "of", a[n].numel(), "first", d[:8].tolist())
        assert same, n


@pytest.mark.parametrize("shape", SHAPES)
def test_pair_conv_vs_staged(shape):
    from test_gpu_sp_stages import decode_sh8

    dev = torch.device("cuda:0")
    B, H, W, C = shape
    img = _image(shape)
    a = run_layers(img, dev, "sp.conv_staged")
    runs = [run_layers(img, dev, "sp.conv1ab_staged") for _ in range(2)]   # conv1b staged, every 3x3 layer on conv3x3_pair_kernel
    h, w = H, W
    for n in NAMES:
        assert torch.equal(runs[0][n], runs[1][n]), f"{n}: conv3x3_pair_kernel is not deterministic"
        cout = {"1": 64, "2": 64, "3": 128, "4": 128, "P": 256, "D": 256}[n[4]]
        if n in ("conv1b", "conv2b", "conv3b"):
            h, w = h // 2, w // 2
        x = decode_sh8(a[n].view(torch.float32), (B, h, w, cout)).double()
        y = decode_sh8(runs[0][n].view(torch.float32), (B, h, w, cout)).double()
        err, top = float((x - y).abs().max()), float(x.abs().max())
        print(f"  {n}: |pair - staged| max {err:.2e}  (|x| max {top:.2f})")
        assert err <= 2e-5 * max(top, 1.0), (n, err, top)


@pytest.mark.parametrize("shape", SHAPES)
def test_conv1ab_mfma_vs_float64(shape):
    from oracle import anchor as AN
    from test_gpu_sp_stages import decode_sh8

    dev = torch.device("cuda:0")
    B, H, W, C = shape
    img = _image(shape)
    Wt = O.make_sp_weights(0)

    def stage1(x, Wd):
        x = O._conv_relu(O._conv_relu(O.to_gray(x), Wd, "conv1a"), Wd, "conv1b")
        return F.max_pool2d(x, 2, 2).permute(0, 2, 3, 1).contiguous()

    ref32, ref64 = stage1(img, Wt), stage1(img.double(), O.cast(Wt, torch.float64))
    runs = [run_layers(img, dev, None, names=("conv1b",))["conv1b"] for _ in range(2)]
    assert torch.equal(runs[0], runs[1]), "conv1ab_mfma_kernel is not deterministic"
    got = decode_sh8(runs[0].view(torch.float32), ref32.shape)
    AN.anchored_close(got, ref32, ref64, f"conv1ab (MFMA conv1a) {shape}, |x| max {float(ref32.abs().max()):.1f}")
    staged = decode_sh8(run_layers(img, dev, "sp.conv1ab_staged", names=("conv1b",))["conv1b"].view(torch.float32), ref32.shape)
    print(f"  fp32-conv1a staged kernel: |err vs fp64| {AN.max_err(staged, ref64):.2e}")
