"""Phase breakdown of the fused feed-forward kernel (csrc/ffn_f16.hip) from in-kernel s_memtime stamps (diagnostic
instantiation, debug tap "ffn.stamps").  Prints the median cycles per phase over the workgroups of the last launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "glue-factory_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gluefactory_amd import get_model  # noqa: E402
from oracle import sp_lg_oracle as O  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, K = 32, 1024
g = torch.Generator().manual_seed(0)
m = get_model("gluefactory_amd.lightglue")({"filter_threshold": 0.1, "return_log_assignment": False}).eval()
m.load_state_dict(O.make_lg_weights(7, final_gain=8.0, match_bias=2.0))
size = torch.tensor([[640.0, 480.0]] * B)
data = {"keypoints0": torch.rand(B, K, 2, generator=g) * 470 + 1, "keypoints1": torch.rand(B, K, 2, generator=g) * 470 + 1,
        "descriptors0": torch.nn.functional.normalize(torch.randn(B, K, 256, generator=g), dim=-1),
        "descriptors1": torch.nn.functional.normalize(torch.randn(B, K, 256, generator=g), dim=-1),
        "view0": {"image_size": size}, "view1": {"image_size": size}}
to = lambda x: {k: (to(v) if isinstance(v, dict) else v.to(dev)) for k, v in x.items()}
d = to(data)
eng = m.engine(dev)
for _ in range(20):  # warm clocks
    m(d)
eng.debug_request("ffn.stamps")
for req in [a for a in sys.argv[1:] if not a.startswith('--')]:  # e.g. lg.ffn8 = the eight-wave kernel
    eng.debug_request(req)
for _ in range(3):
    m(d)
torch.cuda.synchronize()
n = (2 * B * K + 127) // 128
st = eng.debug_read("ffn.stamps", n * 8).cpu().reshape(n, 8)
if "--per-wave" in sys.argv:  # library built with -DMG_FFN_EXP=64
    print("phase 3 waits + barriers per wave (median over workgroups): " + ", ".join(f"{v:.0f}" for v in st.median(0).values.tolist()))
    sys.exit(0)
names = ["start", "phase 1 (ffn.0 GEMM)", "phase 2 (bias, LayerNorm, GELU)", "split + first quarter", "phase 3 (ffn.3 GEMM)", "epilogue"]
med = st.median(0).values
print(f"{n} workgroups; median cycles since workgroup start: " + ", ".join(f"{v:.0f}" for v in med[:6].tolist()))
for i in range(1, 6):
    print(f"  {names[i]:36s} {float(med[i] - med[i - 1]):9.0f} cycles ({float((med[i] - med[i - 1]) / med[5]) * 100:5.1f} %)")
print(f"  phase 3, wave 0: vmcnt wait + LDS wait + barrier of the 16 intervals {float(med[6]):.0f} cycles, of which the vmcnt wait {float(med[7]):.0f} (eight-wave kernel only)")
half = n // 2
for nm, sl in (("workgroups 0..%d (first round of the grid)" % (half - 1), slice(0, half)), ("workgroups %d..%d (second round)" % (half, n - 1), slice(half, n))):
    m2 = st[sl].median(0).values
    print(f"  {nm}: " + ", ".join(f"{float(m2[i] - m2[i - 1]):.0f}" for i in range(1, 6)))
