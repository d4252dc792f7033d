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
for _ in range(3):
    m(d)
torch.cuda.synchronize()
n = (2 * B * K + 127) // 128
st = eng.debug_read("ffn.stamps", n * 8).cpu().reshape(n, 8)
names = ["start", "phase 1 (ffn.0 GEMM)", "phase 2 (bias, LayerNorm, GELU)", "split + first quarter", "phase 3 (ffn.3 GEMM)", "epilogue"]
med = st.median(0).values
print(f"{n} workgroups; median cycles since workgroup start: " + ", ".join(f"{v:.0f}" for v in med[:6].tolist()))
for i in range(1, 6):
    print(f"  {names[i]:36s} {float(med[i] - med[i - 1]):9.0f} cycles ({float((med[i] - med[i - 1]) / med[5]) * 100:5.1f} %)")
