# Determinism probe of the fused conv1a + conv1b kernel (conv_f16.hip; round 2: conv1ab_dma_kernel, where it isolated "hazard 1" of
# DESIGN.md section 4; round 3: conv1ab_mfma_kernel): identity conv1b weights expose the conv1a channels.
# Runs the kernel four times with identity conv1b weights and compares the runs with each other (determinism).
import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/glue-factory_amd")
import torch.nn.functional as F
from oracle import sp_lg_oracle as O
from test_gpu_sp_stages import decode_sh8
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, H, W, C = 2, 480, 640, 1
g = torch.Generator().manual_seed(H * 7 + W)
img = F.avg_pool2d(torch.rand(B, C, H + 4, W + 4, generator=g), 5, 1)
img = (img - img.amin()) / (img.amax() - img.amin())
from gluefactory_amd import get_model
Wt = O.make_sp_weights(0)
w = torch.zeros_like(Wt["conv1b.weight"])
PERM = int(os.environ.get("DIAG_PERM", "0"))
for c in range(64): w[c, c ^ PERM, 1, 1] = 1.0
Wt["conv1b.weight"] = w
Wt["conv1b.bias"] = torch.zeros_like(Wt["conv1b.bias"])
m = get_model("gluefactory_amd.superpoint")({"max_num_keypoints": 16, "nms_radius": 3, "detection_threshold": 0.0, "precision": "f16x3", "force_num_keypoints": True, "dense_outputs": True}).eval()
m.load_state_dict(Wt)
eng = m.engine(dev)
eng.debug_request("sp.conv1b")
outs = []
for r in range(4):
    m({"image": img.to(dev)})
    outs.append(decode_sh8(eng.debug_read("sp.conv1b", B * (H // 2) * (W // 2) * 64).cpu().clone(), (B, H // 2, W // 2, 64)))
for r in range(1, 4):
    d = (outs[r] - outs[0]).abs()
    print("run", r, "vs 0: max", d.max().item(), "frac", (d > 0).float().mean().item(), "per ch parity even/odd", (d[..., 0::2] > 0).float().mean().item(), (d[..., 1::2] > 0).float().mean().item())
