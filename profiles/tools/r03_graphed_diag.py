"""Step-by-step check of hipGraph replay against the eager forward (modes: sp | lg | pipe), as run for profiles/r03_g_*diag*.txt:
   python profiles/tools/r03_graphed_diag.py sp
With the per-image counters of the detect stage cleared by hipMemsetAsync (a memset NODE in the graph) the sp mode returned stale
candidates on new inputs and once faulted; cleared by a kernel (sp_post.hip: zero_ints_kernel) every replay is bit-identical.
"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for q in (ROOT, os.path.join(ROOT, "glue-factory_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, q)
import torch
import torch.nn.functional as F
from oracle import sp_lg_oracle as O
from gluefactory_amd.graphed import GraphedModule
from gluefactory_amd.two_view_pipeline import TwoViewPipeline
from gluefactory_amd import _native

def say(*a):
    print(*a, file=sys.stderr, flush=True)

def pair(seed, B, H, W, dev):
    g = torch.Generator().manual_seed(seed)
    img = F.avg_pool2d(torch.rand(2 * B, 1, H + 8, W + 8, generator=g), 9, 1)
    img = ((img - img.amin()) / (img.amax() - img.amin())).to(dev)
    size = torch.tensor([[float(W), float(H)]] * B, device=dev)
    return {"view0": {"image": img[:B], "image_size": size}, "view1": {"image": img[B:], "image_size": size}}

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "pipe"
pipe = TwoViewPipeline({"extractor": {"name": "gluefactory_nonfree.superpoint", "max_num_keypoints": 512, "nms_radius": 3, "detection_threshold": 0.0,
                                      "force_num_keypoints": True, "precision": "f16x3"},
                        "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1, "precision": "f16x3"}}).eval()
pipe.extractor.load_state_dict(O.make_sp_weights(0))
pipe.matcher.load_state_dict(O.make_lg_weights(7, structured=True, final_gain=3.0, match_bias=2.0))
mod = pipe.extractor if mode == "sp" else pipe.matcher if mode == "lg" else pipe
g = GraphedModule(mod)
for i, (seed, B, H, W) in enumerate([(1, 1, 240, 320), (2, 1, 240, 320), (3, 2, 128, 192), (4, 1, 240, 320)]):
    data = pair(seed, B, H, W, dev)
    if mode == "sp":
        data = {"image": data["view0"]["image"], "image_size": data["view0"]["image_size"]}
    if mode == "lg":
        gg = torch.Generator().manual_seed(seed)
        K = 512
        data = {"keypoints0": (torch.rand(B, K, 2, generator=gg) * torch.tensor([W, H])).to(dev), "keypoints1": (torch.rand(B, K, 2, generator=gg) * torch.tensor([W, H])).to(dev),
                "descriptors0": F.normalize(torch.randn(B, K, 256, generator=gg), dim=-1).to(dev), "descriptors1": F.normalize(torch.randn(B, K, 256, generator=gg), dim=-1).to(dev),
                "view0": {"image_size": data["view0"]["image_size"]}, "view1": {"image_size": data["view1"]["image_size"]}}
    say(f"case {i}: eager ...  generation {_native.lib().mg_workspace_generation()}")
    e = {k: v.clone() for k, v in mod(data).items() if torch.is_tensor(v)}
    torch.cuda.synchronize()
    say(f"case {i}: eager done; graphed ... generation {_native.lib().mg_workspace_generation()}, graphs {len(g._graphs)}")
    out = g(data)
    say(f"case {i}: replay issued; sync ...")
    torch.cuda.synchronize()
    say(f"case {i}: synced; " + ", ".join(f"{k}: {'==' if torch.equal(out[k], e[k]) else 'DIFF %.3g (%d of %d)' % (float((out[k].double() - e[k].double()).abs().max()), int((out[k] != e[k]).sum()), out[k].numel())}" for k in sorted(e)))
say("done")
