"""hipGraph replay of a forward pass, for the launch-bound regime (one or a few pairs per call).

At 32 pairs per step the path is kernel-bound (the ~100 launches of a step are queued far ahead of the GPU).  At ONE pair per call
launch overhead is the obvious suspect -- and measured it is not the bound either: replaying the step as one graph takes a VGA pair
from 3.89 to 3.79 ms (4 pairs: 4.75 -> 4.65 ms; 32 pairs: no change; `profiles/r03_g_graph_vs_eager.json`), because the fused kernels
are tiled for throughput and a single pair leaves each of them a serial loop on a handful of workgroups (the fused FFN runs 16
workgroups x ~90 us per launch whatever the batch).  The wrapper is kept for hosts whose CPU is the bottleneck (a replay costs one
launch call instead of ~100).  `GraphedModule` captures one forward of a module of this package into a HIP graph (torch.cuda.CUDAGraph is
hipGraph on ROCm: the native launches go to `torch.cuda.current_stream()`, which is the capture stream during capture, and torch's
graph-private memory pool keeps every tensor the forward allocates at a fixed address) and replays it per call:

    g = GraphedModule(pipe)            # pipe: a TwoViewPipeline / extractor / matcher of this package, weights loaded
    out = g(data)                      # first call with a new input signature: 3 eager warm-up calls + capture; later calls: copy-in + replay

Rules (they are what stream capture demands, nothing of ours):
  * the forward must not synchronise with the host: use `force_num_keypoints: True` (or `sync_free`) in the extractor, and no adaptive
    LightGlue (its depth / width decisions are read by the host every layer);
  * inputs are copied into static tensors (one device copy per input tensor and call), outputs are the SAME tensors on every call --
    consume or `.clone()` them before the next call;
  * one graph per input signature (shapes and dtypes of all tensors in `data`), kept until `reset()` -- or until something a recorded
    launch depends on has changed: the native library re-allocated workspace (a larger shape arrived, through this wrapper or through
    an eager call), released a model's device weights for a re-tiling, switched precision or took a new debug request (all counted by
    `mg_workspace_generation()`), or a module below received new weights / another precision that it has not uploaded yet (uploads
    happen lazily in the eager forward, which a replay never runs: `_module_state()`).  Every graph remembers both values of its
    capture and is re-captured when either has moved.
  * verified for the SuperPoint + LightGlue path (tests/test_gpu_graphed.py).  One pitfall of ROCm 7.2 found on the way: a
    hipMemsetAsync recorded as a memset NODE was not ordered before the kernel after it, so the detect stage now clears its counters with
    a kernel (sp_post.hip: launch_zero_words, used for every buffer the library clears); SuperGlue / GlueStick are NOT verified under replay.
There is no fallback: a forward that cannot be captured raises.
"""
import torch

from . import _native


def _flatten(d, prefix=""):
    for k in sorted(d):
        v = d[k]
        if isinstance(v, dict):
            yield from _flatten(v, prefix + k + "/")
        elif torch.is_tensor(v):
            yield prefix + k, v


def _rebuild(d, tensors, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out[k] = _rebuild(v, tensors, prefix + k + "/")
        elif torch.is_tensor(v):
            out[k] = tensors[prefix + k]
        else:
            out[k] = v
    return out


class GraphedModule:
    WARMUP = 3  # eager calls before capture: workspaces of the native library are sized on first use (hipMalloc cannot be captured)

    def __init__(self, module):
        self.module = module.eval()
        self._graphs = {}

    def reset(self):
        self._graphs.clear()

    def _module_state(self):
        """What the recorded launches of the native modules below depend on besides the library's own generation counter: the weight
        set each holds (`load_state_dict` bumps `_weights_version`; the upload itself is lazy) and its precision conf."""
        from .base_model import NativeModel

        return tuple((name, m._weights_version, str(m.conf.get("precision", "f16x3")))
                     for name, m in self.module.named_modules() if isinstance(m, NativeModel))

    @staticmethod
    def signature(data):
        return tuple((k, tuple(v.shape), v.dtype, v.device.index) for k, v in _flatten(data))

    def _capture(self, data):
        static = {k: v.clone() for k, v in _flatten(data)}
        static_data = _rebuild(data, static)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.WARMUP):
                self.module(static_data)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            out = self.module(static_data)
        return graph, static, out, (int(_native.lib().mg_workspace_generation()), self._module_state())

    def __call__(self, data):
        sig = self.signature(data)
        entry = self._graphs.get(sig)
        if entry is not None and entry[3] != (int(_native.lib().mg_workspace_generation()), self._module_state()):
            entry = None  # workspace / weights / kernel selection moved since the capture: its recorded launches are stale
        if entry is None:
            if not any(v.is_cuda for _, v in _flatten(data)):
                raise RuntimeError("GraphedModule needs device-resident inputs (stream capture records device work only)")
            self._graphs.pop(sig, None)
            entry = self._graphs[sig] = self._capture(data)
        graph, static, out, _ = entry
        for k, v in _flatten(data):
            static[k].copy_(v, non_blocking=True)
        graph.replay()
        return out
