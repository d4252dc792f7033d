#!/usr/bin/env python3
"""Time attention kernel variants through the stand-alone entry mg_test_attention (one engine per variant; wrong results of the
ablations stay inside this script -- never run an ablation inside the pipeline: garbage attention output reaches index arithmetic
downstream).  The entry also runs three to_sh8 conversions and one from_sh8 (constant overhead, measured by the 'exp15'-style
all-off variant when given).

    python profiles/tools/attn_variants.py [--B 32] [--M 1024] [--N 1024] variant [variant ...]   ("default" = no request)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "glue-factory_amd"))
import torch  # noqa: E402

from gluefactory_amd import _native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--M", type=int, default=1024)
    ap.add_argument("--N", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    T = a.B * (a.M + a.N)
    q, k, v = (torch.randn(T, 256, generator=g).mul_(s).to(dev) for s in (0.35, 1.0, 1.0))
    ref = None
    for name in a.variants:
        e = _native.Engine(dev)
        e.set_precision("f16x3")
        if name != "default":
            e.debug_request(name)
        for cross in (False, True):
            for _ in range(3):
                out = e.test_attention(q, k, v, a.B, a.M, a.N, cross)
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(a.reps):
                out = e.test_attention(q, k, v, a.B, a.M, a.N, cross)
            t1.record()
            torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / a.reps
            note = ""
            if name in ("default", "lg.attn8"):
                if ref is None or ref[0] != cross:
                    ref = (cross, out.clone())
                else:
                    note = f" max |diff| vs first exact variant {float((out - ref[1]).abs().max()):.2e}"
            print(f"{name:16s} cross={int(cross)}  {ms * 1e3:8.1f} us per call (entry incl. 4 conversion launches){note}", flush=True)
        e.close()


if __name__ == "__main__":
    main()
