#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

    python profiles/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Corrections per MI355X_MICROARCH.md (HBM section): counters are in KiB... rocprofv3 reports FETCH_SIZE / WRITE_SIZE
in kilobytes; on gfx950 FETCH_SIZE counts exactly half of the bytes of wide (16 B/lane) coalesced reads, which is how
every kernel here loads, so reads = 2 x FETCH_SIZE; WRITE_SIZE is exact for 16 B/lane stores.
"""
import collections
import csv
import json
import sys

SHORT = {
    "conv1ab_f16x3_kernel": "sp.conv1ab",
    "conv3x3_f16x3_kernel": "sp.conv3x3",
    "conv3x3_dma_kernel": "sp.conv3x3 (32x32x16 DMA)",
    "conv3x3_pair_kernel": "sp.conv3x3",
    "conv1ab_dma_kernel": "sp.conv1ab",
    "conv1ab_mfma_kernel": "sp.conv1ab",
    "gemm_kernel<true, true>": "lg.gemm.sim (precise)",
    "gemm_kernel<false, true>": "lg.gemm.final_proj (precise)",
    "gemm_f16x3_kernel<false": "gemm (tiled weights)",
    "gemm_f16x3_kernel<true": "lg.gemm.sim",
    "attention_f16x3_kernel": "lg.attn",
    "attention_f16x3_pw_kernel": "lg.attn",
    "gemm_ws_f16x3_kernel": "lg.gemm.qkv (weight-stationary)",
    "nms_kernel": "sp.nms",
    "ln_gelu_kernel": "lg.ln_gelu",
    "ffn_f16x3_kernel": "lg.ffn",
    "heads_f16x3_kernel": "sp.heads",
    "softmax_d2s_kernel": "sp.softmax_d2s",
    "select_kernel": "sp.select",
    "sample_kernel": "sp.sample",
}


def agg(path):
    a = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        a[r["Kernel_Name"]][0] += float(r["Counter_Value"])
        a[r["Kernel_Name"]][1] += 1
    return a


def main():
    f, w = agg(sys.argv[1]), agg(sys.argv[2])
    out = {}
    for k, (tot, n) in f.items():
        name = next((v for s, v in SHORT.items() if s in k), None)
        if name is None:
            continue
        wt, wn = w.get(k, [0.0, 1])
        e = out.setdefault(name, {"launches": 0, "read_bytes": 0.0, "write_bytes": 0.0})
        e["launches"] += n
        e["read_bytes"] += 2.0 * tot * 1024
        e["write_bytes"] += wt * 1024
    for e in out.values():
        e["hbm_bytes_per_launch"] = (e["read_bytes"] + e["write_bytes"]) / e["launches"]
        e["read_bytes_per_launch"] = e.pop("read_bytes") / e["launches"]
        e["write_bytes_per_launch"] = e.pop("write_bytes") / e["launches"]
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
