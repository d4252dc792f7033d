#!/usr/bin/env python3
"""Fold rocprofv3 --pmc SQ_* passes (one or more counter_collection.csv files) into per-kernel sums and derived ratios.

    python profiles/summarize_sq.py <out.json> <counter_collection.csv> [more.csv ...]

Units per MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_INSTS_* count wave-instructions; SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles.
Derived: share of a wave's lifetime spent parked (WAIT_ANY: s_waitcnt / barrier), stalled at issue (WAIT_INST_ANY) or
issuing (ACTIVE_INST_ANY); VALU share of issue; LDS bank-conflict rate; VALU instructions per MFMA-busy cycle.
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
    "ffn_f16x3_kernel": "lg.ffn",
    "heads_f16x3_kernel": "sp.heads",
    "mfma_loop": "micro.mfma_peak",
}


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for path in sys.argv[2:]:
        for r in csv.DictReader(open(path)):
            name = next((v for s, v in SHORT.items() if s in r["Kernel_Name"]), None)
            if name is None:
                continue
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[name][r["Counter_Name"]] += 1
    out = {}
    for name, a in acc.items():
        n = max(launches[name].values())
        e = {"launches": n, "per_launch": {k: round(v / launches[name][k], 1) for k, v in sorted(a.items())}}
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            for k, label in (("SQ_WAIT_ANY", "parked_share"), ("SQ_WAIT_INST_ANY", "issue_stall_share"), ("SQ_ACTIVE_INST_ANY", "issuing_share"),
                             ("SQ_ACTIVE_INST_VALU", "valu_issue_share"), ("SQ_WAIT_INST_LDS", "lds_issue_stall_share"),
                             ("SQ_INST_CYCLES_VMEM", "vmem_issue_share")):
                if k in a:
                    e[label] = round(a[k] / wc, 4)
        if a.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in a:
            e["lds_bank_conflict_rate"] = round(a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"], 4)
        if a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and "SQ_INSTS_VALU" in a:
            e["valu_insts_per_mfma"] = round(a["SQ_INSTS_VALU"] / (a["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0), 3)
        out[name] = e
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
