#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass into per-kernel MFMA utilisation.

    python profiles/summarize_mfma.py <counter_collection.csv> <out.json> [n_cu=256]

Per MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's matrix pipe is busy, summed over
all SIMDs (32 per v_mfma_f32_32x32x16_f16); GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs, so the dispatch
lasted GRBM_GUI_ACTIVE / 8 shader cycles.  mfma_util = busy / (cycles x n_cu x 4 SIMDs): the fraction of the chip's
MFMA issue slots that were occupied AT THE CLOCK THE DISPATCH ACTUALLY RAN AT (DVFS lowers it under MFMA load), i.e. the
utilisation against the sustained rate; `clock_ghz` (from the dispatch timestamps, when the pass recorded them) times that
fraction over 2.4 GHz is the utilisation against the 2.5 PFLOP/s datasheet peak.
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
    "mfma_loop": "micro.mfma_peak (calibration: a register-only MFMA loop must read ~1.0)",
}


def main():
    n_cu = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(sys.argv[1])):
        name = next((v for s, v in SHORT.items() if s in r["Kernel_Name"]), None)
        if name is None:
            continue
        a = acc[name]
        a[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a["launches"] += 1
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {}
    for name, a in acc.items():
        cycles = a["GRBM_GUI_ACTIVE"] / 8.0
        if cycles <= 0:
            continue
        util = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * n_cu * 4)
        e = {"launches": int(a["launches"]), "mfma_util_at_actual_clock": round(util, 4)}
        if a["ns"] > 0:
            ghz = cycles / a["ns"]
            e["clock_ghz"] = round(ghz, 3)
            e["mfma_util_vs_2p4ghz_peak"] = round(util * ghz / 2.4, 4)
        out[name] = e
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
