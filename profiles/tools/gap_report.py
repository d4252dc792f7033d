#!/usr/bin/env python3
"""Per-MFMA-gap instruction report of a kernel's inner loop(s) from hipcc's device assembly (-S / -save-temps .s).

    python profiles/tools/gap_report.py file.s [kernel-substring] [--all]

For every basic-block stretch that contains MFMAs it lists, per gap (the instructions between one MFMA and the next), the
instruction count by class and an issue-cost estimate from the measured price list of MI355X_MICROARCH.md (plain VALU 4 cycles,
transcendental 8, v_fma_mix 7, LDS read 4, LDS-DMA piece 60, SALU 4, s_nop n+1 ... an MFMA hides 24).  Used to place fillers in the
one-wave-per-SIMD kernels (attention_pw_f16.hip), where the order in the source is the order in the binary."""
import re
import sys


def cls(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")):
        return "trans"
    if op.startswith("v_fma_mix"):
        return "mix"
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_load", "global_load")) and "lds" in ins:
        return "dma"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_nop":
        return "nop"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


COST = {"trans": 8, "mix": 7, "valu": 4, "acc": 4, "lds": 4, "dma": 60, "vmem": 8, "salu": 4, "wait": 0, "barrier": 0, "other": 0}


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    lines = open(path).read().splitlines()
    in_kernel = False
    gaps = []
    cur = None
    label = ""
    for ln in lines:
        t = ln.strip()
        m = re.match(r"^([A-Za-z_][\w$.]*):", t)
        if m:
            in_kernel = want in m.group(1)
            continue
        if not in_kernel or not t or t.startswith((";", ".", "//")) and not t.startswith(".LBB"):
            continue
        if t.startswith(".LBB"):
            label = t.split(":")[0]
            continue
        ins = t.split(";")[0].strip()
        if not ins:
            continue
        c = cls(ins)
        if c == "mfma":
            if cur is not None:
                gaps.append(cur)
            cur = {"label": label, "n": {}, "cost": 0, "ins": []}
            continue
        if cur is None:
            continue
        cur["n"][c] = cur["n"].get(c, 0) + 1
        cur["ins"].append(ins.split()[0])
        if c == "nop":
            cur["cost"] += 4 * (int(ins.split()[1]) + 1)
        else:
            cur["cost"] += COST[c]
        if c == "barrier" or ins.startswith("s_endpgm"):
            gaps.append(cur)
            cur = None
    tot = 0
    over = 0
    for i, g in enumerate(gaps):
        tot += g["cost"]
        over += max(0, g["cost"] - 24)
        if "--all" in sys.argv or g["cost"] > 24:
            print(f"{i:4d} {g['label']:10s} cost {g['cost']:4d}  " + " ".join(f"{k}:{v}" for k, v in sorted(g["n"].items())))
    print(f"{len(gaps)} gaps, filler issue cost {tot} cycles, {over} cycles beyond the 24-cycle shadows")


if __name__ == "__main__":
    main()
