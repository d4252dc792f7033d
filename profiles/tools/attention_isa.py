#!/usr/bin/env python3
"""Branch-aware instruction count of ONE full key tile of attention_f16x3_kernel's main loop, from the device assembly.

    python3 profiles/tools/attention_isa.py [git-rev-of-attention_f16.hip ...]   ->  stdout (kept as profiles/r03_attention_isa.txt)

The loop body contains wave-uniform branches around work that a full key tile skips (tail masking of a partial tile, the rare
running-maximum rescale); a plain count of the loop's text (what DESIGN.md quoted in round 2: "~330 VALU per tile, hipcc turns the
masking into unconditional compare + select pairs") counts those blocks as executed.  This script follows the forward branches that
skip such blocks and prices the rest with the issue costs of MI355X_MICROARCH.md ('vector-instruction ISSUE cost': transcendental 8
cycles, other VALU 4, an MFMA holds the vector issue port for 8 of its 32 cycles)."""
import re
import subprocess
import sys
import os
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = "glue-factory_amd/csrc/attention_f16.hip"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
NO_MIX = ["-Xclang", "-target-feature", "-Xclang", "-fma-mix-insts"]


def asm_of(rev):
    if rev == "worktree":
        text = open(os.path.join(ROOT, SRC)).read()
    else:
        text = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{SRC}"], capture_output=True, text=True, check=True).stdout
    tmp = f"/tmp/attn_isa_{os.getpid()}.hip"
    open(tmp, "w").write(text.replace('#include "common.h"', f'#include "{ROOT}/glue-factory_amd/csrc/common.h"')
                         .replace('#include "kernels.h"', f'#include "{ROOT}/glue-factory_amd/csrc/kernels.h"'))
    mix = [] if "v_fma_mixlo_f16" in text else NO_MIX
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *mix, "-S", "--cuda-device-only", tmp, "-o", "-"], capture_output=True, text=True).stdout
    os.remove(tmp)
    return out


def hot_path(asm):
    m = re.search(r"^(_ZN[^\n]*attention_f16x3_kernel[^\n]*):\s*;.*?^\.Lfunc_end\d+:", asm, re.S | re.M)
    lines = [l.strip().split(";")[0].strip() for l in m.group(0).splitlines()]
    lines = [l for l in lines if l and (not l.startswith(".") or re.match(r"\.LBB\d+_\d+:", l))]
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
    for i in range(mf[-1], len(lines)):  # the loop's back edge
        mm = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if mm and labels.get(mm.group(1), 10 ** 9) < mf[0]:
            head, back = labels[mm.group(1)], i
            break
    seg = lines[head:back + 1]
    c, i = Counter(), 0
    while i < len(seg):
        l = seg[i]
        mm = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels:
            tgt = labels[mm.group(1)] - head
            # a forward branch over > 20 instructions that skips no LDS store: tail masking / rescale (skipped on a full tile);
            # the branch over the next tile's LDS stores is NOT taken (every tile but the last stores the next one)
            if tgt > i + 20 and "ds_write" not in " ".join(seg[i:tgt]):
                i = tgt
                continue
        if not l.endswith(":"):
            c[l.split()[0]] += 1
        i += 1
    return c


def main():
    revs = sys.argv[1:] or ["worktree"]
    for rev in revs:
        c = hot_path(asm_of(rev))
        mfma = sum(v for k, v in c.items() if k.startswith("v_mfma"))
        trans = sum(v for k, v in c.items() if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", k))
        valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        issue = trans * 8 + (valu - trans) * 4 + mfma * 8
        print(f"== attention_f16.hip @ {rev}: one full 32-key tile, one wave (64 queries)")
        print(f"   MFMA {mfma}   VALU {valu} (of them transcendental: {trans} v_exp_f32)   LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}   "
              f"VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_')))}   SALU/other {sum(c.values()) - mfma - valu}")
        print(f"   vector issue port: {trans} x 8 + {valu - trans} x 4 + {mfma} x 8 = {issue} cycles per wave; two waves per SIMD: {2 * issue}")
        print(f"   matrix pipe: {mfma} x 32 = {mfma * 32} cycles per wave; two waves per SIMD: {2 * mfma * 32}   ->  bound by the "
              f"{'vector issue port' if 2 * issue > 2 * mfma * 32 else 'matrix pipe'}")
        top = sorted(((v, k) for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma")), reverse=True)[:12]
        print("   VALU mix:", ", ".join(f"{k} {v}" for v, k in top))


if __name__ == "__main__":
    main()
