#!/bin/bash
# Counter passes over the stand-alone attention entry (profiles/tools/attn_variants.py), per kernel instantiation:
#   profiles/tools/attn_pmc.sh <tag> variant [variant ...]   -> gpurun_out/<tag>_attn_pmc.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES"; do
    i=$((i + 1))
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pmc_${tag}_a$i" -o pmc -- \
        python3 "$root/profiles/tools/attn_variants.py" --reps 4 "$@" > "$out/${tag}_attn_pmc_$i.log" 2>&1 || { tail -5 "$out/${tag}_attn_pmc_$i.log"; exit 1; }
done
cd "$root"
python3 - "$out" "$tag" <<'PY' | tee "$out/${tag}_attn_pmc.txt"
import collections, csv, glob, sys
out, tag = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{out}/pmc_{tag}_a*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attention" not in k:
            continue
        import re
        m = re.search(r"(attention\w*kernel(<\d+>)?)", k)
        k = m.group(1) if m else k[:60]
        a = acc[k]
        a[r["Counter_Name"]] += float(r["Counter_Value"])
        a["n_" + r["Counter_Name"]] += 1
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA") and r.get("Start_Timestamp"):
            a["ns_" + r["Counter_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, a in sorted(acc.items()):
    n = a["n_GRBM_GUI_ACTIVE"] or 1
    cyc = a["GRBM_GUI_ACTIVE"] / 8 / n
    us = a["ns_GRBM_GUI_ACTIVE"] / n / 1e3
    print(f"{k}: {us:7.1f} us/launch, {cyc / 1e3:7.1f} k shader cycles -> {cyc / (us * 1e3):.2f} GHz; MFMA busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / n / (cyc * 1024):.3f}")
    per = lambda c: a[c] / (a['n_' + c] or 1)
    wc = per("SQ_WAVE_CYCLES") or 1
    print("    per launch: " + ", ".join(f"{c} {per(c):.3g}" for c in sorted(a) if not c.startswith(("n_", "ns_"))))
    print(f"    of wave cycles: wait_any {per('SQ_WAIT_ANY') / wc:.3f}, wait_inst_any {per('SQ_WAIT_INST_ANY') / wc:.3f}, active_inst_any {per('SQ_ACTIVE_INST_ANY') / wc:.3f}, "
          f"active_valu {per('SQ_ACTIVE_INST_VALU') / wc:.3f}, wait_inst_lds {per('SQ_WAIT_INST_LDS') / wc:.3f}; LDS conflict rate {per('SQ_LDS_BANK_CONFLICT') / (per('SQ_LDS_IDX_ACTIVE') or 1):.3f}; "
          f"VALU/MFMA {per('SQ_INSTS_VALU') / (per('SQ_INSTS_MFMA') or 1):.2f}")
PY
rm -rf "$out"/pmc_${tag}_a*
