#!/bin/bash
# SQ / MFMA counters of the kernels whose name matches a regex, over any command (two passes):
#   profiles/tools/kernel_pmc.sh <tag> <kernel-regex> <program> [args...]   -> gpurun_out/<tag>_kernel_pmc.txt
# (<program> must be the executable itself -- python3 ... -- not a shell wrapper: the profiler's preload initialises the GPU)
tag=$1; rx=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES"; do
    i=$((i + 1))
    timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pmc_${tag}_k$i" -o pmc -- "$@" > "$out/${tag}_kernel_pmc_$i.log" 2>&1 || { tail -5 "$out/${tag}_kernel_pmc_$i.log"; exit 1; }
done
cd "$root"
python3 - "$out" "$tag" "$rx" <<'PY' | tee "$out/${tag}_kernel_pmc.txt"
import collections, csv, glob, re, sys
out, tag, rx = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{out}/pmc_{tag}_k*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(rx, k)
        if not m:
            continue
        k = m.group(0)[:70]
        a = acc[k]
        a[r["Counter_Name"]] += float(r["Counter_Value"])
        a["n_" + r["Counter_Name"]] += 1
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("Start_Timestamp"):
            a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, a in sorted(acc.items()):
    n = a["n_GRBM_GUI_ACTIVE"] or 1
    cyc = a["GRBM_GUI_ACTIVE"] / 8 / n
    us = a["ns"] / n / 1e3
    per = lambda c: a[c] / (a['n_' + c] or 1)
    wc = per("SQ_WAVE_CYCLES") or 1
    print(f"{k}: {int(n)} launches, {us:7.1f} us/launch, {cyc / 1e3:7.1f} k shader cycles -> {cyc / max(us * 1e3, 1):.2f} GHz; MFMA busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / n / (cyc * 1024):.3f}")
    print(f"    of wave cycles: wait_any {per('SQ_WAIT_ANY') / wc:.3f}, wait_inst_any {per('SQ_WAIT_INST_ANY') / wc:.3f}, active_inst_any {per('SQ_ACTIVE_INST_ANY') / wc:.3f}, "
          f"active_valu {per('SQ_ACTIVE_INST_VALU') / wc:.3f}, wait_inst_lds {per('SQ_WAIT_INST_LDS') / wc:.3f}; LDS conflict rate {per('SQ_LDS_BANK_CONFLICT') / (per('SQ_LDS_IDX_ACTIVE') or 1):.3f}; "
          f"VALU/MFMA {per('SQ_INSTS_VALU') / (per('SQ_INSTS_MFMA') or 1):.2f}; wave quad-cycles per launch {wc:.3g}")
PY
rm -rf "$out"/pmc_${tag}_k*
