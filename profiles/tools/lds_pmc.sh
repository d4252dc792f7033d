#!/bin/bash
# LDS counters of the kernels whose name matches a regex (one pass):
#   profiles/tools/lds_pmc.sh <tag> <kernel-regex> <program> [args...]   -> gpurun_out/<tag>_lds_pmc.txt
tag=$1; rx=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$out/pmc_${tag}_lds" -o pmc -- "$@" > "$out/${tag}_lds_pmc.log" 2>&1 || { tail -5 "$out/${tag}_lds_pmc.log"; exit 1; }
cd "$root"
python3 - "$out" "$tag" "$rx" <<'PY' | tee "$out/${tag}_lds_pmc.txt"
import collections, csv, glob, re, sys
out, tag, rx = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{out}/pmc_{tag}_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(rx, r["Kernel_Name"])
        if not m:
            continue
        a = acc[m.group(0)[:70]]
        a[r["Counter_Name"]] += float(r["Counter_Value"])
        a["n_" + r["Counter_Name"]] += 1
for k, a in sorted(acc.items()):
    per = lambda c: a[c] / (a["n_" + c] or 1)
    n = per("SQ_INSTS_LDS") or 1
    print(f"{k}: {int(a['n_SQ_INSTS_LDS'])} launches; per launch: LDS instructions {n:.4g}, LDS array cycles {per('SQ_LDS_IDX_ACTIVE'):.4g} ({per('SQ_LDS_IDX_ACTIVE') / n:.2f} per instruction), "
          f"bank-conflict cycles {per('SQ_LDS_BANK_CONFLICT'):.4g}, address-conflict {per('SQ_LDS_ADDR_CONFLICT'):.4g}, UNALIGNED stall {per('SQ_LDS_UNALIGNED_STALL'):.4g} "
          f"({per('SQ_LDS_UNALIGNED_STALL') / n:.1f} per instruction), wait_inst_lds {per('SQ_WAIT_INST_LDS'):.4g}, active_inst_lds {per('SQ_ACTIVE_INST_LDS'):.4g}, wave cycles {per('SQ_WAVE_CYCLES'):.4g}")
PY
rm -rf "$out"/pmc_${tag}_lds
