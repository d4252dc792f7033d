#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$out/pmc_ic" -o pmc -- "$@" > "$out/r04_at_icache.log" 2>&1 || { tail -5 "$out/r04_at_icache.log"; exit 1; }
cd "$root"
python3 - "$out" <<'PY' | tee "$out/r04_at_icache_pmc.txt"
import collections, csv, glob, re, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{out}/pmc_ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"ffn_f16x3_kernel|gemm_ws_f16x3_kernel|attention_f16x3_pw_kernel|conv1ab_mfma_kernel|conv3x3_pair_kernel", r["Kernel_Name"])
        if not m: continue
        a = acc[m.group(0)]
        a[r["Counter_Name"]] += float(r["Counter_Value"]); a["n_" + r["Counter_Name"]] += 1
for k, a in sorted(acc.items()):
    per = lambda c: a[c] / (a["n_" + c] or 1)
    print(f"{k}: icache req {per('SQC_ICACHE_REQ'):.4g} hits {per('SQC_ICACHE_HITS'):.4g} misses {per('SQC_ICACHE_MISSES'):.4g} ifetch {per('SQ_IFETCH'):.4g}; wave cycles {per('SQ_WAVE_CYCLES'):.4g} wait_any {per('SQ_WAIT_ANY'):.4g} wait_inst_any {per('SQ_WAIT_INST_ANY'):.4g} active_inst_any {per('SQ_ACTIVE_INST_ANY'):.4g}")
PY
rm -rf "$out/pmc_ic"
