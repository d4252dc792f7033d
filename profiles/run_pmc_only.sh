#!/bin/bash
# Counter passes only (no bench lines), on the GPU box:  profiles/run_pmc_only.sh <tag>
#   FETCH_SIZE / WRITE_SIZE -> gpurun_out/<tag>_f16x3_hbm_traffic.json;  MFMA busy -> <tag>_f16x3_mfma_util.json;
#   SQ instruction mix / LDS conflicts + calibration -> <tag>_f16x3_sq_counters.json, <tag>_mfma_calibration.json;  micro -> <tag>_micro.txt
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_${tag}_$c" -o pmc -- \
        python3 "$root/bench.py" --no-cpu-baseline --no-f32 --steps 2 --warmup 1 > "$out/${tag}_pmc_$c.log" 2>&1
done
cd "$root"
f=$(find "$out/pmc_${tag}_FETCH_SIZE" -name "*counter_collection.csv" | head -1)
w=$(find "$out/pmc_${tag}_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
python3 profiles/summarize_pmc.py "$f" "$w" "$out/${tag}_f16x3_hbm_traffic.json" > /dev/null
rm -rf "$out"/pmc_${tag}_FETCH_SIZE "$out"/pmc_${tag}_WRITE_SIZE
profiles/run_mfma_pmc.sh "$tag" > "$out/${tag}_mfma_summary.txt" 2>&1 || true
profiles/run_sq_pmc.sh "$tag" > "$out/${tag}_sq_summary.txt" 2>&1 || true
(tests/micro/mfma_peak; tests/micro/bw_pattern) > "$out/${tag}_micro.txt" 2>&1 || true
cat "$out/${tag}_micro.txt"
tail -12 "$out/${tag}_sq_summary.txt"
