#!/bin/bash
# Full evidence pass on the GPU box (run through gpurun from the repo root):  profiles/run_evidence.sh <tag>
#   1. bench.py default (f16x3, with the CPU baseline) and --precision f32        -> gpurun_out/<tag>_*_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command                        -> gpurun_out/<tag>_f16x3_kernel_stats.csv
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) -> gpurun_out/<tag>_f16x3_hbm_traffic.json
#   4. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (profiles/run_mfma_pmc.sh)     -> gpurun_out/<tag>_f16x3_mfma_util.json
# Copy what should be judged into profiles/.
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out"
mkdir -p "$out"
cd "$root"
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 > "$out/${tag}_f16x3_bench.json" 2> "$out/${tag}_f16x3_bench.err"
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --precision f32 --no-cpu-baseline > "$out/${tag}_f32_bench.json" 2> "$out/${tag}_f32_bench.err"
profiles/run_kernel_trace.sh "${tag}_f16x3" --steps 4 --warmup 2 > "$out/${tag}_trace_summary.txt"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_${tag}_$c" -o pmc -- \
        python3 "$root/bench.py" --no-cpu-baseline --no-f32 --steps 2 --warmup 1 > "$out/${tag}_pmc_$c.log" 2>&1
done
cd "$root"
f=$(find "$out/pmc_${tag}_FETCH_SIZE" -name "*counter_collection.csv" | head -1)
w=$(find "$out/pmc_${tag}_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
python3 profiles/summarize_pmc.py "$f" "$w" "$out/${tag}_f16x3_hbm_traffic.json" > /dev/null
rm -rf "$out"/pmc_${tag}_* "$out"/prof_${tag}_*
profiles/run_mfma_pmc.sh "$tag" > "$out/${tag}_mfma_summary.txt" 2>&1 || true
for m in mfma_peak bw_pattern; do  # micro-benchmarks are built on demand (the binaries are not tracked)
    [ -x tests/micro/$m ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tests/micro/$m tests/micro/$m.hip
done
(tests/micro/mfma_peak; tests/micro/bw_pattern) > "$out/${tag}_micro.txt" 2>&1 || true
cat "$out/${tag}_micro.txt"
cat "$out/${tag}_trace_summary.txt"
python3 -c "
import json
for p in ('f16x3', 'f32'):
    d = json.load(open('$out/${tag}_' + p + '_bench.json'))
    print(p, d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['roofline'])
print(json.dumps(json.load(open('$out/${tag}_f16x3_hbm_traffic.json')), indent=0)[:1500])
"
