#!/bin/bash
# MFMA-utilisation counter pass of bench.py on the GPU box:  profiles/run_mfma_pmc.sh <tag>  -> gpurun_out/<tag>_f16x3_mfma_util.json
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_${tag}_mfma" -o pmc -- \
    python3 "$root/bench.py" --no-cpu-baseline --no-f32 --steps 2 --warmup 1 > "$out/${tag}_pmc_mfma.log" 2>&1
cd "$root"
f=$(find "$out/pmc_${tag}_mfma" -name "*counter_collection.csv" | head -1)
head -2 "$f" | cut -c1-600
python3 profiles/summarize_mfma.py "$f" "$out/${tag}_f16x3_mfma_util.json"
rm -rf "$out/pmc_${tag}_mfma"
