#!/bin/bash
# Instruction-mix / stall counters of bench.py on the GPU box, two SQ passes (8 slots each), plus the calibration of the
# MFMA-busy normalisation on the register-only MFMA loop (tests/micro/mfma_peak must read ~100 %):
#   profiles/run_sq_pmc.sh <tag>  -> gpurun_out/<tag>_f16x3_sq_counters.json, gpurun_out/<tag>_mfma_calibration.json
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out"
mkdir -p "$out"
[ -x "$root/tests/micro/mfma_peak" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o "$root/tests/micro/mfma_peak" "$root/tests/micro/mfma_peak.hip"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
    i=$((i + 1))
    timeout -k 10 500 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pmc_${tag}_sq$i" -o pmc -- \
        python3 "$root/bench.py" --no-cpu-baseline --no-f32 --steps 2 --warmup 1 > "$out/${tag}_pmc_sq$i.log" 2>&1
done
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_${tag}_calib" -o pmc -- \
    "$root/tests/micro/mfma_peak" > "$out/${tag}_pmc_calib.log" 2>&1
cd "$root"
python3 profiles/summarize_sq.py "$out/${tag}_f16x3_sq_counters.json" $(find "$out/pmc_${tag}_sq1" "$out/pmc_${tag}_sq2" -name "*counter_collection.csv") > /dev/null
python3 profiles/summarize_mfma.py "$(find "$out/pmc_${tag}_calib" -name "*counter_collection.csv" | head -1)" "$out/${tag}_mfma_calibration.json"
rm -rf "$out"/pmc_${tag}_sq* "$out"/pmc_${tag}_calib
python3 -c "
import json
d = json.load(open('$out/${tag}_f16x3_sq_counters.json'))
for k, v in d.items():
    print(k, {kk: vv for kk, vv in v.items() if kk != 'per_launch'})
"
