#!/bin/bash
# Secondary workloads at HEAD on the GPU box (run through gpurun from the repo root):  profiles/run_secondary.sh <tag>
#   configs[3]  SuperPoint + SuperGlue, 2048 keypoints   -> gpurun_out/<tag>_superglue_bench.json, <tag>_superglue_kernel_stats.csv
#   configs[4]  SuperPoint + GlueStick, synthetic segments -> gpurun_out/<tag>_gluestick_bench.json
#   configs[2]  per-GPU share (32 pairs, 1024x1024, 2048 keypoints) -> gpurun_out/<tag>_config2_share_bench.json
#   the packed-fp32 reproducer and the MFMA layout probe    -> gpurun_out/<tag>_pk_hazard.txt, <tag>_mfma16_layout.txt
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out"
mkdir -p "$out"
cd "$root"
timeout -k 10 300 python3 bench.py --matcher superglue --kpts 2048 --steps 10 --warmup 3 > "$out/${tag}_superglue_bench.json" 2> "$out/${tag}_superglue_bench.err"
profiles/run_kernel_trace.sh "${tag}_superglue" --matcher superglue --kpts 2048 --steps 4 --warmup 2 > "$out/${tag}_superglue_trace_summary.txt"
timeout -k 10 200 python3 bench.py --matcher gluestick --steps 10 --warmup 3 --no-cpu-baseline > "$out/${tag}_gluestick_bench.json" 2> "$out/${tag}_gluestick_bench.err"
timeout -k 10 300 python3 bench.py --size 1024x1024 --kpts 2048 --steps 10 --warmup 3 --no-cpu-baseline > "$out/${tag}_config2_share_bench.json" 2> "$out/${tag}_config2_share_bench.err"
for m in pk_hazard mfma16_layout; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tests/micro/$m tests/micro/$m.hip 2>/dev/null
    timeout -k 10 150 tests/micro/$m > "$out/${tag}_$m.txt" 2>&1 || true
done
python3 - <<PY
import json
for w in ("superglue", "gluestick", "config2_share"):
    d = json.loads(open("$out/${tag}_" + w + "_bench.json").read().strip().splitlines()[-1])
    print(w, d["value"], d["unit"], d["ms_per_step"], "ms/step", {k: v["ms_per_step"] for k, v in list(d["kernels"].items())[:8]})
PY
cat "$out/${tag}_pk_hazard.txt"
