#!/bin/bash
# Kernel-trace + stats pass of one bench.py command on the GPU box (run through gpurun from the repo root):
#   profiles/run_kernel_trace.sh <tag> [bench.py args...]
# writes gpurun_out/<tag>_kernel_stats.csv; copy the summary into profiles/ to have it judged.
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_$tag" -o "$tag" -- \
    python3 "$root/bench.py" --no-cpu-baseline --no-f32 "$@" > "$root/gpurun_out/${tag}_prof.log" 2>&1
f=$(find "$root/gpurun_out/prof_$tag" -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$root/gpurun_out/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as fh:  # same columns, kernel names cut to 160 characters (template arguments run long)
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:160]
        w.writerow(r)
for r in rows[:16]:
    print(f'{r["Name"][:64]:64s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:9.1f} us {r["Percentage"]:>6s}%')
PY
