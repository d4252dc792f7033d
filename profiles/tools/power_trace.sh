#!/bin/bash
# Board power and clocks while the timed region of bench.py runs (is the step power-limited?):
#   profiles/tools/power_trace.sh <tag> [bench args]   -> gpurun_out/<tag>_power.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out="$root/gpurun_out"
mkdir -p "$out"
log="$out/${tag}_power_raw.txt"
: > "$log"
( while true; do date +%s.%N >> "$log"; /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|junction\|cap" >> "$log"; sleep 0.3; done ) &
sampler=$!
/opt/rocm/bin/rocm-smi --showmaxpower --showperflevel 2>&1 | grep -v "^=\|^$" > "$out/${tag}_power_caps.txt"
python "$root/bench.py" --no-cpu-baseline --no-f32 --steps 300 --warmup 20 "$@" > "$out/${tag}_power_bench.json" 2> "$out/${tag}_power_bench.err"
rc=$?
kill $sampler 2>/dev/null; wait $sampler 2>/dev/null
python3 - "$log" "$out/${tag}_power_caps.txt" "$out/${tag}_power_bench.json" <<'PY' | tee "$out/${tag}_power.txt"
import json, re, sys
raw = open(sys.argv[1]).read().splitlines()
pw, sclk = [], []
for l in raw:
    m = re.search(r"Power \(W\):\s*([\d.]+)", l)
    if m: pw.append(float(m.group(1)))
    m = re.search(r"sclk clock level.*\((\d+)Mhz\)", l)
    if m: sclk.append(int(m.group(1)))
print(open(sys.argv[2]).read().strip())
if pw:
    s = sorted(pw)
    print(f"{len(pw)} power samples: min {s[0]:.0f} W, median {s[len(s)//2]:.0f} W, p90 {s[int(len(s)*0.9)]:.0f} W, max {s[-1]:.0f} W")
if sclk:
    s = sorted(sclk)
    print(f"{len(sclk)} sclk samples: min {s[0]} MHz, median {s[len(s)//2]} MHz, max {s[-1]} MHz")
try:
    d = json.load(open(sys.argv[3]))
    print(f"bench: {d['value']} pairs/s, {d['ms_per_step']} ms/step over {d['steps']} steps")
except Exception as e:
    print("bench line unreadable:", e)
print("---- last raw sample ----")
print("\n".join(raw[-8:]))
PY
exit $rc
