#!/bin/bash
# A/B of attention kernel variants through bench.py's per-kernel breakdown (one process per variant: debug requests are sticky).
#   profiles/tools/attn_ab.sh OUTPREFIX variant [variant ...]      variant = "" (default kernel) | lg.attn8 | lg.attn.exp<N>
out=$1; shift
for v in "$@"; do
  name=${v:-default}
  MG_BENCH_DEBUG_REQUEST=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32 > ${out}_${name}.json 2> ${out}_${name}.err || exit 1
  python - "$name" ${out}_${name}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); k = d["kernels"]
print(f"{sys.argv[1]:16s} step {d['ms_per_step']:7.3f} ms  attn self {k['lg.attn.self']['ms_per_step']:.3f} cross {k['lg.attn.cross']['ms_per_step']:.3f}  matches {d['config']['matches_per_step_rank0']}", flush=True)
PY
done
