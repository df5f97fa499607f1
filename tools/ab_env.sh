#!/bin/bash
# A/B of one environment switch on the 1024-px step: alternating runs of bench.py on ONE box.
#   tools/ab_env.sh VAR A_VALUE B_VALUE [runs] [extra bench args...]
# prints "VAR=value steps/s" per run, then mean +- sd per arm (and the SMI clock readings if rocm-smi works)
var=$1; a=$2; b=$3; runs=${4:-5}; shift 4
for i in $(seq 1 $runs); do
  for v in $a $b; do
    line=$(env $var=$v python3 bench.py --no-cpu-baseline --no-e2e --no-pyramid --no-families --steps 60 --warmup 10 "$@" 2>/dev/null | tail -1)
    echo "$var=$v $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done | tee /tmp/ab_$$.txt
python3 - /tmp/ab_$$.txt <<'PY'
import sys, statistics as st
arms = {}
for l in open(sys.argv[1]):
    k, v, ms = l.split()
    arms.setdefault(k, []).append(float(ms))
for k, v in arms.items():
    print(f"{k}: ms/step mean {st.mean(v):.4f} sd {st.stdev(v) if len(v) > 1 else 0:.4f} min {min(v):.4f} n={len(v)}")
PY
