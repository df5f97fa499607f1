#!/bin/bash
# tools/cross_process_run.sh OUTDIR ITERATIONS neighbour-mode[@ENV=VAL,...] ...   (env applies to the neighbour only)
cd "$(dirname "$0")/.."
OUT=$1; IT=$2; shift 2; mkdir -p $OUT
for spec in "$@"; do
mode=${spec%%@*}; envs=""; [ "$spec" != "$mode" ] && envs=$(echo "${spec#*@}" | tr ',' ' ')
rm -f $OUT/stop $OUT/stop.ready
env $envs timeout -k 5 300 python3 tools/experiments/x3_neighbour.py $OUT/stop 280 $mode > $OUT/neigh.log 2>&1 &
NP=$!
for i in $(seq 1 90); do [ -e $OUT/stop.ready ] && break; sleep 1; done
echo "== neighbour: $spec (ready: $([ -e $OUT/stop.ready ] && echo yes || echo NO)); victim env: ${VICTIM_ENV:-none}"
env ${VICTIM_ENV} timeout -k 5 260 python3 tools/experiments/cross_process_probe.py $IT 2>&1 | grep -v amdgpu.ids | tail -8
touch $OUT/stop; wait $NP; tail -1 $OUT/neigh.log
done
