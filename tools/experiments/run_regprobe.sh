#!/bin/bash
# register-integrity probe alone, then next to the bf16x3 GEMM of another process
cd "$(dirname "$0")/../.."
OUT=$1; mkdir -p $OUT
echo "== alone"; ./tools/experiments/regprobe_host tools/experiments/regprobe.co 100 60 | tail -12
for mode in mstats fwd; do
rm -f $OUT/stop $OUT/stop.ready
timeout -k 5 200 python3 tools/experiments/x3_neighbour.py $OUT/stop 150 $mode > $OUT/neigh.log 2>&1 &
NP=$!
for i in $(seq 1 90); do [ -e $OUT/stop.ready ] && break; sleep 1; done
echo "== next to neighbour $mode (ready: $([ -e $OUT/stop.ready ] && echo yes || echo NO))"
timeout -k 5 120 ./tools/experiments/regprobe_host tools/experiments/regprobe.co 300 60 | tail -60
touch $OUT/stop; wait $NP; tail -1 $OUT/neigh.log
done
