#!/bin/bash
# Profiles of one round on the GPU box: tools/profile_round.sh rNN   (writes gpurun_out/prof_rNN/, summaries -> copy to profiles/)
#   1. rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-e2e --no-pyramid` (the timed 1024-px steps + the
#      per-family pass + the pairwise kernel), 2./3. separate --pmc FETCH_SIZE / WRITE_SIZE passes of the same command with
#      --no-families --no-graph --no-long-window --steps 6 --warmup 2 (counters serialise the kernels; their timing is not used)
r=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$r
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e --no-pyramid --no-live-pmc"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $B --steps 40 --warmup 5 > $out/bench_under_rocprof.json 2> $out/trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $B --no-families --no-graph --no-long-window --steps 6 --warmup 2 > /dev/null 2> $out/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $B --no-families --no-graph --no-long-window --steps 6 --warmup 2 > /dev/null 2> $out/write.err || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/summarize_rocprof.py trace $out/trace $out/$r && python3 tools/summarize_rocprof.py pmc $out/fetch $out/write $out/$r
find $out -name "*_kernel_trace.csv" -size +8M -delete; find $out -name "*counter_collection.csv" -size +8M -delete
ls -la $out
