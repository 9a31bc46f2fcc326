#!/bin/bash
# usage (GPU box): bash tools/trace_bench.sh <tag> [bench args...]   -> gpurun_out/trace_<tag>.csv.gz + bench line
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --bench-a-steps 0 "$@" > /tmp/b_$TAG.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
grep -a "ms_per_step" /tmp/b_$TAG.log > $GRAFT_REPO_ROOT/gpurun_out/trace_${TAG}_bench.txt
python3 $GRAFT_REPO_ROOT/tools/trace_pack.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv") $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG.csv.gz $TRACE_GRID
