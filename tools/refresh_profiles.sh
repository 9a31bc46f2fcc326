#!/bin/bash
# usage (GPU box): bash tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>_* (copy the ones to keep into profiles/)
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${TAG}_bench_bf16_default.json 2>$O/${TAG}_bench_default.err
python bench.py --precision fp32 --no-cpu-baseline --fp32-steps 0 > $O/${TAG}_bench_fp32.json 2>/dev/null
python bench.py --gae 1 --no-cpu-baseline --fp32-steps 0 > $O/${TAG}_bench_bf16_benchA_gae1.json 2>/dev/null
python tools/bench_conv.py --batch 64 > $O/${TAG}_conv_layers_bf16_b64.txt 2>&1
python tools/bench_elementwise.py > $O/${TAG}_elementwise_bf16_b64.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o r01 -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --fp32-steps 0 --bench-a-steps 0 > /tmp/b_$TAG.log 2>&1
python3 $R/tools/prof_summary.py $(find /tmp/prof_$TAG -name "*kernel_stats.csv") 12 > $O/${TAG}_kernel_stats_bench_bf16.csv
grep -a ms_per_step /tmp/b_$TAG.log > $O/${TAG}_kernel_stats_bench_line.txt
python3 $R/tools/trace_pack.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv") $O/trace_$TAG.csv.gz
cd $R
python tools/collect_traffic.py --out $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.log 2>&1
