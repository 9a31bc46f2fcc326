# usage (GPU box): bash tools/two_ranks_diag.sh <tag> [ENV=VAL ...] -> gpurun_out/two_diag_<tag>.json: tools/ddp_two_ranks_one_gpu.py with TWO_DIAG=1
# (every exchange mode repeated; local gradients entering / averaged buckets leaving each exchange compared run to run)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=$1; shift
PORT=$((29561 + RANDOM % 200))
(env "$@" RANK=1 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT TWO_DIAG=1 python tools/ddp_two_ranks_one_gpu.py 2 > /dev/null 2>gpurun_out/two_r1_$TAG.err &)
env "$@" RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT TWO_DIAG=1 timeout 900 python tools/ddp_two_ranks_one_gpu.py 2 > gpurun_out/two_diag_$TAG.json 2>gpurun_out/two_r0_$TAG.err; tail -2 gpurun_out/two_r0_$TAG.err
sleep 3
