#!/bin/bash
# usage (GPU box): bash tools/ddp_1rank_ab.sh <out>   -> bench.py without a process group / under torchrun with ONE rank (RCCL path:
# flat buckets, in-backward launches, broadcast, NaN flag), A/B/B/A; the difference is the cost of the data-parallel plumbing itself
OUT=$1
cd $GRAFT_REPO_ROOT
ARGS="--steps 16 --warmup 6 --no-cpu-baseline --roofline-steps 0 --bench-a-steps 0 --fp32-steps 0"
val() { python -c "import sys,json;l=sys.stdin.read();j=json.loads(l[l.index('{'):]);print('%s %.2f images/s %.2f ms' % (sys.argv[1], j['value'], j['ms_per_step']))" "$1"; }
plain() { python bench.py --gpus 1 $ARGS 2>/dev/null | tail -1 | val "no process group "; }
ddp() { STYLEX_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $ARGS 2>/dev/null | tail -1 | val "torchrun, 1 rank   "; }
{ plain; ddp; ddp; plain; } > $OUT 2>&1
