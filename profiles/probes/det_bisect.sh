#!/bin/bash
# which switch makes two identically seeded Trainers diverge?  bash tools/probes/det_bisect.sh [repeats] [configs...]
N=${1:-3}; shift
run() { r=""; for i in $(seq 1 $N); do r="$r $(env $1 python tools/determinism_check.py ${DET_ARGS:-3} 2>&1 | grep -E 'bit-identical' | cut -d: -f2)"; done; echo "== $1:$r"; }
if [ $# -eq 0 ]; then set -- X=1 STYLEX_RES_GEMM=0 STYLEX_FROZEN_FUSE=0 STYLEX_UPLOAD_STREAM=0 STYLEX_CONV_PIPE=0 STYLEX_LPIPS_FUSE=0 STYLEX_EQL=0 STYLEX_MODCOEFF=0 STYLEX_PAD_RGB=0 STYLEX_STREAMS=0; fi
for v in "$@"; do run "$v"; done
