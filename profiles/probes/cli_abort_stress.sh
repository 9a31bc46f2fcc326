#!/bin/bash
# usage (GPU box): bash tools/probes/cli_abort_stress.sh [runs]  — the 64 px CLI test in a loop with cudnn.deterministic on
# (STYLEX_DETERMINISTIC=1) and the toRGB side stream on, faulthandler armed; any non-zero exit keeps its full output.
N=${1:-60}
O=$GRAFT_REPO_ROOT/gpurun_out/cli_stress
mkdir -p $O
cd $GRAFT_REPO_ROOT
fail=0
for i in $(seq 1 $N); do
  STYLEX_DETERMINISTIC=1 STYLEX_G_SIDE=${G_SIDE:-1} timeout 300 python -X faulthandler -m pytest tests/test_hip_cli_gpu.py -q -x > $O/run.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fail=$((fail+1)); cp $O/run.log $O/fail_${i}_rc${rc}.log; fi
  echo "$i rc=$rc" >> $O/summary.txt
done
echo "runs=$N failures=$fail" >> $O/summary.txt
tail -3 $O/summary.txt
