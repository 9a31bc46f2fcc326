#!/bin/bash
# kernel trace of the <= 8 px layers in isolation: per-shape duration of conv_gather_kernel / splitk_epilogue_kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gather_trace -- python3 $R/tools/bench_conv.py --batch 64 --iters 10 --only "G0.conv1,G1.conv1,D5.conv1,D6.conv1,D7.conv1,D5.down,D6.down" > $R/gpurun_out/gather_trace.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/gather_trace/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gather" in n or "splitk" in n or "wgrad" in n:
        key = (n[:60], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"), r.get("Workgroup_Size_X", r.get("Workgroup_Size")))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v.sort()
    print("%-62s grid %s x %s wg %s  n=%d  med %.1f us  min %.1f" % (k[0], k[1], k[2], k[3], len(v), v[len(v)//2], v[0]))
PY
tail -12 gpurun_out/gather_trace.log
