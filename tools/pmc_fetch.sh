#!/bin/bash
# usage: tools/pmc_fetch.sh <layer-filter>  (GPU box) — HBM-side bytes per launch of the conv kernels of one layer
# (FETCH_SIZE and WRITE_SIZE in separate passes; bytes = (2*FETCH + WRITE) * 1024 as in tools/collect_traffic.py)
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcf_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcf_$C -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --precision bf16 --batch ${BATCH:-64} --only "$1" --iters 2 > /tmp/pmcf.log 2>&1
done
python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for C in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("/tmp/pmcf_%s/*counter_collection.csv"%C)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]!=C: continue
        k=r["Kernel_Name"][28:80]
        tot[k][C]+=float(r["Counter_Value"])
        if C=="FETCH_SIZE": cnt[k]+=1
for k,v in tot.items():
    if "halo" in k or "igemm" in k or "wgrad" in k:
        n=max(cnt[k],1)
        print("%-52s launches=%d  fetch %.1f MB  write %.1f MB per launch" % (k, n, 2*v["FETCH_SIZE"]*1024/n/1e6, v["WRITE_SIZE"]*1024/n/1e6))
PY
