#!/bin/bash
# usage: tools/pmc_conv.sh <layer-filter>   (on the GPU box) — SQ counters of the conv kernels for one layer
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc1 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --precision bf16 --batch ${BATCH:-64} --only "$1" --iters 2 > /tmp/pmc1.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmc1/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][28:75]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVE_CYCLES": cnt[k]+=1
for k,v in agg.items():
    if "halo" in k or "igemm" in k or "wgrad" in k:
        n=max(cnt[k],1); w=v["SQ_WAVES"]/n
        wc=v["SQ_WAVE_CYCLES"]/n
        print("%-48s disp=%d waves=%d  cyc/wave=%.0f  wait=%.0f%% inst_stall=%.0f%% active=%.0f%%  mfma_cyc/wave=%.0f  lds_conf/lds=%.2f" % (
            k, cnt[k], w, 4*wc/max(w,1), 100*v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"], 100*v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"],
            100*v["SQ_ACTIVE_INST_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_VALU_MFMA_BUSY_CYCLES"]/n/max(w,1), v["SQ_LDS_BANK_CONFLICT"]/max(v["SQ_LDS_IDX_ACTIVE"],1)))
PY
