#!/bin/bash
# usage: tools/pmc_conv.sh <layer-filter>   (on the GPU box) — SQ counters of the conv kernels for one layer, and (second
# pass) GRBM_GUI_ACTIVE / kernel duration = the effective clock the kernel ran at
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc1 /tmp/pmc2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc1 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --precision bf16 --batch ${BATCH:-64} --only "$1" --iters 2 > /tmp/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc2 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --precision bf16 --batch ${BATCH:-64} --only "$1" --iters 2 > /tmp/pmc2.log 2>&1
python3 - <<PY
import csv,glob,collections
def load(d):
    f=glob.glob(d+"/*counter_collection.csv")[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][28:75]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES","GRBM_GUI_ACTIVE"):
            cnt[k]+=1
            if "End_Timestamp" in r: dur[k]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    return agg,cnt,dur
agg,cnt,_=load("/tmp/pmc1")
for k,v in agg.items():
    if "halo" in k or "igemm" in k or "wgrad" in k or "pipe" in k or "gather" in k or "line64" in k or "s2d" in k:
        n=max(cnt[k],1); w=v["SQ_WAVES"]/n
        wc=v["SQ_WAVE_CYCLES"]/n
        print("%-48s disp=%d waves=%d  cyc/wave=%.0f  wait=%.0f%% inst_stall=%.0f%% active=%.0f%%  mfma_cyc/wave=%.0f  lds_conf/lds=%.2f" % (
            k, cnt[k], w, 4*wc/max(w,1), 100*v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"], 100*v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"],
            100*v["SQ_ACTIVE_INST_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_VALU_MFMA_BUSY_CYCLES"]/n/max(w,1), v["SQ_LDS_BANK_CONFLICT"]/max(v["SQ_LDS_IDX_ACTIVE"],1)))
try:
    agg,cnt,dur=load("/tmp/pmc2")
    print(sorted(set(c for v in agg.values() for c in v)))
    for k,v in agg.items():
        if "halo" in k or "igemm" in k or "wgrad" in k or "pipe" in k or "gather" in k or "line64" in k or "s2d" in k:
            n=max(cnt[k],1)
            print("%-48s disp=%d  gui_active/disp=%.0f  dur_us=%.1f  clock_GHz=%.2f  valu=%.0f lds=%.0f salu=%.0f vmem=%.0f lds_wait=%.0f" % (
                k, n, v["GRBM_GUI_ACTIVE"]/n, dur[k]/n/1e3, v["GRBM_GUI_ACTIVE"]/max(dur[k],1), v["SQ_INSTS_VALU"]/n, v["SQ_INSTS_LDS"]/n, v["SQ_INSTS_SALU"]/n, v["SQ_INSTS_VMEM"]/n, v["SQ_WAIT_INST_LDS"]/n))
except Exception as e:
    print("pass 2 failed:", e); print(open("/tmp/pmc2.log").read()[-1500:])
PY
