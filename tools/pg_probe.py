"""Does the mere existence of an RCCL communicator slow the step down?  PG=0|1 python tools/pg_probe.py (GPU box).
Measured: host enqueue unchanged (74 vs 76 ms), free-running step 106.5 vs 103.1 ms with the communicator — a
GPU-side ~3 % that NCCL_MAX_NCHANNELS / NCCL_P2P_DISABLE do not remove."""
import os, sys, time, argparse
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv=['bench.py']; sys.path.insert(0,ROOT); os.chdir(ROOT)
import torch, torch.distributed as dist
import bench
mode=os.environ.get("PG","0")
dev=torch.device("cuda:0"); torch.cuda.set_device(0)
torch.set_num_threads(8)
if mode!="0":
    os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29571")
    dist.init_process_group("nccl", rank=0, world_size=1)
    t=torch.ones(4,device=dev); dist.all_reduce(t)   # create the communicator
a=argparse.Namespace(batch=32,image_size=256,gae=2,classifier='resnet',workdir='/tmp/sb',precision='bf16')
sys.path[:0]=[os.path.join(ROOT,'explaining-in-style-reproducibility-study_amd','stylex')]
import ops, hip_backend as hb
hb.load_library(); ops.set_precision('bf16')
tr=bench.build_trainer(a,dev,0,1)   # is_ddp False: the group just exists
for i in range(8): tr.train()
tr.steps=1
enq=[]; torch.cuda.synchronize()
for i in range(12):
    torch.cuda.synchronize(); t=time.perf_counter(); tr.train(); enq.append((time.perf_counter()-t)*1e3); 
torch.cuda.synchronize(); t=time.perf_counter()
for i in range(12): tr.train()
torch.cuda.synchronize(); free=(time.perf_counter()-t)*1e3/12
import threading
print("PG",mode,"host enqueue ms (median) %.1f"%sorted(enq)[len(enq)//2],"free-running %.1f ms/step"%free,"threads",threading.active_count(), "os threads", len(os.listdir('/proc/self/task')))
