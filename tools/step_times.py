import sys, os, time, torch
sys.argv=['bench.py']
sys.path.insert(0,'/root/repo'); os.chdir('/root/repo')
import bench
import argparse
a=argparse.Namespace(batch=32,image_size=256,gae=2,classifier='resnet',workdir='/tmp/sb',precision='bf16')
sys.path[:0]=[os.path.join('/root/repo','explaining-in-style-reproducibility-study_amd','stylex')]
import ops, hip_backend as hb
hb.load_library(); ops.set_precision('bf16')
dev=torch.device('cuda:0')
tr=bench.build_trainer(a,dev,0,1)
for i in range(4): tr.train()
tr.steps=0
for i in range(8):
    torch.cuda.synchronize(); t=time.perf_counter(); tr.train(); torch.cuda.synchronize()
    print(i, "GP" if i%4==0 else "  ", "%.1f ms" % ((time.perf_counter()-t)*1e3))
