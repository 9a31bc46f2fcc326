"""Where does the HOST time of a train() call go?  cProfile over a few steady-state calls of the bench workload
(the GPU runs asynchronously; the profile is the enqueue side).  python tools/host_profile.py [calls=4] [top=45]"""
import argparse, cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ncalls = int(sys.argv[1]) if len(sys.argv) > 1 else 4
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
sys.argv = ["bench.py"]
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
import bench
a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb", precision="bf16")
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb, ops
hb.load_library(); ops.set_precision("bf16")
tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
for _ in range(5):
    tr.train()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(ncalls):
    tr.train()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(top)
