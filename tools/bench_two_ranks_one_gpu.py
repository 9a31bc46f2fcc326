"""bench.py's N = 2 code path with the real kernels on a 1-GPU box (run ON the GPU box):

    python tools/bench_two_ranks_one_gpu.py [bench.py flags, e.g. --steps 6 --warmup 3]

Both ranks run on cuda:0 and exchange gradients over gloo (RCCL refuses two ranks on one device): the process-group
set-up, the barriers, the MAX-reduced timing, the instrumented roofline steps with their collectives on EVERY rank and the
rank-0-only JSON line are bench.py's own.  The images/s it prints is NOT a scaling number — two ranks share one GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--rank-worker":
    flags = sys.argv[2:]
    sys.path.insert(0, ROOT)
    sys.argv = ["bench.py"]
    import bench

    bench.main(["--gpus", "2"] + flags, backend="gloo")
    sys.exit(0)

port = str(29700 + os.getpid() % 200)
procs = []
for rank in (1, 0):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank-worker"] + sys.argv[1:], env=env, cwd=ROOT,
                                  stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
rc = 0
for p in procs:
    so, se = p.communicate(timeout=1500)
    if p.returncode != 0:
        rc = p.returncode
        sys.stderr.write(se[-3000:])
    if so:
        sys.stdout.write(so)
sys.exit(rc)
