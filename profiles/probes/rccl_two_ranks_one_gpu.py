"""Can two RCCL ranks share ONE GPU (to exercise the N>1 code path on the 1-GPU boxes of this environment)?
python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/probes/rccl_two_ranks_one_gpu.py"""
import os
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world)
t = torch.full((1024,), float(rank + 1), device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.AVG)
torch.cuda.synchronize()
print("rank", rank, "all_reduce AVG ->", t[0].item(), flush=True)
dist.destroy_process_group()
