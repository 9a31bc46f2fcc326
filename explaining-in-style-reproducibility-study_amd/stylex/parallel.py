"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The reference's multi-GPU path is vestigial (DDP wrappers around S/G/D only, the encoder is never
wrapped — stylex/stylex_train.py:1188-1193, README.md:81).  Here gradient exchange is explicit:
after the last micro-step of each phase the phase's gradients are packed into flat ~25 MB buckets
(large enough to run the xGMI links at bandwidth, small enough to pipeline), all-reduced
asynchronously, averaged, and scattered back.  Non-final micro-steps never communicate
(== ``no_sync``, :274-285).
"""
import torch
import torch.distributed as dist


def is_dist():
    """True inside an initialised process group (a 1-rank group still runs the collectives: used to smoke-test
    the RCCL path on a single-GPU box)."""
    return dist.is_available() and dist.is_initialized()


def broadcast_parameters(module, src=0):
    """Make every rank start from rank-``src`` weights and buffers."""
    if not is_dist():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


class GradSync:
    def __init__(self, params, bucket_bytes=25 * 1024 * 1024):
        seen, self.params = set(), []
        for p in params:
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                self.params.append(p)
        self.buckets, cur, cur_bytes = [], [], 0
        for p in reversed(self.params):  # reverse order ~ the order gradients become ready
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)

    @torch.no_grad()
    def all_reduce(self):
        if not is_dist():
            return
        world = dist.get_world_size()
        flats, works = [], []
        for bucket in self.buckets:
            parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket]
            flat = torch.cat(parts)
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
            flats.append(flat)
        for bucket, flat, work in zip(self.buckets, flats, works):
            work.wait()
            flat.div_(world)
            off, dst, src = 0, [], []
            for p in bucket:
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    dst.append(p.grad)
                    src.append(g)
                off += n
            if dst:
                torch._foreach_copy_(dst, src)  # one multi-tensor launch per bucket instead of one per parameter


def all_reduce_scalar_flag(flag, device):
    """OR of a boolean across ranks (used so that every rank agrees on a NaN restart)."""
    if not is_dist():
        return flag
    t = torch.tensor([1.0 if flag else 0.0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0)
