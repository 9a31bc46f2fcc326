"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The reference's multi-GPU path is vestigial (DDP wrappers around S/G/D only, the encoder is never
wrapped — stylex/stylex_train.py:1188-1193, README.md:81).  Here gradient exchange is explicit:
the gradients of a phase are packed into flat 32 MB buckets and all-reduced (averaging done by the collective,
ReduceOp.AVG on RCCL) from inside the last backward of the phase, bucket by bucket as they complete, then
scattered back (class GradSync).  Non-final micro-steps never communicate
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
    """Bucketed gradient all-reduce, overlapped with the backward pass.

    Parameters are grouped into flat buckets in reverse order (~ the order their gradients become ready).  `arm()`
    before the last backward of a phase turns on per-parameter post-accumulate hooks: when every gradient of the
    next bucket IN INDEX ORDER is ready its all-reduce is launched from inside the backward, so the xGMI transfer
    runs under the rest of the backward (a 2-GPU all-reduce of the 400 MB of a step is ~7 ms on one xGMI link,
    ~6 % of a step when issued afterwards).  Buckets are launched strictly in index order on every rank — the
    collective sequence never depends on which gradient happened to arrive first.  `all_reduce()` after the
    backward launches what is left (parameters without a gradient contribute zeros), waits, and scatters the
    averaged gradients back.  Without `arm()` everything is launched by `all_reduce()` (same result)."""

    def __init__(self, params, bucket_bytes=None, overlap=None):
        # Default: launch after the backward, few large buckets (every collective has ~100 us of launch latency; the
        # 1-rank RCCL path measured 589 images/s this way and 576 with in-backward launches, which cannot pay off
        # without a second GPU to talk to).  STYLEX_DDP_OVERLAP=1 (or overlap=True) selects the in-backward launch
        # with 32 MB buckets — to be judged on real multi-GPU scaling numbers.
        import os

        self.overlap = (os.environ.get("STYLEX_DDP_OVERLAP", "0") == "1") if overlap is None else bool(overlap)
        if bucket_bytes is None:
            bucket_bytes = (32 if self.overlap else 128) * 1024 * 1024
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
        self._bucket_of = {id(p): bi for bi, bucket in enumerate(self.buckets) for p in bucket}
        self._armed = False
        self._reset()
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def _reset(self):
        self._ready = [0] * len(self.buckets)
        self._events = [[] for _ in self.buckets]
        self._next = 0
        self._flats, self._works = [], []

    def arm(self):
        """Call right before the last backward of the phase (earlier micro-step backwards only accumulate)."""
        if is_dist() and self.overlap:
            self._reset()
            self._armed = True

    def _on_grad(self, p):
        if not self._armed:
            return
        bi = self._bucket_of[id(p)]
        if p.grad is not None and p.grad.is_cuda:
            ev = torch.cuda.Event()  # gradients of one bucket may be produced on different HIP streams
            ev.record()
            self._events[bi].append(ev)
        self._ready[bi] += 1
        while self._next < len(self.buckets) and self._ready[self._next] == len(self.buckets[self._next]):
            self._launch(self._next)

    @torch.no_grad()
    def _launch(self, bi):
        assert bi == self._next
        bucket = self.buckets[bi]
        if self._events[bi]:
            cur = torch.cuda.current_stream()
            for ev in self._events[bi]:
                cur.wait_event(ev)
        parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket]
        flat = torch.cat(parts)
        avg = flat.is_cuda  # RCCL averages inside the collective; gloo (CPU tests) has no AVG
        self._works.append(dist.all_reduce(flat, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True))
        self._flats.append(flat)
        self._next += 1

    @torch.no_grad()
    def all_reduce(self):
        if not is_dist():
            return
        if not self._armed:
            self._reset()
        self._armed = False
        world = dist.get_world_size()
        while self._next < len(self.buckets):
            self._launch(self._next)
        for bucket, flat, work in zip(self.buckets, self._flats, self._works):
            work.wait()
            if not flat.is_cuda:
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
        self._reset()


def all_reduce_scalar_flag(flag, device):
    """OR of a boolean across ranks (used so that every rank agrees on a NaN restart)."""
    if not is_dist():
        return flag
    t = torch.tensor([1.0 if flag else 0.0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0)
