"""Does the autograd engine protect the LIFETIME of a gradient that is produced on one HIP stream and consumed by a
node on another (caching-allocator record_stream), or only order the two streams?

SideOp.backward (side stream) returns a freshly allocated gradient g; MainOp.backward (main stream) first enqueues a
long sleep, then reads g.  When MainOp.backward returns, g's last reference dies.  If the engine did not record the
consumer stream on g, its block goes straight back to the side stream's pool, a new side-stream allocation reuses and
overwrites it while `main` still sleeps, and x.grad reads the overwritten values.      python tools/probes/engine_stream_probe.py"""
import torch

dev = torch.device("cuda:0")
side = torch.cuda.Stream(device=dev)
N = 1 << 22
seen = {}


class MainOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        seen["ptr"] = g.data_ptr()
        torch.cuda._sleep(int(2e9))  # ~1 s of `main` time before g is read
        return g * 1.0


class SideOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        return a * 1.0

    @staticmethod
    def backward(ctx, g):
        return torch.full((N,), 1.0, device=dev)  # allocated from the side stream's pool


def run(protect=False):
    x = torch.zeros(N, device=dev, requires_grad=True)
    main = torch.cuda.current_stream()
    a = MainOp.apply(x)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        a.record_stream(side)
        b = SideOp.apply(a)
        loss = b.sum()
    main.wait_stream(side)
    loss.backward()
    with torch.cuda.stream(side):
        # same size: the allocator hands back g's block if it is free (several: other free blocks of that size exist)
        hs = [torch.full((N,), 2.0, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    return any(h.data_ptr() == seen["ptr"] for h in hs), float(x.grad.mean())


class MainOp2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        return a * 1.0

    @staticmethod
    def backward(ctx, g):
        torch.cuda._sleep(int(2e9))  # `main` is busy when the engine enqueues the accumulation
        return torch.full((N,), 3.0, device=dev)  # written ~1 s from now


class MainOp0(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        return g * 1.0


class SideOpRec(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        return a * 1.0

    @staticmethod
    def backward(ctx, g):
        out = torch.full((N,), 4.0, device=dev)
        seen["ptr"] = out.data_ptr()
        return out


def run_accumulate(protect=False):
    """a feeds a side-stream op AND a main-stream op: the engine sums the two gradients itself."""
    poison = [torch.full((N,), 100.0, device=dev) for _ in range(6)]  # stale contents of the blocks handed out below
    with torch.cuda.stream(side):
        poison += [torch.full((N,), 100.0, device=dev) for _ in range(6)]
    torch.cuda.synchronize()
    del poison
    x = torch.zeros(N, device=dev, requires_grad=True)
    main = torch.cuda.current_stream()
    a = MainOp0.apply(x)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        a.record_stream(side)
        a_s = a
        b = SideOpRec.apply(a_s)
        lb = b.sum()
    c = MainOp2.apply(a)  # created last: its backward (the sleep) runs first
    main.wait_stream(side)
    (lb + c.sum()).backward()
    with torch.cuda.stream(side):
        hs = [torch.full((N,), 5.0, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    return any(h.data_ptr() == seen["ptr"] for h in hs), float(x.grad.mean())


if __name__ == "__main__":
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path[:0] = [os.path.join(root, "explaining-in-style-reproducibility-study_amd", "stylex"),
                    os.path.join(root, "explaining-in-style-reproducibility-study_amd")]
    reused, val = run(False)
    print("engine only: block reused while consumer pending: %s; x.grad mean = %.3f (1.0 = intact, 2.0 = overwritten)" % (reused, val))
    reused, val = run_accumulate(False)
    print("accumulation, engine only: side gradient's block reused: %s; x.grad mean = %.3f (7.0 = intact)" % (reused, val))

