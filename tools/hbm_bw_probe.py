import torch
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (128, 512, 1024, 2048):
    x = torch.empty(mb * 2**20 // 2, dtype=torch.bfloat16, device="cuda")
    y = torch.empty_like(x)
    tf = t(lambda: x.fill_(1.0)); tc = t(lambda: y.copy_(x)); ts = t(lambda: x.sum())
    print("%5d MiB: fill %.3f ms = %.2f TB/s write | copy %.3f ms = %.2f TB/s (r+w) | sum %.3f ms = %.2f TB/s read" % (
        mb, tf, mb * 2**20 / tf / 1e9, tc, 2 * mb * 2**20 / tc / 1e9, ts, mb * 2**20 / ts / 1e9))
