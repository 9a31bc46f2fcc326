"""Frozen ResNet-18 (224 px, B=64) forward + input gradient on stock PyTorch / MIOpen: fp32 NCHW (today) vs bf16
autocast in NCHW / channels_last; time per fwd+bwd and the error of logits and input gradient vs fp32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch
from tv_models import ResNet18
torch.backends.cudnn.benchmark = True
dev = "cuda:0"
torch.manual_seed(0)
m = ResNet18().to(dev).eval()
m.fc = torch.nn.Linear(512, 2).to(dev)
for p in m.parameters(): p.requires_grad = False
x0 = torch.randn(64, 3, 224, 224, device=dev)
def run(mode):
    x = x0.clone().requires_grad_()
    if mode == "fp32":
        y = m(x)
    elif mode == "bf16_nchw":
        with torch.autocast("cuda", dtype=torch.bfloat16): y = m(x)
    elif mode == "bf16_cl":
        with torch.autocast("cuda", dtype=torch.bfloat16): y = m(x.contiguous(memory_format=torch.channels_last))
    elif mode == "fp16_cl":
        with torch.autocast("cuda", dtype=torch.float16): y = m(x.contiguous(memory_format=torch.channels_last))
    y = y.float()
    (y[:, 0] - y[:, 1]).sum().backward()
    return y.detach(), x.grad
def timeit(mode, n=10):
    for _ in range(3): run(mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run(mode)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
yr, gr = run("fp32")
for mode in ("fp32", "bf16_nchw", "bf16_cl", "fp16_cl"):
    y, g = run(mode)
    t = timeit(mode)
    print("%-10s %.2f ms fwd+bwd | logits rel err %.2e | input-grad rel L2 err %.2e" % (
        mode, t, float((y - yr).norm() / yr.norm()), float((g - gr).norm() / gr.norm())))
