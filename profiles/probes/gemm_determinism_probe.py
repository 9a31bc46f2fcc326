"""Are the hipBLASLt GEMMs of the 1x1 residual convs bit-reproducible call to call (same stream, and with a second
stream running other GEMMs concurrently)?"""
import torch

dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(128 * 128 * 128, 8, 64), (128 * 64 * 64, 64, 128), (128 * 32 * 32, 128, 256), (128 * 16 * 16, 256, 512),
          (128 * 8 * 8, 512, 512), (128 * 4 * 4, 512, 512), (128 * 2 * 2, 512, 512), (128 * 1, 512, 512),
          (64 * 64 * 64, 64, 128), (64 * 16 * 16, 256, 512), (64 * 4, 512, 512)]
side = torch.cuda.Stream()
big_a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
for m, c, n in shapes:
    x = torch.randn(m, c, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(n, c, device=dev) / c ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev).to(torch.bfloat16)
    dy = torch.randn(m, n, device=dev, dtype=torch.bfloat16)
    res = {}
    for mode in ("alone", "concurrent"):
        f0 = torch.addmm(b, x, w.t())
        d0 = torch.mm(dy, w)
        nf = nd = 0
        for _ in range(20):
            if mode == "concurrent":
                with torch.cuda.stream(side):
                    torch.mm(big_a, big_a)
            nf += int(not torch.equal(torch.addmm(b, x, w.t()), f0))
            nd += int(not torch.equal(torch.mm(dy, w), d0))
        torch.cuda.synchronize()
        res[mode] = (nf, nd)
    print("M=%8d C=%4d N=%4d  mismatches of 20 (fwd, dgrad): alone %s  concurrent %s" % (m, c, n, res["alone"], res["concurrent"]))
