"""Is the to-RGB backward kernel bit-reproducible when other streams keep the GPU busy?  (race hunt)
    python tools/probes/torgb_concurrency_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex"),
                os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")]
import torch  # noqa: E402

import hip_backend as hb  # noqa: E402

dev = "cuda:0"
hb.load_library()
side = torch.cuda.Stream()
g = torch.Generator(device=dev).manual_seed(1)
for (B, C, H) in ((64, 512, 8), (64, 512, 16), (64, 64, 128), (64, 128, 64)):
    x = torch.randn(B, C, H, H, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy_nchw = torch.randn(B, 4, H, H, device=dev, generator=g).to(torch.bfloat16)
    s1 = torch.rand(B, C, device=dev, generator=g) + 0.5
    w = torch.randn(3, C, 1, 1, device=dev, generator=g)
    gx0, t0 = hb.torgb_bwd(x, gy_nchw.contiguous(memory_format=torch.channels_last), s1, w)
    torch.cuda.synchronize()
    big = torch.randn(64, 64, 256, 256, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bad = 0
    for it in range(100):
        main = torch.cuda.current_stream()
        junk = [hb.blur3x3_fwd(big) for _ in range(3)]  # keeps `main` busy
        side.wait_stream(main)
        with torch.cuda.stream(side):
            z = torch.zeros(B, 4, H, H, device=dev, dtype=torch.bfloat16)
            z[:, :3].copy_(gy_nchw[:, :3])
            z[:, 3:].copy_(gy_nchw[:, 3:])
            gx, t = hb.torgb_bwd(x, z.contiguous(memory_format=torch.channels_last), s1, w)
            scratch = [torch.empty_like(gx).normal_() for _ in range(2)]  # churn the side pool
        main.wait_stream(side)
        gx.record_stream(main)
        t.record_stream(main)
        ok = torch.equal(gx, gx0) and torch.equal(t, t0)
        bad += 0 if ok else 1
        del junk, scratch, z
    torch.cuda.synchronize()
    print("B=%d C=%d H=%d: %d of 100 concurrent launches differ from the serial result" % (B, C, H, bad))
