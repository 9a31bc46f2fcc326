"""torgb_bwd on fixed inputs, repeated: is its output bit-stable while another process keeps the GPU busy?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch
import hip_backend as hb
hb.load_library()
dev = "cuda:0"
torch.manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
bad_total = 0
for (b, c, h) in ((64, 64, 128), (64, 32, 256), (64, 128, 64), (64, 512, 16)):
    x = torch.randn(b, c, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(b, 4, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    s1 = torch.randn(b, c, device=dev) + 1
    w = torch.randn(3, c, 1, 1, device=dev)
    ref_gx, ref_t = hb.torgb_bwd(x, gy, s1, w)
    ref_gx, ref_t = ref_gx.clone(), ref_t.clone()
    bad = 0
    for r in range(reps):
        gx, t = hb.torgb_bwd(x, gy, s1, w)
        junk = torch.empty(b * c * h * h // 2, device=dev).normal_()  # allocator churn + another writer on the stream
        if not torch.equal(gx, ref_gx) or not torch.equal(t, ref_t):
            bad += 1
            d = (gx.float() - ref_gx.float())
            nz = d.ne(0)
            print("  rep %d: %d elements differ, max %.4g; first index %s" % (r, int(nz.sum()), float(d.abs().max()), nz.nonzero()[0].tolist()))
        del junk
    print("torgb_bwd %s: %d of %d repeats differ" % ((b, c, h), bad, reps), flush=True)
    bad_total += bad
print("TOTAL", bad_total)
