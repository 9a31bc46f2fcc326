"""Summarise a rocprofv3 kernel_stats.csv: per-step ms grouped into families."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
fam = {}
def family(n):
    if "wgrad_pipe" in n: return "wgrad pipe (LDS-DMA)"
    if "conv3x3_pipe" in n: return "pipelined fwd/dgrad"
    if "conv3x3_line64" in n: return "line64 fwd/dgrad (64 ch @256/128 px)"
    if "conv_s2d" in n: return "stride-2 fwd/dgrad (space-to-depth)"
    if "lpips_" in n or "resize_norm" in n or "conv_image_grad" in n or "nhwc_bf16" in n or "relu_gate_add" in n or "affine_act" in n or "affine_relu_maxpool" in n or "nchw_f32_to" in n:
        return "frozen-net kernels (LPIPS taps, resize, image grad, bridges)"
    if "conv_gather" in n: return "gather fwd/dgrad (<= 8 px)"
    if "conv3x3_rgb" in n: return "rgb first layer"
    if "modcoeff" in n or "wsq_kernel" in n: return "modulation coefficients"
    if "conv3x3_halo" in n: return "halo fwd/dgrad"
    if "conv_igemm" in n: return "igemm generic fwd/dgrad"
    if "conv_wgrad_kernel" in n: return "wgrad generic"
    if "conv_wgrad_tr" in n: return "wgrad tr (general)"
    if "act_bwd_reduce" in n: return "fused act-bwd + bias-grad"
    if "subsample2" in n: return "subsample"
    if "wgrad_reduce" in n or "splitk_epilogue" in n: return "split reduce"
    if "pack_weight" in n: return "pack_weight"
    if "bias_act" in n: return "bias_act"
    if "blur3x3" in n: return "blur"
    if "upsample2x" in n: return "upsample"
    if "rowwise" in n or "modconv" in n or "scale_reduce" in n: return "other stylex"
    if "at::native" in n and "reduce" in n: return "torch reduce"
    if "at::native" in n: return "torch elementwise"
    if "miopen" in n.lower() or "igemm_" in n or "naive_conv" in n or "Im2d2Col" in n or "Col2Im" in n or "ck::" in n or "_ZN2ck" in n or "BatchNorm" in n: return "MIOpen (classifier/LPIPS)"
    if n.startswith("Cijk"): return "rocBLAS"
    return "misc"
tot = 0
for r in rows:
    f = family(r["Name"]); t = float(r["TotalDurationNs"]) / 1e6
    fam[f] = fam.get(f, 0) + t; tot += t
print("total %.1f ms  (%.1f ms/step over %g steps)" % (tot, tot / steps, steps))
for f, t in sorted(fam.items(), key=lambda kv: -kv[1]):
    print("  %-40s %8.1f ms/step  %5.1f%%" % (f, t / steps, 100 * t / tot))
# per-kernel rows (rocprofv3's own Calls / AverageNs columns): bench.py's roofline.avg_launch_ms is checked against these
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if top:
    print("per-kernel (top %d by total time; whole run incl. warm-up and instrumented steps):" % top)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        print("%-100s %6s %9.2f ms avg %8.1f us" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
