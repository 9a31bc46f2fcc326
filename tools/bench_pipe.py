"""A/B of the persistent pipelined LDS-DMA conv kernel (conv_pipe.hip) against the per-tile kernel it replaces
(STYLEX_CONV_PIPE=0), interleaved in one process on the discriminator / encoder 3x3 shapes of the 256 px model.
Usage (GPU box): python tools/bench_pipe.py [--batch 64] [--rounds 5] [--iters 10]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402

import hip_backend as hb  # noqa: E402

LAYERS = [  # name, C, N, res
    ("D0.conv2", 64, 64, 256),
    ("D1.conv1", 64, 128, 128),
    ("D1.conv2", 128, 128, 128),
    ("D2.conv1", 128, 256, 64),
    ("D2.conv2", 256, 256, 64),
    ("D3.conv1", 256, 512, 32),
    ("D3.conv2", 512, 512, 32),
]


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--dbg", default="", help="comma list of STYLEX_PIPE_DBG ablation masks timed as extra arms (results wrong by design)")
    a = ap.parse_args()
    P = hb.BF16_ACT
    dev = "cuda:0"
    print("B=%d; ms = median over %d interleaved rounds of %d launches; pipe = conv_pipe.hip, tile = conv_halo_dma.hip" % (
        a.batch, a.rounds, a.iters))
    print("%-9s %4s %4s %4s | %-14s | %8s %8s %6s | %7s %7s | %s" % ("layer", "C", "N", "res", "op", "tile ms", "pipe ms", "x", "tile TF", "pipe TF", "equal"))
    tot = {"tile": 0.0, "pipe": 0.0}
    for (name, c, n, res) in LAYERS:
        if a.only and a.only not in name:
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        mk = lambda *sh: torch.randn(*sh, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
        x, dy, gate = mk(a.batch, c, res, res), mk(a.batch, n, res, res), mk(a.batch, c, res, res)
        w = torch.randn(n, c, 3, 3, device=dev, generator=g) / (9 * c) ** 0.5
        bias = torch.randn(n, device=dev, generator=g)
        bits = (gate.permute(0, 2, 3, 1).float() > 0).reshape(a.batch, res, res, c // 8, 8).to(torch.int32)
        gmask = (bits * (2 ** torch.arange(8, device=dev, dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous()
        del bits
        ops = [("fwd bias+lrelu", lambda: hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True)),
               ("fwd +mask out", lambda: hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True, want_mask=True)[0]),
               ("dgrad+gmask", lambda: hb.conv2d_bwd_data(dy, w, (a.batch, c, res, res), 1, 1, P, gate_mask=gmask)),
               ("dgrad", lambda: hb.conv2d_bwd_data(dy, w, (a.batch, c, res, res), 1, 1, P)),
               ("dgrad+gate", lambda: hb.conv2d_bwd_data(dy, w, (a.batch, c, res, res), 1, 1, P, gate=gate))]
        fl = 2.0 * a.batch * n * res * res * c * 9
        for (op, fn) in ops:
            out = {}
            ts = {"tile": [], "pipe": []}
            for arm in ("tile", "pipe"):
                os.environ["STYLEX_CONV_PIPE"] = "0" if arm == "tile" else "1"
                out[arm] = fn()
            torch.cuda.synchronize()
            for _ in range(a.rounds):
                for arm in ("tile", "pipe"):
                    os.environ["STYLEX_CONV_PIPE"] = "0" if arm == "tile" else "1"
                    ts[arm].append(timeit(fn, a.iters))
            os.environ.pop("STYLEX_CONV_PIPE", None)
            extra = ""
            for d in [d for d in a.dbg.split(",") if d]:
                os.environ["STYLEX_PIPE_DBG"] = d
                fn()
                torch.cuda.synchronize()
                tt = sorted(timeit(fn, a.iters) for _ in range(a.rounds))[a.rounds // 2]
                os.environ.pop("STYLEX_PIPE_DBG", None)
                extra += " dbg%s=%.3f" % (d, tt)
            med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
            eq = bool(torch.equal(out["tile"], out["pipe"]))
            if op in ("fwd bias+lrelu", "dgrad"):
                tot["tile"] += med["tile"]
                tot["pipe"] += med["pipe"]
            print("%-9s %4d %4d %4d | %-14s | %8.3f %8.3f %6.2f | %7.0f %7.0f | %s" % (
                name, c, n, res, op, med["tile"], med["pipe"], med["tile"] / med["pipe"], fl / med["tile"] / 1e9,
                fl / med["pipe"] / 1e9, str(eq) + extra))
    print("sum fwd+dgrad: tile %.3f ms, pipe %.3f ms (x%.2f)" % (tot["tile"], tot["pipe"], tot["tile"] / max(tot["pipe"], 1e-9)))


if __name__ == "__main__":
    main()
