"""HBM traffic of the dominant conv class from PMC counters (run ON the GPU box):

    python tools/collect_traffic.py [--out profiles/r01_pmc_traffic.json]

Two separate rocprofv3 passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: 3 + 2 of the 4
TCC slots), each `rocprofv3 --kernel-trace --pmc <C> -- python3 bench.py ...` — no sys/hip/hsa trace
domains.  Units and correction as /opt/skills/guides/MI355X_MICROARCH.md §HBM: the counters are in KiB and
FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950 => bytes = (2*FETCH + WRITE)*1024.
The weight-gradient class is cleanly separable by kernel name (forward and data-gradient share kernels)."""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WGRAD = ("conv3x3_wgrad_pipe_kernel", "conv_wgrad_tr_kernel", "conv_wgrad_tr_dma_kernel", "conv_wgrad_kernel")
WGRAD_AUX = ("wgrad_reduce_kernel", "wgrad_reduce_small_kernel", "fold_weight_s2d_kernel")
FWD = ("conv3x3_halo_bf16_kernel", "conv3x3_halo_dma_kernel", "conv_igemm_kernel", "conv_gather_kernel", "conv3x3_pipe_kernel", "conv3x3_line64_kernel", "conv3x3_rgb_kernel", "conv_s2d_dgrad_kernel", "conv_s2d_fwd_kernel", "conv_gather_line_kernel")  # forward and data gradient share these kernels
FWD_AUX = ("splitk_epilogue_kernel", "pack_weight_kernel", "pack_weight_s2d_kernel")


def run_pass(counter, bench_args):
    d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_pmc_traffic.json"))
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    bench_args = ["--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--roofline-steps", "0", "--fp32-steps", "0", "--bench-a-steps", "0", "--precision", a.precision]
    fetch, cnt = run_pass("FETCH_SIZE", bench_args)
    write, _ = run_pass("WRITE_SIZE", bench_args)
    per_kernel = {}
    for k in set(fetch) | set(write):
        if "anonymous namespace" not in k:
            continue
        per_kernel[k[:110]] = {"dispatches": cnt[k], "fetch_KiB_raw": fetch.get(k, 0.0), "write_KiB": write.get(k, 0.0),
                               "bytes_corrected": (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0,
                               "bytes_per_launch": (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0 / max(1, cnt[k])}
    main_disp = sum(v["dispatches"] for k, v in per_kernel.items() if any(w in k for w in WGRAD))
    fam_bytes = sum(v["bytes_corrected"] for k, v in per_kernel.items() if any(w in k for w in WGRAD + WGRAD_AUX))
    fwd_disp = sum(v["dispatches"] for k, v in per_kernel.items() if any(w in k for w in FWD))
    fwd_bytes = sum(v["bytes_corrected"] for k, v in per_kernel.items() if any(w in k for w in FWD + FWD_AUX))
    sys.path.insert(0, ROOT)
    import bench  # csrc_sha16: bench.py only quotes a traffic file collected on the kernel sources it runs

    out = {"csrc_sha16": bench.csrc_sha16(), "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py " + " ".join(bench_args),
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
           "bwd_weight": {"launches": main_disp, "bytes_total": fam_bytes,
                          "bytes_per_launch": fam_bytes / max(1, main_disp)},
           "fwd_bwd_data": {"launches": fwd_disp, "bytes_total": fwd_bytes,
                            "bytes_per_launch": fwd_bytes / max(1, fwd_disp)},
           "kernels": per_kernel}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("bwd_weight", "fwd_bwd_data")}))


if __name__ == "__main__":
    main()
