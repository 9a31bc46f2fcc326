"""bench.py — StylEx G+D+enc train-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W              # one GPU
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one ``Trainer.train()`` call = one D optimiser step + one G optimiser step over
``gradient_accumulate_every`` micro-batches per phase (reference stylex/stylex_train.py:1249-1506).
Workload (Bench B = the metric's "G+D+enc step"): FFHQ-shaped synthetic 256x256
batches, 32 images per GPU per micro-batch, GAE=2 with alternating training => one noise and one
encoder micro-step per phase; steps start at 0 so one call in four carries the gradient penalty.
Inputs are device-resident before the timed region.  images/s counts B*GAE images per call.

Prints ONE JSON line (rank 0) with the contract keys plus
  roofline     — dominant conv KERNEL (rocprofv3 name), algorithmic FLOPs / hipEvent time, vs the dense MFMA peak;
                 its kernel class, the per-kernel table and the per-layer mixed roofline beside it
  fp32_parity_mode — the same workload in the exact-fp32 mode (images/s, fraction of the fp32 MFMA peak)
  cpu_baseline — the CPU oracle ("port") timed on the host cores on a bounded sample (N=1 only)
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
for p in (os.path.join(PKG, "stylex"), PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "bf16_f32act": 2500.0, "fp32": 157.3}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0  # HBM3E spec peak (same guide; 6.29 TB/s is the best measured copy rate)


def csrc_sha16():
    """Hash of the kernel sources, their build flags and the C-ABI header: a PMC traffic file is only quoted for the code it was measured on."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.h"))
                    + glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(PKG, "csrc", "Makefile")]):  # (+ build flags)
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_traffic(class_name, precision, kernel=None):
    """HBM bytes per launch from the newest profiles/r*_pmc_traffic.json that was collected on EXACTLY these kernel
    sources (tools/collect_traffic.py stamps the file with csrc_sha16); None otherwise — a stale counter file is never
    quoted.  With `kernel` (a rocprofv3 kernel name) the per-kernel entry of the file's "kernels" table, else the
    figure of the whole class."""
    import glob

    want = csrc_sha16()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            tj = json.load(open(f))
        except Exception:
            continue
        if tj.get("csrc_sha16") != want or precision not in tj.get("command", ""):
            continue
        if kernel is not None:
            for name, row in tj.get("kernels", {}).items():
                if kernel in name:
                    return round(row["bytes_per_launch"]), os.path.basename(f)
            continue
        if class_name in tj:
            return round(tj[class_name]["bytes_per_launch"]), os.path.basename(f)
    return None, None


def layer_roofline(layers, precision, top=14):
    """Per-layer mixed roofline: each (class, conv shape) is priced against the roof that bounds it —
    MFMA when its algorithmic intensity FLOPs/byte is above the ridge peak_flops/peak_bw, HBM below — and the class
    total against the sum of those per-layer minimum times."""
    peak_f, peak_b = PEAK_TFLOPS[precision] * 1e12, PEAK_HBM_GBS * 1e9
    rows, t_min, t_all = [], 0.0, 0.0
    for L in layers:
        if L["ms"] <= 0 or L["launches"] == 0:
            continue
        sec = L["ms"] * 1e-3
        tf, tb = L["flops"] / peak_f, L["bytes"] / peak_b
        bound = "mfma" if tf >= tb else "hbm"
        t_min += max(tf, tb)
        t_all += sec
        rows.append({"layer": "%s %d->%d k%d s%d%s @%dx%d B=%d" % (L["cls"], L["C"] // (4 if L["s2d"] else 1), L["N"], L["k"],
                                                                   2 if L["s2d"] else L["stride"], " s2d" if L["s2d"] else "",
                                                                   L["H"] * (2 if L["s2d"] else 1), L["W"] * (2 if L["s2d"] else 1), L["B"]),
                     "bound": bound, "launches": L["launches"], "ms_per_launch": round(L["ms"] / L["launches"], 4),
                     "tflops": round(L["flops"] / sec / 1e12, 1), "gbs": round(L["bytes"] / sec / 1e9, 0),
                     "frac": round(max(tf, tb) / sec, 3), "ms_total": round(L["ms"], 2)})
    rows.sort(key=lambda r: -r["ms_total"])
    return {"mixed_frac": round(t_min / t_all, 4) if t_all else None,
            "note": "sum over layers of max(FLOPs/peak_mfma, bytes/peak_hbm) / measured time; per-layer frac likewise",
            "layers": rows[:top]}


def conv_gflop_forward(image_size, network_capacity=16, fmap_max=512):
    """Forward conv GFLOP per image of G, D and the encoder from the layer lists (reference Generator :747-825,
    DiscriminatorE :842-909): 17.767 / 35.511 / 35.513 @256 px, 4.46 / 8.88 @128, 1.13 / 2.22 @64."""
    from math import log2

    n = int(log2(image_size) - 1)
    f = [min(fmap_max, network_capacity * 2 ** (i + 1)) for i in range(n)][::-1]
    gfil = [f[0]] + f
    g = 2 * 4 * 4 * gfil[0] * gfil[0] * 9  # initial_conv
    for i in range(n):
        res, cin, cout = 4 * 2 ** i, gfil[i], gfil[i + 1]
        g += 2 * res * res * (cin * cout * 9 + cout * cout * 9 + cout * 3)  # conv1, conv2, to_rgb
    dfil = [3] + [min(fmap_max, 4 * network_capacity * 2 ** i) for i in range(n + 1)]
    d = 0
    for i in range(len(dfil) - 1):
        res, cin, cout, last = image_size // 2 ** i, dfil[i], dfil[i + 1], i == len(dfil) - 2
        ro = res if last else res // 2
        d += 2 * ro * ro * cin * cout + 2 * res * res * (cin * cout * 9 + cout * cout * 9)  # conv_res, conv1, conv2
        if not last:
            d += 2 * ro * ro * cout * cout * 9  # down-sampling conv
    rl = image_size // 2 ** (len(dfil) - 2)
    d += 2 * rl * rl * dfil[-1] * dfil[-1] * 9  # final_conv
    e = d + 2 * (rl * rl * dfil[-1]) * 511  # the encoder's fc is 512 wide instead of 1
    return g / 1e9, d / 1e9, e / 1e9


def algorithmic_gflop_per_image(gae, image_size=256, network_capacity=16, fmap_max=512):
    """Bench A (GAE=1) 4G+8.75D = 381.8 @256 px; Bench B (GAE=2) (8G+17.5D+7E)/2 = 506.1 @256 px,
    126.6 @128 px, 31.7 @64 px (round 4 used the 256 px constants at every --image-size)."""
    G_FWD, D_FWD, E_FWD = conv_gflop_forward(image_size, network_capacity, fmap_max)
    if gae == 1:
        return 4 * G_FWD + 8.75 * D_FWD
    noise = (G_FWD + 6 * D_FWD) + (3 * G_FWD + 2 * D_FWD) + 0.75 * D_FWD
    enc = (E_FWD + G_FWD + 6 * D_FWD) + (6 * E_FWD + 3 * G_FWD + 2 * D_FWD) + 0.75 * D_FWD
    return (noise * (gae - gae // 2) + enc * (gae // 2)) / gae


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def build_trainer(args, device, rank, world):
    import stylex_train as st

    gen = torch.Generator().manual_seed(7 + rank)
    ring = [torch.rand(args.batch, 3, args.image_size, args.image_size, generator=gen).to(device) for _ in range(8)]
    seed_all(42)
    tr = st.Trainer(name="bench", base_dir=args.workdir, image_size=args.image_size,
                    network_capacity=getattr(args, "network_capacity", 16), fmap_max=getattr(args, "fmap_max", 512), batch_size=args.batch * world, gradient_accumulate_every=args.gae, lr=2e-4,
                    ttur_mult=1.5, mixed_prob=0.9, rec_scaling=1, kl_scaling=1, aug_prob=0.,
                    alternating_training=True, classifier_name=args.classifier, classifier_path=None,
                    evaluate_every=10 ** 9, save_every=10 ** 9, tensorboard_dir=None,
                    is_ddp=world > 1 or os.environ.get("STYLEX_FORCE_DDP") == "1", rank=rank,
                    world_size=world, device=device, graphs=bool(getattr(args, "graphs", 0)),
                    pl_every=int(getattr(args, "pl_every", 32)),
                    device_rng=(bool(getattr(args, "device_rng", 0)) and device.type == "cuda") or None)
    tr.loader = st.cycle(ring)
    tr.dataset = list(range(10 ** 6))
    tr.save = lambda *a, **k: None
    tr.evaluate = lambda *a, **k: None
    tr.init_StylEx()
    seed_all(42 + rank)  # rank-distinct latents/noise, rank-identical weights (broadcast in init_StylEx)
    return tr


def fp32_record(args, device, tr_old):
    """The exact-fp32 parity mode (the mode every golden-vector test runs in) on the same workload: images/s over
    `--fp32-steps` calls starting on a gradient-penalty step, and the conv class fraction of the fp32 MFMA peak."""
    import copy

    import hip_backend as hb
    import ops

    del tr_old
    torch.cuda.empty_cache()
    a = copy.copy(args)
    a.graphs = 0
    prev = ops.get_precision()
    ops.set_precision("fp32")
    try:
        tr = build_trainer(a, device, 0, 1)
        for _ in range(2):
            tr.train()
        tr.steps = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.fp32_steps):
            tr.train()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prev_streams = os.environ.get("STYLEX_STREAMS")
        os.environ["STYLEX_STREAMS"] = "0"
        hb.timing_enable(1)
        tr.steps = 1
        tr.train()
        torch.cuda.synchronize()
        rep = hb.timing_report()
        hb.timing_enable(0)
        if prev_streams is None:
            os.environ.pop("STYLEX_STREAMS", None)
        else:
            os.environ["STYLEX_STREAMS"] = prev_streams
        del tr
        torch.cuda.empty_cache()
    finally:
        ops.set_precision(prev)
    ms = rep["fwd"]["ms"] + rep["bwd_data"]["ms"]
    fl = rep["fwd"]["flops"] + rep["bwd_data"]["flops"]
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    value = args.batch * args.gae * args.fp32_steps / dt
    return {"value": round(value, 2), "unit": "images/sec", "steps": args.fp32_steps,
            "ms_per_step": round(dt / args.fp32_steps * 1e3, 2), "dtype": "fp32",
            "class_achieved_tflops": round(ach, 2), "peak": PEAK_TFLOPS["fp32"],
            "class_frac": round(ach / PEAK_TFLOPS["fp32"], 4),
            "whole_step_frac": round(algorithmic_gflop_per_image(args.gae, args.image_size, args.network_capacity, args.fmap_max) * value / 1e3 / PEAK_TFLOPS["fp32"], 4),
            "note": "same workload in the exact-fp32 MFMA mode of the parity tests; class = forward + data-gradient conv "
                    "launches of one plain instrumented step"}


def bench_a_record(args, device):
    """Bench A of SURVEY §8(d) — the "G+D step" BASELINE.md §4 quotes the >= 60 % roofline target on: the same
    workload at gradient_accumulate_every = 1 (noise micro-step only), timed like the headline (N = 1 only)."""
    import copy

    a = copy.copy(args)
    a.gae, a.graphs = 1, 0
    tr = build_trainer(a, device, 0, 1)
    for _ in range(4):
        tr.train()
    tr.steps = args.start_step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.bench_a_steps):
        tr.train()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del tr
    torch.cuda.empty_cache()
    value = args.batch * args.bench_a_steps / dt
    gf = algorithmic_gflop_per_image(1, args.image_size, args.network_capacity, args.fmap_max)
    return {"workload": "Bench A: same shapes, gradient_accumulate_every = 1 (noise micro-step only), GP every 4th call",
            "value": round(value, 2), "unit": "images/sec", "steps": args.bench_a_steps,
            "ms_per_step": round(dt / args.bench_a_steps * 1e3, 2), "algorithmic_conv_gflop_per_image": round(gf, 1),
            "step_conv_tflops": round(gf * value / 1e3, 2),
            "whole_step_frac": round(gf * value / 1e3 / PEAK_TFLOPS[args.precision], 4)}


def frozen_split(args, tr, device, iters=6):
    """How the frozen networks' time splits (round-4 VERDICT item 7c): the classifier FORWARD (the part north_star pins
    to stock PyTorch-ROCm), the classifier's data-gradient pass and LPIPS-AlexNet (forward on two batches + backward),
    each timed alone on the benchmark's batch shape with hipEvents and weighted with its calls per train() call
    (encoder micro-step: classifier forward x3 of which one is differentiated, LPIPS x1; stylex_train.py:1086-1145)."""
    import stylex_train as st

    g = torch.Generator(device=device).manual_seed(3)
    x = torch.rand(args.batch, 3, args.image_size, args.image_size, device=device, generator=g)
    y = torch.rand(args.batch, 3, args.image_size, args.image_size, device=device, generator=g)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def cls_fwd():
        with torch.no_grad():
            tr.classifier.classify_images(st._frozen_layout(x))

    def cls_fwd_bwd():
        xi = x.clone().requires_grad_(True)
        tr.classifier.classify_images(st._frozen_layout(xi)).sum().backward()

    def lp_fwd_bwd():
        yi = y.clone().requires_grad_(True)
        st.perceptual_loss(x, yi, tr.lpips_fn).backward()

    t_f, t_fb, t_lp = timed(cls_fwd), timed(cls_fwd_bwd), timed(lp_fwd_bwd)
    enc_steps = args.gae // 2  # encoder micro-steps per phase; the classifier / LPIPS run in the generator phase only
    return {"classifier_forward_ms_per_call": round(t_f, 3), "classifier_forward_backward_ms_per_call": round(t_fb, 3),
            "lpips_forward_backward_ms_per_call": round(t_lp, 3),
            "per_step_ms": {"classifier_forward (stock, north_star)": round(enc_steps * (2 * t_f + min(t_f, t_fb)), 3),
                            "classifier_data_gradient": round(enc_steps * max(0.0, t_fb - t_f), 3),
                            "lpips_alexnet": round(enc_steps * t_lp, 3)},
            "note": "each part alone on the GPU (no overlap with other streams), batch %d @%d px, %d encoder micro-step(s) "
                    "per train() call" % (args.batch, args.image_size, enc_steps)}


def cpu_baseline(args):
    """The CPU oracle (oracle/stylex_oracle.py, a port of the reference's CPU path) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import stylex_oracle as so
    from lpips_standin import LPIPSStandIn
    from standins import TinyClassifier

    # the oracle's grouped convolutions stop scaling (and collapse under oversubscription) beyond a
    # handful of threads: 256 threads on the GPU box measured 60x SLOWER than 8.  Use <= 16.
    cores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(cores)
    bs = args.cpu_batch
    gen = torch.Generator().manual_seed(7)
    ring = [torch.rand(bs, 3, args.image_size, args.image_size, generator=gen) for _ in range(4)]

    def cyc():
        while True:
            for b in ring:
                yield b

    seed_all(42)
    tr = so.OracleTrainer(TinyClassifier(seed=99), LPIPSStandIn(seed=4242), cyc(), image_size=args.image_size,
                          network_capacity=16, fmap_max=512, batch_size=bs, gradient_accumulate_every=args.gae,
                          lr=2e-4, ttur_mult=1.5, rec_scaling=1, kl_scaling=1)
    # SURVEY §8(d) protocol, bounded: one untimed warm-up call (allocator / oneDNN primitive caches), then ONE call with
    # the gradient penalty and ONE without, timed separately and combined in the 1:3 ratio of the real schedule
    # (steps % 4 == 0 carries the penalty) — ~30 s of CPU work instead of the 15 calls (minutes) the full protocol takes.
    tr.steps = 1
    tr.train()
    t_gp = t_plain = 0.0
    tr.steps = 4
    t0 = time.time()
    tr.train()
    t_gp = time.time() - t0
    t0 = time.time()
    tr.train()
    t_plain = time.time() - t0
    per4 = t_gp + 3 * t_plain
    return {"value": 4 * bs * args.gae / per4, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "CPU oracle Trainer.train() at %dpx, batch %d, GAE %d: 1 warm-up call, then 1 gradient-penalty call "
                      "(%.1f s) + 1 plain call (%.1f s), combined 1:3 as in the real schedule; stand-in classifier/LPIPS"
                      % (args.image_size, bs, args.gae, t_gp, t_plain)}


def main(argv=None, backend="nccl", device=None):
    """`backend` / `device` are overridden only by tests/test_ddp_gloo.py (2 ranks over gloo on the CPU test double): the
    N > 1 plumbing below — barrier, MAX-reduce of the elapsed time, rank-0-only JSON line, process-group teardown — is
    then the code a real `--gpus N` run executes, minus the kernels."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--precision", default=os.environ.get("STYLEX_PRECISION", "bf16"), choices=["bf16", "bf16_f32act", "fp32"])
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per micro-batch")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--gae", type=int, default=2, help="gradient_accumulate_every (2 = noise + encoder micro-step)")
    ap.add_argument("--classifier", default="resnet")
    ap.add_argument("--network-capacity", type=int, default=16)
    ap.add_argument("--fmap-max", type=int, default=512)
    ap.add_argument("--workdir", default="/tmp/stylex_bench")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--miopen-find", action="store_true",
                    help="let MIOpen search its algorithms for the frozen classifier / LPIPS during the warm-up "
                         "(cudnn.benchmark; default: immediate mode, the reference's cli.py:38 setting and the Trainer's)")
    ap.add_argument("--no-miopen-find", action="store_true", help="(default since round 4; accepted for old scripts)")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--roofline-steps", type=int, default=4)
    ap.add_argument("--per-layer-top", type=int, default=14, help="rows of roofline.per_layer.layers (0 = every layer)")
    ap.add_argument("--fp32-steps", type=int, default=4,
                    help="timed train() calls of the fp32 parity mode for the fp32_parity_mode sub-record (0 = skip)")
    ap.add_argument("--bench-a-steps", type=int, default=12,
                    help="timed train() calls of Bench A (GAE = 1) for the bench_a sub-record (0 = skip; N = 1 only)")
    ap.add_argument("--graphs", type=int, default=int(os.environ.get("STYLEX_GRAPHS", "0")),
                    help="1 = replay the step as captured HIP graphs after the eager warm-up calls (0 = eager enqueue, "
                         "the default: at 256 px the step is GPU-bound and the capture of the multi-stream step is not "
                         "stable on ROCm 7.2, see DESIGN.md)")
    ap.add_argument("--pl-every", type=int, default=32,
                    help="path-length regularisation interval (reference: 32; BASELINE config 4 says 16); active from "
                         "--start-step > 5000 on")
    ap.add_argument("--start-step", type=int, default=0,
                    help="Trainer.steps at the start of the timed region (0: one call in 4 carries the gradient penalty, "
                         "no path-length steps; 5024: path-length steps every --pl-every calls as well)")
    ap.add_argument("--device-rng", type=int, default=1, choices=[0, 1],
                    help="1 (default): latents / noise planes drawn on the GPU generator — what every rank of the data-parallel "
                         "path does (Trainer: device_rng defaults to True under DDP), so the N = 1 line is the same program as "
                         "the N > 1 lines of the scaling curve; 0: the reference's CPU-generator draws + upload "
                         "(stylex_train.py:319-337; the Trainer's single-GPU default, what every parity fixture pins)")
    ap.add_argument("--host-share", type=int, default=int(os.environ.get("STYLEX_HOST_SHARE", "1")),
                    help="emulate the host share of one rank on an N-GPU node: pin this process (before anything touches "
                         "the GPU; no re-exec) to 1/N of the cores it may run on")
    args = ap.parse_args(argv)
    if args.host_share > 1:
        cores = sorted(os.sched_getaffinity(0))
        keep = cores[:max(1, len(cores) // args.host_share)]
        os.sched_setaffinity(0, keep)  # threads started later (autograd engine, prefetch, BLAS pools) inherit it

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    on_gpu = device is None
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs the MI355X"
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda:%d" % local_rank)

    def dev_sync():
        if on_gpu:
            torch.cuda.synchronize()

    # host side of a rank = kernel launches + the CPU RNG draws: a few threads per rank, not one pool per core per rank
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // max(1, world))))
    force_ddp = os.environ.get("STYLEX_FORCE_DDP") == "1"  # 1-rank RCCL smoke test of the N>1 code path
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # no device_id: the eager communicator set-up it triggers cost 3 % of step throughput on the 1-rank RCCL
        # path (584 vs 604 images/s); torch.cuda.set_device above already binds the rank to its GPU
        dist.init_process_group(backend, rank=rank, world_size=world)

    import hip_backend as hb
    import ops

    if on_gpu:
        hb.load_library()  # no fallback: fail loudly if the extension is missing
        ops.set_precision(args.precision)
    # The frozen classifier / LPIPS convolutions run on stock MIOpen in IMMEDIATE mode — the reference's setting
    # (cli.py:38) and what a `cli.py` training run of this package uses, so the headline number is the one a user gets.
    # Rounds 1-3 let MIOpen search during the warm-up (cudnn.benchmark); round 4 measured the two modes equal on this
    # workload (818 immediate vs 805-814 searched images/s, profiles/r04_c_ab_miopen_modes.txt) and found the exhaustive
    # search unsafe on this stack (head of stylex/hip_backend.py).  --miopen-find restores the search.
    os.environ["STYLEX_MIOPEN_BENCHMARK"] = "1" if args.miopen_find else "0"
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    tr = build_trainer(args, device, rank, world)

    def sync():
        if world > 1:
            dist.barrier()
        dev_sync()

    for _ in range(args.warmup):
        tr.train()
    # graph mode: both step shapes (with / without the gradient penalty) must have been captured before the timed
    # region starts; with the default W=8 they are (4 eager calls, then one capture each) — a shorter warm-up gets
    # the missing untimed calls here
    extra = 0
    while tr.graphs and len(tr._graph_cache) < 2 and extra < 12:
        tr.train()
        extra += 1
    tr.steps = args.start_step  # the timed region starts on a GP step: 1 call in 4 carries the penalty
    if args.start_step > 5000 and tr.pl_mean is None:
        tr.pl_mean = 1.0  # as after the first path-length step of a real run
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    images = world * args.batch * args.gae * args.steps
    value = images / dt

    # ---- roofline of the dominant conv kernel class, hipEvent-timed on the launch stream
    roof = None
    if args.roofline_steps > 0:  # on EVERY rank: train() contains the gradient all-reduce collectives
        # kernel quality is measured with the branch-level stream concurrency of the Trainer switched off, so that an
        # event pair brackets one kernel class running alone (the timed region above runs with it on)
        prev_streams = os.environ.get("STYLEX_STREAMS")
        os.environ["STYLEX_STREAMS"] = "0"
        tr.graphs = False  # the per-launch hipEvent pairs are recorded by the C-ABI calls: eager enqueue
        for _ in range(2):
            tr.train()
        torch.cuda.synchronize()
        hb.timing_enable(1)
        tr.steps = 0
        for _ in range(args.roofline_steps):
            tr.train()
        torch.cuda.synchronize()
        rep = hb.timing_report()
        layers = hb.timing_layers()
        kernels = hb.timing_kernels()
        hb.timing_enable(0)
        if prev_streams is None:
            os.environ.pop("STYLEX_STREAMS", None)
        else:
            os.environ["STYLEX_STREAMS"] = prev_streams
        # forward and data-gradient launches run the SAME kernels (LDS-halo / implicit-GEMM conv with swapped
        # roles), so they form one kernel class; the weight gradient has its own kernels
    if args.roofline_steps > 0 and rank == 0:
        merged = {"fwd_bwd_data": {k: rep["fwd"][k] + rep["bwd_data"][k] for k in ("ms", "flops", "launches", "bytes")},
                  "bwd_weight": rep["bwd_weight"]}
        name, r = max(merged.items(), key=lambda kv: kv[1]["ms"])
        ach = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0
        peak = PEAK_TFLOPS[args.precision]
        # HBM bytes per launch of that class from the PMC passes of tools/collect_traffic.py (committed under
        # profiles/); null when no measurement exists for the class / precision
        traffic, traffic_file = load_traffic(name, args.precision)
        cls_layers = [L for L in layers if (L["cls"] == "bwd_weight") == (name == "bwd_weight")]
        # the dominant KERNEL (most hipEvent time over the instrumented steps; forward and data-gradient launches of one
        # kernel are one row), named as rocprofv3 --kernel-trace --stats prints it so that its average launch duration
        # can be checked against profiles/r*_kernel_stats_bench_*.csv; the class it belongs to stays beside it
        by_kernel = {}
        for K in kernels:
            if not K["kernel"]:
                continue
            a = by_kernel.setdefault(K["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            for f in ("ms", "flops", "bytes", "launches"):
                a[f] += K[f]
        ktable = sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"])
        kname, kr = ktable[0] if ktable else ("conv_%s" % name, r)
        k_ach = kr["flops"] / (kr["ms"] * 1e-3) / 1e12 if kr["ms"] > 0 else 0.0
        k_traffic, k_file = load_traffic(name, args.precision, kernel=kname)
        what = "PMC passes in profiles/%s (same kernel sources, csrc_sha16 %s)"
        roof = {"bound": "mfma", "kernel": kname, "achieved": round(k_ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(k_ach / peak, 4), "traffic": k_traffic, "launches": kr["launches"],
                "avg_launch_ms": round(kr["ms"] / max(1, kr["launches"]), 4),
                "algorithmic_bytes": round(kr["bytes"] / max(1, kr["launches"])),
                "ms_per_step": round(kr["ms"] / args.roofline_steps, 3),
                "note": "dominant kernel by hipEvent time on the launch stream over %d instrumented steps (stream "
                        "concurrency off), name as rocprofv3 prints it; achieved = algorithmic FLOPs of its launches / "
                        "their summed duration; traffic: %s.  `class` = every launch of the kernel class it belongs to "
                        "(the figure rounds 1-3 reported as roofline.frac), `whole_step_frac` = algorithmic conv FLOPs "
                        "of a step / step time / peak"
                        % (args.roofline_steps, (what % (k_file, csrc_sha16())) if k_traffic is not None else
                           "null — no per-kernel PMC entry under profiles/ for these kernel sources (csrc_sha16 %s)"
                           % csrc_sha16()),
                "class": {"name": "conv_%s (implicit-GEMM MFMA)" % name, "achieved": round(ach, 2),
                          "frac": round(ach / peak, 4), "traffic": traffic, "launches": r["launches"],
                          "avg_launch_ms": round(r["ms"] / max(1, r["launches"]), 4),
                          "algorithmic_bytes": round(r["bytes"] / max(1, r["launches"])),
                          "traffic_source": (what % (traffic_file, csrc_sha16())) if traffic is not None else None},
                "kernels": [{"kernel": kn, "ms_per_step": round(v["ms"] / args.roofline_steps, 3),
                             "launches_per_step": round(v["launches"] / args.roofline_steps, 1),
                             "avg_launch_ms": round(v["ms"] / max(1, v["launches"]), 4),
                             "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1),
                             "gbs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 0)} for kn, v in ktable[:12]],
                "per_layer": layer_roofline(cls_layers, args.precision, args.per_layer_top or None),
                "classes": {k: {"tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2), "ms": round(v["ms"], 2),
                                "launches": v["launches"]} for k, v in rep.items()}}
    if world > 1:
        dist.barrier()

    # ---- the frozen networks' share, part by part, and Bench A (N = 1 only)
    frozen_rec = bench_a = None
    if rank == 0 and world == 1 and args.roofline_steps > 0:
        frozen_rec = frozen_split(args, tr, device)
    if rank == 0 and world == 1 and args.bench_a_steps > 0 and args.gae != 1:
        bench_a = bench_a_record(args, device)

    # ---- the fp32 parity mode on the same workload (N = 1 only): images/s and the class fraction of the fp32 MFMA peak
    fp32_rec = None
    if rank == 0 and world == 1 and args.fp32_steps > 0 and args.precision != "fp32":
        fp32_rec = fp32_record(args, device, tr)
        tr = None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        gf = algorithmic_gflop_per_image(args.gae, args.image_size, args.network_capacity, args.fmap_max)
        line = {
            "metric": "StylEx G+D+enc train-step images/sec @%dpx" % args.image_size, "value": round(value, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic (torch.rand 256x256 batches resident in HBM, random-init weights, seeded "
                    "random-weight ResNet-18 classifier and LPIPS-AlexNet)",
            "config": {"workload": "FFHQ-shaped %dx%d StylEx, batch %d/GPU, GAE=%d (noise+encoder micro-steps), GP every "
                                   "4th step, %s classifier" % (args.image_size, args.image_size, args.batch, args.gae,
                                                                "ResNet-18" if args.classifier == "resnet" else "MobileNetV2"),
                       "image_size": args.image_size, "batch_per_gpu": args.batch, "start_step": args.start_step,
                       "pl_every": args.pl_every if args.start_step > 5000 else None,
                       "gradient_accumulate_every": args.gae, "global_batch": world * args.batch,
                       "parallelism": "dp%d" % world,
                       "host_cores": len(os.sched_getaffinity(0)),
                       "rng": ("latents / noise planes drawn on the GPU generator (as every rank of the data-parallel path; "
                               "--device-rng 0 = the reference's CPU draws + upload: -3 %, profiles/r06_d_ab_imagegrad_devrng.txt)")
                              if args.device_rng else "CPU generator draws + upload (reference order)",
                       "frozen_nets": ("classifier forward: stock MIOpen fp32, immediate mode (north_star; reference cli.py:38); in the bf16 "
                                       "mode its data gradient and LPIPS-AlexNet's layers 2-5 run on this library's bf16 kernels "
                                       "(frozen_resnet._ResNetBodyHybrid, lpips_alex._taps_bf16; STYLEX_FROZEN_BWD_BF16=0 / "
                                       "STYLEX_LPIPS_BF16=0: all on MIOpen fp32), outside the roofline classes")
                                      if not args.miopen_find else "stock MIOpen fp32, algorithms searched during the warm-up"},
            "algorithmic_conv_gflop_per_image": round(gf, 1),
            "step_conv_tflops": round(gf * value / 1e3, 2),
            "roofline": roof, "cpu_baseline": cpu, "fp32_parity_mode": fp32_rec, "bench_a": bench_a,
            "frozen_nets": frozen_rec,
        }
        if roof is not None:
            roof["whole_step_frac"] = round(gf * value / 1e3 / PEAK_TFLOPS[args.precision], 4)
    else:
        line = None
    # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio (seen after the line when
    # stdout is a pipe) — tear the process group down and flush C stdio first
    def flush_c_stdio():
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass

    if dist.is_initialized():
        flush_c_stdio()
        if world > 1:
            dist.barrier()  # every rank has flushed before rank 0 prints
        dist.destroy_process_group()
    flush_c_stdio()
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
