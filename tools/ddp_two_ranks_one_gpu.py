"""Two data-parallel ranks of the HIP Trainer on ONE MI355X, exchanging gradients over gloo (run ON the GPU box;
tests/test_hip_parity.py starts it once per rank):

    RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 python tools/ddp_two_ranks_one_gpu.py [calls] &
    RANK=1 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 python tools/ddp_two_ranks_one_gpu.py [calls]

No box with two GPUs is reachable from this environment and RCCL refuses two ranks on one device, so this is the closest
the N > 1 path gets to hardware: BOTH processes run the real HIP kernels on cuda:0 at config 2's network sizes (256 px,
capacity 16, 512 channels: ~400 MB of gradients in 32 MB buckets) under the real multi-stream backward, and the buckets
go through a real two-rank all-reduce (gloo stages device tensors through the host; the averaging is sum + divide
instead of ReduceOp.AVG, nothing else differs from the RCCL path: same hooks, same launch order, same stream joins).

Each rank sees different images and draws.  Checked, per exchange mode (collectives after the backward /
in-backward bucket launch from the autograd hooks):
  * the replicas end every train() call with BIT-IDENTICAL trained networks — G, D, the style network, the encoder
    (MAX == MIN over the ranks of the flattened parameter vector; the moving-average copies GE / SE are maintained on
    the main rank only, as in the reference, stylex_train.py:1475-1479, and are left out);
  * the in-backward launch ends with the same bits as the post-backward exchange on every rank — a bucket reduced before
    one of its gradients was complete (a missing stream join) would differ;
  * the first-use self-check of the in-backward path ran and passed (GradSync._selfcheck), several buckets each;
  * the exchange really mixed the ranks: a rank's parameters differ from what it gets alone (is_ddp=False).
Rank 0 prints ONE JSON line."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
for p in (os.path.join(PKG, "stylex"), PKG, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def trained(tr):
    """The parameters the optimisers step (GE / SE are rank 0's moving-average copies)."""
    return [p for n, p in tr.StylEx.named_parameters() if not n.startswith(("GE.", "SE."))]


def run(mode, calls, dev, rank, world, size, cap, fmax):
    import stylex_train as st
    from lpips_standin import LPIPSStandIn
    from standins import TinyClassifier

    per_rank, gae = int(os.environ.get("TWO_BATCH", "2")), 2
    gd = torch.Generator().manual_seed(7 + rank)  # every rank its own images
    batches = [torch.rand(per_rank, 3, size, size, generator=gd) for _ in range(8)]
    torch.manual_seed(42)  # same initial weights whatever the mode (DDP broadcasts rank 0's anyway)
    np.random.seed(42)
    random.seed(42)
    os.environ["STYLEX_DDP_OVERLAP"] = "1" if mode == "overlap" else "0"
    ddp = mode != "alone"
    # bench.py's Trainer (config 2 / 3): the seeded random-weight ResNet-18 classifier and LPIPS-AlexNet the package builds
    # itself; TWO_STANDINS=1: the oracle's small stand-ins instead (their ATen resize / pooling backward passes add
    # with atomics: run-to-run noise once two processes share the GPU, which hides what this tool looks for)
    extra = {}
    if os.environ.get("TWO_STANDINS") == "1":
        extra = dict(classifier=TinyClassifier(seed=99).to(dev), lpips_fn=LPIPSStandIn(seed=4242).to(dev))
    tr = st.Trainer(name="two_" + mode, base_dir="/tmp/stylex_two_ranks_%d" % rank, image_size=size, network_capacity=cap,
                    fmap_max=fmax, batch_size=per_rank * (world if ddp else 1), gradient_accumulate_every=gae, lr=2e-4,
                    ttur_mult=1.5, mixed_prob=0.9, rec_scaling=1, kl_scaling=1, aug_prob=0., alternating_training=True,
                    classifier_name="resnet", classifier_path=None, evaluate_every=10 ** 9, save_every=10 ** 9,
                    tensorboard_dir=None, device=dev, is_ddp=ddp, rank=rank if ddp else 0, world_size=world if ddp else 1,
                    device_rng=os.environ.get("TWO_DEVICE_RNG", "1") == "1", **extra)  # the same (device) draws with and without the exchange
    tr.loader = st.cycle(batches)
    tr.dataset = list(range(1000))
    tr.save = tr.evaluate = lambda *a, **k: None
    tr.init_StylEx()
    rec = {"buckets": None, "selfcheck": None, "pre": [], "post": []}
    if ddp and os.environ.get("TWO_DIAG") == "1":  # the local gradients entering each exchange and the averaged buckets leaving it
        import parallel

        def wrap(sync):
            inner = sync.all_reduce

            def traced():
                torch.cuda.synchronize()
                rec["pre"].append([None if p.grad is None else p.grad.detach().clone() for p in sync.params])
                inner()
                torch.cuda.synchronize()
                rec["post"].append([f.clone() for f in sync.flats])
            sync.all_reduce = traced
        wrap(tr._d_sync)
        wrap(tr._g_sync)
        name_of = {id(p): n for n, p in tr.StylEx.named_parameters()}
        rec["names"] = [[name_of[id(p)] for p in tr._d_sync.params], [name_of[id(p)] for p in tr._g_sync.params]]
    if ddp:
        assert tr._d_sync.overlap == (mode == "overlap") and tr._g_sync.overlap == (mode == "overlap")
        rec["buckets"] = [len(tr._d_sync.buckets), len(tr._g_sync.buckets)]
        left0 = (tr._d_sync._selfcheck_left, tr._g_sync._selfcheck_left)
    torch.manual_seed(43 + rank)
    np.random.seed(43 + rank)
    random.seed(43 + rank)
    same, scal, snaps = [], [], []
    for _ in range(calls):
        tr.train()
        if os.environ.get("TWO_DIAG") == "1":
            snaps.append({n: p.detach().clone() for n, p in tr.StylEx.named_parameters() if not n.startswith(("GE.", "SE."))})
        scal.append([float(tr.d_loss), float(tr.g_loss), float(tr.total_rec_loss), float(tr.total_kl_loss)])
        if ddp:  # replicas bit-identical after EVERY call
            flat = torch.cat([p.detach().reshape(-1) for p in trained(tr)])
            hi, lo = flat.clone(), flat.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            same.append(bool(torch.equal(hi, lo)))
    torch.cuda.synchronize()
    if ddp:
        rec["selfcheck"] = {"ran": [left0[0] - tr._d_sync._selfcheck_left, left0[1] - tr._g_sync._selfcheck_left],
                            "still_overlapped": [bool(tr._d_sync.overlap), bool(tr._g_sync.overlap)]}
    rec["replicas_identical"] = same
    rec["scalars"] = scal
    rec["snaps"] = snaps
    rec["params"] = torch.cat([p.detach().reshape(-1) for p in trained(tr)]).clone()
    return rec


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    size, cap, fmax = (int(os.environ.get(k, d)) for k, d in (("TWO_SIZE", 256), ("TWO_CAP", 16), ("TWO_FMAX", 512)))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)  # both ranks on the one GPU of the box
    dev = torch.device("cuda:0")
    import hip_backend as hb
    import ops

    hb.load_library()
    ops.set_precision(os.environ.get("STYLEX_PRECISION", "bf16"))
    torch.backends.cudnn.deterministic = True  # the stand-in classifier / LPIPS run library convolutions
    os.environ["STYLEX_MIOPEN_BENCHMARK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        run("alone", 1, dev, rank, world, size, cap, fmax)  # warm-up Trainer: the first Trainer of a process orders its backward differently
        out = {m: run(m, calls, dev, rank, world, size, cap, fmax) for m in ("alone", "post", "overlap")}
        mine = {
            "rank": rank,
            "replicas_identical": {m: out[m]["replicas_identical"] for m in ("post", "overlap")},
            "overlap_equals_post": bool(torch.equal(out["overlap"]["params"], out["post"]["params"]))
            and out["overlap"]["scalars"] == out["post"]["scalars"],
            "exchange_changed_the_update": not bool(torch.equal(out["post"]["params"], out["alone"]["params"])),
            "selfcheck": out["overlap"]["selfcheck"],
            "buckets": {m: out[m]["buckets"] for m in ("post", "overlap")},
            "finite": bool(np.isfinite(np.array(out["overlap"]["scalars"])).all()),
            "scalars_overlap": out["overlap"]["scalars"],
            "scalars_post": out["post"]["scalars"],
            "max_abs_overlap_minus_post": float((out["overlap"]["params"] - out["post"]["params"]).abs().max()),
            "n_param_values": int(out["post"]["params"].numel()),
        }
        if os.environ.get("TWO_DIAG") == "1":  # which exchange is not reproducible, from which call on, in which parameters
            again = {m: run(m, calls, dev, rank, world, size, cap, fmax) for m in ("alone", "post", "overlap")}
            third = {m: run(m, calls, dev, rank, world, size, cap, fmax) for m in ("post",)}
            diag = {}
            for a, b, tag in ((out["alone"], again["alone"], "alone_vs_alone"), (out["post"], again["post"], "post_vs_post2"),
                              (out["post"], third["post"], "post_vs_post3"), (again["post"], third["post"], "post2_vs_post3"),
                              (out["overlap"], again["overlap"], "overlap_vs_overlap2"), (out["post"], out["overlap"], "post_vs_overlap"),
                              (again["post"], again["overlap"], "post2_vs_overlap2")):
                per_call = []
                for sa, sb in zip(a["snaps"], b["snaps"]):
                    bad = [(n, float((sa[n] - sb[n]).abs().max())) for n in sa if not torch.equal(sa[n], sb[n])]
                    per_call.append({"n_bad": len(bad), "of": len(sa), "first": bad[:3]})
                diag[tag] = per_call
                if a["pre"] and b["pre"]:  # exchange by exchange (D, G, D, G ...): local gradients in, averaged buckets out
                    ex = []
                    for k in range(min(len(a["pre"]), len(b["pre"]))):
                        pre_bad = sum(1 for x, y in zip(a["pre"][k], b["pre"][k]) if (x is None) != (y is None) or (x is not None and not torch.equal(x, y)))
                        post_bad = sum(1 for x, y in zip(a["post"][k], b["post"][k]) if not torch.equal(x, y))
                        ex.append([pre_bad, len(a["pre"][k]), post_bad, len(a["post"][k])])
                        if pre_bad and tag.startswith("post") and "overlap" not in tag and (tag + "/first_pre") not in diag:
                            det = []
                            for i, (x, y) in enumerate(zip(a["pre"][k], b["pre"][k])):
                                if x is not None and y is not None and not torch.equal(x, y):
                                    det.append([i, list(x.shape), float((x - y).abs().max()), float(x.abs().max())])
                            names = a["names"][k % 2]
                            diag[tag + "/first_pre"] = {"exchange": k, "params[index, shape, maxdiff, maxabs]": det[:10] + det[-4:],
                                                        "same": [names[i] for i, (x, y) in enumerate(zip(a["pre"][k], b["pre"][k]))
                                                                 if x is not None and y is not None and torch.equal(x, y)],
                                                        "differ": [names[i] for i, *_ in det], "none": [names[i] for i, x in enumerate(a["pre"][k]) if x is None]}
                    diag[tag + "/exchanges[pre_bad, n, post_bad, n_buckets]"] = ex
            mine["diag"] = diag
        both = [None] * world
        dist.all_gather_object(both, mine)
    finally:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"calls": calls, "world": world, "size": size, "ranks": both}), flush=True)


if __name__ == "__main__":
    main()
