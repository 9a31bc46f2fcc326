"""1-rank RCCL check of the gradient exchange (run ON the GPU box; tests/test_hip_parity.py runs it as a subprocess):

    python tools/ddp_overlap_identity.py [steps]

Builds the same seeded Trainer three times inside ONE 1-rank RCCL process group — no process group semantics
(is_ddp=False), GradSync with the collectives after the backward (default), GradSync with the in-backward bucket launch
(overlap=True, 32 MB buckets from autograd hooks) — runs `steps` train() calls each and prints ONE JSON line with the
parameter checksums.  With one rank the all-reduce is the identity (AVG over one rank), so all three must end with
BIT-IDENTICAL parameters: it proves that the flat-bucket packing, the view binding of .grad, the hook-driven launch order
and the no-gradient handling do not change a single bit of the update — on the real RCCL code path (the gloo tests
cover the multi-rank arithmetic on the CPU)."""
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


def run(mode, steps, dev):
    import stylex_train as st
    from lpips_standin import LPIPSStandIn
    from standins import TinyClassifier

    size, cap, fmax, bs, gae = 64, 8, 128, 4, 2
    gd = torch.Generator().manual_seed(7)
    batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
    torch.manual_seed(42)
    np.random.seed(42)
    random.seed(42)
    if mode == "overlap":
        os.environ["STYLEX_DDP_OVERLAP"] = "1"
    else:
        os.environ["STYLEX_DDP_OVERLAP"] = "0"
    tr = st.Trainer(name="ddp_" + mode, base_dir="/tmp/stylex_ddp_identity", image_size=size, network_capacity=cap,
                    fmap_max=fmax, batch_size=bs, gradient_accumulate_every=gae, lr=2e-4, ttur_mult=1.5, rec_scaling=1,
                    kl_scaling=1, classifier=TinyClassifier(seed=99).to(dev), lpips_fn=LPIPSStandIn(seed=4242).to(dev),
                    classifier_name="resnet", evaluate_every=10 ** 9, save_every=10 ** 9, device=dev,
                    is_ddp=mode != "single", rank=0, world_size=1, device_rng=False)
    tr.loader = st.cycle(batches)
    tr.dataset = list(range(1000))
    tr.save = tr.evaluate = lambda *a, **k: None
    tr.init_StylEx()
    if mode != "single":
        assert tr._d_sync.overlap == (mode == "overlap") and tr._g_sync.overlap == (mode == "overlap")
    torch.manual_seed(43)
    np.random.seed(43)
    random.seed(43)
    scal = []
    for _ in range(steps):
        tr.train()
        scal.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss])
    torch.cuda.synchronize()
    sums = [float(p.detach().double().sum()) for p in tr.StylEx.parameters()]
    absum = [float(p.detach().double().abs().sum()) for p in tr.StylEx.parameters()]
    return {"scalars": scal, "sum": sums, "abs": absum}


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    import hip_backend as hb
    import ops

    hb.load_library()
    ops.set_precision(os.environ.get("STYLEX_PRECISION", "bf16"))
    torch.backends.cudnn.deterministic = True
    os.environ["STYLEX_MIOPEN_BENCHMARK"] = "0"
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        run("single", 1, dev)  # warm-up Trainer: the first Trainer of a process orders its backward differently
        out = {m: run(m, steps, dev) for m in ("single", "post", "overlap")}
    finally:
        dist.destroy_process_group()
    ident = {m: out[m]["sum"] == out["single"]["sum"] and out[m]["abs"] == out["single"]["abs"]
             and out[m]["scalars"] == out["single"]["scalars"] for m in ("post", "overlap")}
    print(json.dumps({"steps": steps, "identical_to_single": ident, "n_params": len(out["single"]["sum"]),
                      "scalars_single": out["single"]["scalars"], "scalars_overlap": out["overlap"]["scalars"]}), flush=True)


if __name__ == "__main__":
    main()
