"""N>1 path on CPU: two processes over gloo.  Checks (a) GradSync averages bucketed gradients and
handles missing grads, (b) weights are broadcast from rank 0, (c) a 2-rank Trainer step with the
global batch split across ranks produces the same D/G parameter update as... each other (replicas
stay bit-identical), and the all-reduced gradient equals the mean of the per-rank gradients.
The op surface is the CPU test double (no GPU here); the collective code is exactly what runs over
RCCL on the GPU box."""
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ops
        import parallel
        import stylex_train as st
        from cpu_ops import CpuOracleOps
        from lpips_standin import LPIPSStandIn
        from standins import TinyClassifier

        ops.use_impl(CpuOracleOps)
        # (a) GradSync
        torch.manual_seed(100 + rank)
        ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2))]
        ps[0].grad = torch.full((5, 3), float(rank + 1))
        ps[1].grad = torch.arange(7.0) * (rank + 1)
        # ps[2].grad stays None on purpose
        parallel.GradSync(ps, bucket_bytes=40).all_reduce()
        mean = (1 + world) / 2.0
        assert torch.allclose(ps[0].grad, torch.full((5, 3), mean))
        assert torch.allclose(ps[1].grad, torch.arange(7.0) * mean)
        assert torch.equal(ps[2].grad, torch.zeros(2, 2))
        # (b)+(c) Trainer under DDP
        size, bs = 32, 4  # global batch 4 -> 2 per rank
        gd = torch.Generator().manual_seed(7 + rank)
        batches = [torch.rand(bs // world, 3, size, size, generator=gd) for _ in range(8)]
        torch.manual_seed(1000 + rank)  # different seeds: weights must still agree after broadcast
        tr = st.Trainer(name="r%d" % rank, base_dir=tmp, image_size=size, network_capacity=4, fmap_max=64,
                        batch_size=bs, gradient_accumulate_every=2, lr=2e-4, ttur_mult=1.5, rec_scaling=1,
                        kl_scaling=1, classifier=TinyClassifier(seed=99), lpips_fn=LPIPSStandIn(seed=4242),
                        classifier_name="resnet", evaluate_every=10 ** 9, save_every=10 ** 9, is_ddp=True, rank=rank,
                        world_size=world, device=torch.device("cpu"))
        tr.loader = st.cycle(batches)
        tr.save = lambda *a, **k: None
        tr.evaluate = lambda *a, **k: None
        tr.init_StylEx()
        # the trained networks; the moving-average copies GE / SE are rank 0's alone (reference stylex_train.py:1475-1479:
        # `if self.is_main and ...: EMA() / reset_parameter_averaging()`), so they differ across ranks from step 2 on
        trained = lambda: [p for n, p in tr.StylEx.named_parameters() if not n.startswith(("GE.", "SE."))]  # noqa: E731
        w0 = torch.cat([p.detach().reshape(-1) for p in trained()])
        random.seed(5 + rank)
        np.random.seed(5 + rank)
        torch.manual_seed(5 + rank)
        for _ in range(4):  # step 2 resets the moving average on the main rank, step 4 is the next gradient-penalty call
            tr.train()
        w1 = torch.cat([p.detach().reshape(-1) for p in trained()])
        gather0 = [torch.zeros_like(w0) for _ in range(world)]
        gather1 = [torch.zeros_like(w1) for _ in range(world)]
        dist.all_gather(gather0, w0)
        dist.all_gather(gather1, w1)
        ok_init = all(torch.equal(gather0[0], g) for g in gather0)
        ok_after = all(torch.equal(gather1[0], g) for g in gather1)
        moved = float((w1 - w0).abs().max())
        q.put((rank, ok_init, ok_after, moved, tr.d_loss, tr.g_loss))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo(tmp_path):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_init, ok_after, moved, d_loss, g_loss in res:
        assert ok_init, "weights differ across ranks after broadcast"
        assert ok_after, "replicas diverged after four all-reduced steps"
        assert moved > 0
        assert np.isfinite(d_loss) and np.isfinite(g_loss)


def _gradsync_worker(rank, world, port, q):
    import torch.nn as nn

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import parallel

        torch.manual_seed(0)
        net = nn.Sequential(nn.Linear(40, 300), nn.ReLU(), nn.Linear(300, 300), nn.ReLU(), nn.Linear(300, 7))
        unused = nn.Linear(5, 5)  # a parameter set that never receives a gradient (contributes zeros)
        params = list(unused.parameters()) + list(net.parameters())  # unused lands in the LAST bucket
        sync = parallel.GradSync(params, bucket_bytes=2 * 1024, overlap=True)  # several buckets
        assert len(sync.buckets) >= 3, len(sync.buckets)
        x = torch.randn(16, 40, generator=torch.Generator().manual_seed(100 + rank))
        # reference: local gradients, averaged by hand
        net(x).square().mean().backward()
        local = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered) / world
        results = {}
        for mode in ("armed", "plain"):
            for p in params:
                p.grad = None
            if mode == "armed":
                sync.arm()
                launched_before = 0
            net(x).square().mean().backward()
            if mode == "armed":
                launched_before = sync._next  # buckets already in flight when the backward returned
            sync.all_reduce()
            # a parameter without a gradient on any rank contributes zeros to the collective and is handed back with
            # .grad = None (as on one GPU: Adam skips it instead of stepping on a zero gradient)
            assert all(p.grad is None for p in unused.parameters())
            got = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
            results[mode] = (float((got - want).abs().max()), launched_before if mode == "armed" else None)
            if mode == "armed":
                results["selfcheck_ran"] = (sync._selfcheck_left, sync.overlap)  # one of the two first-use checks is spent
        # round 6: the first-use self-check of the in-backward path — ranks that disagree on an averaged bucket fall back
        import warnings

        if rank == 1:
            sync.flats[0][0] += 1.0  # what a bucket reduced before its gradients were complete would look like
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            ok = sync._selfcheck()
        results["selfcheck_mismatch"] = (ok, sync.overlap, len(wl))
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradsync_overlapped_matches_manual_average():
    """GradSync launches buckets from inside the backward (hooks, strict index order) and yields exactly the
    rank-average; a never-used parameter's bucket is flushed with zeros by all_reduce()."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gradsync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, results in res:
        assert results["armed"][0] < 1e-6 and results["plain"][0] < 1e-6, results
        assert results["armed"][1] >= 2, "no bucket was launched during the backward: %r" % (results,)
        assert results["selfcheck_ran"] == (1, True), results  # ran once, passed, the overlap path stays on
        assert results["selfcheck_mismatch"] == (False, False, 1), results  # EVERY rank sees the mismatch, falls back, warns once


def _uneven_worker(rank, world, port, q):
    import torch.nn as nn

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import parallel

        torch.manual_seed(0)
        heads = [nn.Linear(30, 50) for _ in range(6)]  # six independent sub-graphs -> six buckets
        params = [p for h in heads for p in h.parameters()]
        sync = parallel.GradSync(params, bucket_bytes=50 * 30 * 4, overlap=True)
        assert len(sync.buckets) == 6
        x = torch.randn(8, 30, generator=torch.Generator().manual_seed(10 + rank))
        order = [(i + 2 * rank) % 6 for i in range(6)]  # every rank finishes its buckets in a different order
        if rank % 2:
            order.reverse()
        # manual average for reference
        for h in heads:
            h(x).square().mean().backward()
        local = torch.cat([p.grad.reshape(-1).clone() for p in params])
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered) / world
        sync.zero_grad()
        launch_log = []
        orig = sync._launch
        sync._launch = lambda bi: (launch_log.append(bi), orig(bi))[1]
        sync.arm()
        for i in order:
            heads[i](x).square().mean().backward()
        in_backward = len(launch_log)
        sync.all_reduce()
        got = torch.cat([p.grad.reshape(-1) for p in params])
        views_ok = all(p.grad.data_ptr() == sync._views[id(p)].data_ptr() for p in params)
        q.put((rank, float((got - want).abs().max()), launch_log, in_backward, views_ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradsync_world4_uneven_bucket_readiness():
    """Four ranks whose gradients become ready in four different orders: the collectives still go out strictly in
    bucket-index order on every rank (no cross-rank mismatch / deadlock), the result is the rank average, and the
    gradients stay views of the persistent flat buckets (no pack / scatter copies)."""
    world = 4
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, log, in_backward, views_ok in res:
        assert err < 1e-6, (rank, err)
        assert log == list(range(6)), (rank, log)
        assert views_ok
    assert any(r[3] > 0 for r in res)


def _nan_worker(rank, world, port, tmp, q, on_save_step=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ops
        import stylex_train as st
        from cpu_ops import CpuOracleOps
        from lpips_standin import LPIPSStandIn
        from standins import TinyClassifier

        ops.use_impl(CpuOracleOps)
        size = 16
        gd = torch.Generator().manual_seed(7 + rank)
        batches = [torch.rand(2, 3, size, size, generator=gd) for _ in range(8)]
        tr = st.Trainer(name="n%d" % rank, base_dir=tmp, image_size=size, network_capacity=2, fmap_max=16, batch_size=4,
                        gradient_accumulate_every=1, classifier=TinyClassifier(seed=99), lpips_fn=LPIPSStandIn(seed=4242),
                        classifier_name="resnet", evaluate_every=10 ** 9, save_every=2 if on_save_step else 10 ** 9,
                        is_ddp=True, rank=rank, world_size=world, device=torch.device("cpu"))
        tr.loader = st.cycle(batches)
        tr.save = lambda *a, **k: None
        tr.evaluate = lambda *a, **k: None
        loads = []

        def fake_load(num=-1):  # the real load() ends in a parameter broadcast: a collective every rank must enter
            loads.append((tr.steps, num))
            import parallel

            parallel.broadcast_parameters(tr.StylEx.D)

        tr.load = fake_load
        tr.init_StylEx()
        real_stack = tr._loss_stack
        events = []
        for call in range(4):
            # ONLY rank 1 sees a NaN loss: on its second call, or (on_save_step) on the call whose step number is a
            # multiple of save_every — there the scalars are resolved inside the same call, before the checkpoint
            if call == (2 if on_save_step else 1) and rank == 1:
                tr._loss_stack = lambda acc: real_stack(acc) * float("nan")
            else:
                tr._loss_stack = real_stack
            try:
                tr.train()
                events.append("ok")
            except st.NanException:
                events.append("nan")
        q.put((rank, events, loads))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("on_save_step", [False, True])
def test_nan_restart_is_collective(tmp_path, on_save_step):
    """A NaN loss on ONE rank makes EVERY rank take the reload-and-raise path at the same train() call (the flag is
    MAX-reduced on the device inside the step that produced it), so the collectives of load()/the next step line up.
    on_save_step: the NaN falls on a checkpoint step, where the scalars are resolved at the end of the SAME call — on
    every rank, not only on rank 0 (round-2 hole: rank 0 entered load()'s broadcast while the others went on into the
    next discriminator all-reduce)."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nan_worker, args=(r, world, port, str(tmp_path), q, on_save_step)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ev0, loads0), (r1, ev1, loads1) = res
    # detected while enqueueing the call after the NaN one / inside the NaN call itself on a checkpoint step
    assert ev0 == ev1 == ["ok", "ok", "nan", "ok"], (ev0, ev1)
    assert loads0 == loads1 and len(loads0) == 1


@pytest.mark.timeout(900)
def test_bench_py_two_ranks_over_gloo(tmp_path):
    """bench.py's own N > 1 path, launched the way the driver launches it (`python -m torch.distributed.run
    --nproc-per-node 2 ... bench.py --gpus 2 --steps K --warmup W`), on the CPU test double over gloo: process-group
    set-up, rank-split batches, the in-backward bucket all-reduce (the default), barrier + MAX-reduce of the elapsed
    time, exactly ONE JSON line (rank 0, last thing on stdout), process-group teardown.  The first real `--gpus 8` run
    must not die in this plumbing."""
    import json
    import subprocess

    launcher = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bench_gloo_double.py")
    env = dict(os.environ, OMP_NUM_THREADS="2", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    env.pop("STYLEX_DDP_OVERLAP", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), launcher, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--image-size", "64", "--network-capacity", "2", "--fmap-max", "16", "--roofline-steps", "0", "--fp32-steps", "0",
           "--bench-a-steps", "0", "--no-cpu-baseline", "--workdir", str(tmp_path)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    assert out.stdout.rstrip().endswith(lines[0]), "the JSON line must be the last thing on stdout"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["parallelism"] == "dp2" and j["config"]["global_batch"] == 4
    assert j["value"] > 0 and abs(j["value"] - 2 * 2 * 2 * 2 / (j["ms_per_step"] * 2 / 1e3)) / j["value"] < 0.02
    assert j["algorithmic_conv_gflop_per_image"] < 10  # size-aware: this 64 px / capacity-2 model, not the 256 px constants
    assert j["roofline"] is None and j["cpu_baseline"] is None
