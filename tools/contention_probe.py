"""Which kernel's result depends on what ELSE the GPU is doing?  (run ON the GPU box)

    python tools/contention_probe.py [g|d] [repeats] [batch] [size]

Two identically seeded Trainers are bit-identical when a process has the GPU to itself (tests/test_hip_determinism_gpu.py),
but two processes sharing the device showed run-to-run differences in the generator phase (round 6, found by
tools/ddp_two_ranks_one_gpu.py) — with one HIP stream, with serialised launches, with either BLAS: not a missing stream
dependency.  This probe repeats ONE forward + backward of the generator (or discriminator) on fixed inputs while a
second process keeps the device busy, records every block's activations and gradients, and names the first tensor — in
execution order — that is not bit-identical to the first repeat."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
for p in (os.path.join(PKG, "stylex"), PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HAMMER = r"""
import torch, time
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device="cuda")
t0 = time.time()
while time.time() - t0 < %d:
    for _ in range(20):
        c = a @ a
        d = b * 1.0001 + 1.0
        e = torch.nn.functional.conv2d(torch.randn(16, 64, 128, 128, device="cuda"), torch.randn(64, 64, 3, 3, device="cuda"), padding=1)
    torch.cuda.synchronize()
"""


def train_mode(repeats, batch, size, log, rec, watch):
    """`repeats` identically seeded Trainers, ONE train() call each (after a throw-away Trainer), with every generator /
    discriminator / encoder block's output and output gradient recorded in execution order."""
    import argparse

    sys.path.insert(0, ROOT)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    torch.backends.cudnn.deterministic = True
    a = argparse.Namespace(batch=batch, image_size=size, gae=2, classifier="resnet", workdir="/tmp/sb_probe", precision="bf16",
                           device_rng=int(os.environ.get("PROBE_DEVICE_RNG", "1")))

    def instrument(tr):
        m = tr.StylEx
        for li, blk in enumerate(m.G.blocks):
            fm, trf = blk.forward_main, blk.to_rgb.forward

            def fm2(x, istyle, inoise, styles=None, fm=fm, li=li):
                x, sc = fm(x, istyle, inoise, styles) if styles is not None else fm(x, istyle, inoise)
                return watch("G.x%d" % li, x), sc

            def tr2(x, prev, istyle, style=None, padded=False, trf=trf, li=li):
                return watch("G.rgb%d" % li, trf(x, prev, istyle, style=style, padded=padded))

            blk.forward_main, blk.to_rgb.forward = fm2, tr2
        for name, net in (("D", m.D), ("E", m.encoder)):
            for li, blk in enumerate(net.blocks):
                f = blk.forward
                blk.forward = lambda x, f=f, li=li, name=name: watch("%s.b%d" % (name, li), f(x))

    bench.seed_all(42)
    tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
    tr.train()
    torch.cuda.synchronize()
    del tr
    first, bad_runs = None, 0
    for r in range(repeats):
        bench.seed_all(42)
        tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
        instrument(tr)
        log.clear()
        for c in range(int(os.environ.get("PROBE_CALLS", "1"))):
            rec("---- call %d" % c, torch.zeros(1))
            tr.train()
        torch.cuda.synchronize()
        cur = list(log) + [("grad " + n, p.grad.detach().clone()) for n, p in tr.StylEx.named_parameters() if p.grad is not None]
        del tr
        torch.cuda.empty_cache()
        if first is None:
            first = cur
            print("recorded %d tensors per call" % len(cur), flush=True)
            continue
        if [t for t, _ in cur] != [t for t, _ in first]:
            print("repeat %d: a different SEQUENCE of recorded tensors (%d vs %d)" % (r, len(cur), len(first)))
            for i, ((ta, _), (tb, _)) in enumerate(zip(first, cur)):
                if ta != tb:
                    print("    first difference at position %d: %s vs %s" % (i, ta, tb))
                    break
            bad_runs += 1
            continue
        bad = [(i, tag, float((x.float() - y.float()).abs().max()), float(x.float().abs().max()))
               for i, ((tag, x), (_, y)) in enumerate(zip(first, cur)) if x.shape != y.shape or not torch.equal(x, y)]
        if bad:
            bad_runs += 1
            print("repeat %d: %d of %d tensors differ; in execution order the first are:" % (r, len(bad), len(cur)))
            for i, tag, d, mx in bad[:10]:
                print("    #%-4d %-34s maxdiff %.4g (max |.| %.4g)   [before it: %s]" % (i, tag, d, mx, first[i - 1][0] if i else "-"))
        else:
            print("repeat %d: bit-identical" % r)
        sys.stdout.flush()
    print("SUMMARY train: %d of %d repeats differ from the first" % (bad_runs, repeats - 1))


def calls_mode(repeats, batch, size):
    """Every public hip_backend call of `PROBE_CALLS` train() calls, logged in call order with checksums of its tensor
    arguments and results (kept on the device until the end: no host synchronisation per call); `repeats` identically
    seeded Trainers; reports the first call whose RESULT differs from the first Trainer's while its ARGUMENTS agree."""
    import argparse
    import inspect

    sys.path.insert(0, ROOT)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    import hip_backend as hb

    torch.backends.cudnn.deterministic = True
    keep_names = set(filter(None, os.environ.get("PROBE_KEEP", "torgb_bwd").split(",")))  # calls whose tensors are kept whole
    kept = []
    skip = {"load_library", "is_cl", "act_dtype", "conv_shape", "empty_cl", "torgb_ok", "pack_cache_clear", "timing_enable",
            "timing_report", "timing_layers", "to_cl", "timing_pause", "timing_paused", "timing_kernels", "mark_updated"}
    log = []

    def tensors(o):
        if isinstance(o, torch.Tensor):
            return [o] if o.is_cuda and o.numel() else []
        if isinstance(o, (tuple, list)):
            return [t for x in o for t in tensors(x)]
        if isinstance(o, dict):
            return [t for x in o.values() for t in tensors(x)]
        return []

    def cs(ts):
        if not ts:
            return None
        return torch.stack([torch.stack((t.detach().double().sum(), t.detach().double().abs().sum())) for t in ts])

    def wrap(name, fn):
        def inner(*a, **k):
            ins = tensors(a) + tensors(k)
            cin = cs(ins)
            out = fn(*a, **k)
            log.append((name, [tuple(t.shape) for t in ins][:4], cin, cs(tensors(out)), cs(ins)))  # (last: the arguments AFTER the call)
            if name in keep_names:
                kept.append((len(log) - 1, [t.detach().clone() for t in tensors(out)], [t.detach().clone() for t in ins]))
            return out

        return inner

    for name, fn in list(vars(hb).items()):
        if inspect.isfunction(fn) and fn.__module__ == hb.__name__ and not name.startswith("_") and name not in skip:
            setattr(hb, name, wrap(name, fn))
    a = argparse.Namespace(batch=batch, image_size=size, gae=2, classifier="resnet", workdir="/tmp/sb_probe", precision="bf16",
                           device_rng=int(os.environ.get("PROBE_DEVICE_RNG", "1")))
    bench.seed_all(42)
    tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
    tr.train()
    torch.cuda.synchronize()
    del tr
    first, bad_runs = None, 0
    host = lambda t: None if t is None else t.cpu().tolist()  # noqa: E731
    for r in range(repeats):
        bench.seed_all(42)
        tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
        log.clear()
        for _c in range(int(os.environ.get("PROBE_CALLS", "2"))):
            tr.train()
        torch.cuda.synchronize()
        cur = [(n, sh, host(ci), host(co), host(ca)) for n, sh, ci, co, ca in log]
        cur_kept = list(kept)
        kept.clear()
        del tr
        torch.cuda.empty_cache()
        if first is None:
            first, first_kept = cur, cur_kept
            print("recorded %d hip_backend calls per Trainer" % len(cur), flush=True)
            continue
        if [c[0] for c in cur] != [c[0] for c in first]:
            print("repeat %d: a different call SEQUENCE (%d vs %d calls)" % (r, len(cur), len(first)))
            for i, (x, y) in enumerate(zip(first, cur)):
                if x[0] != y[0]:
                    print("    first difference at call %d: %s %s vs %s %s" % (i, x[0], x[1], y[0], y[1]))
                    break
            bad_runs += 1
            continue
        shown = 0
        for i, (x, y) in enumerate(zip(first, cur)):
            if x[2:] != y[2:]:
                same_in = x[2] == y[2]
                print("repeat %d: call %d %s %s: arguments %s, results %s, arguments after the call %s" % (
                    r, i, x[0], x[1], "SAME" if same_in else "differ", "SAME" if x[3] == y[3] else "DIFFER",
                    "SAME" if x[4] == y[4] else "DIFFER"))
                if shown == 0:
                    for j in range(max(0, i - 5), i):
                        print("      before: %d %s %s" % (j, first[j][0], first[j][1]))
                    print("      A in %s\n      B in %s\n      A out %s\n      B out %s" % (x[2], y[2], x[3], y[3]))
                shown += 1
                if shown >= 4:
                    break
        for (ia, oa, ina), (ib, ob, inb) in zip(first_kept, cur_kept):
            if ia != ib:
                break
            for k, (ta, tb) in enumerate(list(zip(ina, inb)) + list(zip(oa, ob))):
                if ta.shape == tb.shape and not torch.equal(ta, tb):
                    d = (ta.float() - tb.float()).reshape(-1)
                    idx = d.ne(0).nonzero().reshape(-1)
                    what = "argument %d" % k if k < len(ina) else "result %d" % (k - len(ina))
                    print("   kept call %d %s %s shape %s strides %s: %d elements differ; flat storage-order positions %s ... %s" % (
                        ia, first[ia][0], what, tuple(ta.shape), tuple(ta.stride()), idx.numel(), idx[:12].tolist(), idx[-4:].tolist()))
                    flat_a = ta.permute(0, 2, 3, 1).reshape(-1) if ta.dim() == 4 else ta.reshape(-1)
                    flat_b = tb.permute(0, 2, 3, 1).reshape(-1) if tb.dim() == 4 else tb.reshape(-1)
                    j = (flat_a.float() - flat_b.float()).ne(0).nonzero().reshape(-1)
                    print("      NHWC-order positions %s ... %s; A %s B %s" % (j[:12].tolist(), j[-4:].tolist(),
                          flat_a[j[:8]].float().tolist(), flat_b[j[:8]].float().tolist()))
                    if first[ia][0] == "torgb_bwd" and k >= len(ina):  # which of the two is right?  (definition, fp32)
                        x_, gy_, s1_, w_ = ina[0], ina[1], ina[2].float(), ina[3].float().reshape(3, -1)
                        m_ = w_[None] * s1_[:, None, :]  # [B, 3, C]
                        ref = torch.einsum("bnhw,bnc->bchw", gy_[:, :3].float(), m_)
                        ea = (ta.float() - ref).abs()
                        eb = (tb.float() - ref).abs()
                        sel = (ta.float() - tb.float()).ne(0)
                        print("      against the definition, at the differing elements: first Trainer max err %.4g, this Trainer max err %.4g "
                              "(elsewhere %.4g)" % (float(ea[sel].max()), float(eb[sel].max()), float(ea[~sel].max())))
                        pos = sel.nonzero()
                        bs = sorted(set(pos[:, 0].tolist()))
                        print("      images %s; channels %s; rows %s; cols %s" % (bs, sorted(set(pos[:, 1].tolist()))[:40],
                              sorted(set(pos[:, 2].tolist()))[:40], sorted(set(pos[:, 3].tolist()))[:40]))
                        wrong = tb if float(eb[sel].max()) > float(ea[sel].max()) else ta
                        q = pos[0].tolist()
                        print("      first differing element %s: definition %.6g, first %.6g, this %.6g; gy there %s; wrong value's neighbours along C %s" % (
                            q, float(ref[tuple(q)]), float(ta[tuple(q)]), float(tb[tuple(q)]), gy_[q[0], :, q[2], q[3]].float().tolist(),
                            wrong[q[0], max(0, q[1] - 2):q[1] + 3, q[2], q[3]].float().tolist()))
                        # where could the wrong value have come from?  the same formula with another pixel's gy / another channel's products
                        H, W = ref.shape[2], ref.shape[3]
                        for q in pos[:6].tolist() + pos[-2:].tolist():
                            bq, cq, yq, xq = q
                            wv = float(wrong[bq, cq, yq, xq])
                            cands = []
                            for dy in range(-2, 3):
                                for dx in range(-33, 34):
                                    yy, xx = yq + dy, xq + dx
                                    if 0 <= yy < H and 0 <= xx < W:
                                        v = (gy_[bq, :3, yy, xx].float()[:, None] * m_[bq]).sum(0).to(torch.bfloat16).float()  # [C]
                                        hit = (v == wv).nonzero().reshape(-1).tolist()
                                        cands += [(dy, dx, c2) for c2 in hit]
                            t2 = [float(gy_[bq, n, yq, xq]) * float(m_[bq, n, cq]) for n in range(3)]
                            print("      element %s: right %.6g wrong %.6g; terms %s; (dy, dx, channel) whose formula gives the wrong value: %s" % (
                                q, float(ref[bq, cq, yq, xq]), wv, ["%.4g" % t for t in t2], cands[:8]))
                    shown += 1
                    break
            else:
                continue
            break
        if shown:
            bad_runs += 1
        else:
            print("repeat %d: all %d calls bit-identical (by checksum)" % (r, len(cur)))
        sys.stdout.flush()
    print("SUMMARY calls: %d of %d repeats differ from the first" % (bad_runs, repeats - 1))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "g"
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    size = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    import hip_backend as hb
    import networks
    import ops

    hb.load_library()
    ops.set_precision(os.environ.get("STYLEX_PRECISION", "bf16"))
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    hammer = None
    if os.environ.get("PROBE_HAMMER", "1") == "1":
        hammer = subprocess.Popen([sys.executable, "-c", HAMMER % int(os.environ.get("PROBE_HAMMER_S", "240"))])
    try:
        log = []  # (tag, tensor) in execution order of the current repeat

        def rec(tag, t):
            log.append((tag, t.detach().clone()))

        def watch(tag, t):
            rec("fwd " + tag, t)
            if t.requires_grad:
                t.register_hook(lambda g, tag=tag: rec("bwd d(" + tag + ")", g))
            return t

        if which == "calls":
            return calls_mode(repeats, batch, size)
        if which == "train":
            return train_mode(repeats, batch, size, log, rec, watch)
        if which == "g":
            net = networks.Generator(size, 514, network_capacity=16, fmap_max=512).to(dev)
            for li, blk in enumerate(net.blocks):
                fm, tr = blk.forward_main, blk.to_rgb.forward

                def fm2(x, istyle, inoise, styles=None, fm=fm, li=li):
                    x, sc = fm(x, istyle, inoise, styles) if styles is not None else fm(x, istyle, inoise)
                    return watch("x%d" % li, x), sc

                def tr2(x, prev, istyle, style=None, padded=False, tr=tr, li=li):
                    return watch("rgb%d" % li, tr(x, prev, istyle, style=style, padded=padded))

                blk.forward_main, blk.to_rgb.forward = fm2, tr2
            w = torch.randn(batch, net.num_layers, 514, device=dev, requires_grad=True)
            noise = torch.rand(batch, size, size, 1, device=dev)
            cot = torch.randn(batch, 3, size, size, device=dev)
            run = lambda: net(w, noise)  # noqa: E731
            leaves = [("w", w)]
        else:
            net = networks.DiscriminatorE(size, network_capacity=16, fmap_max=512).to(dev)
            for li, blk in enumerate(net.blocks):
                f = blk.forward
                blk.forward = lambda x, f=f, li=li: watch("d%d" % li, f(x))
            x = torch.rand(batch, 3, size, size, device=dev, requires_grad=True)
            out0 = net(x)
            out0 = out0[0] if isinstance(out0, tuple) else out0
            cot = torch.randn_like(out0)
            run = lambda: net(x)  # noqa: E731
            leaves = [("x", x)]
        first = None
        bad_runs = 0
        for r in range(repeats):
            log.clear()
            for p in net.parameters():
                p.grad = None
            for _, t in leaves:
                t.grad = None
            out = run()
            out = out[0] if isinstance(out, tuple) else out
            rec("fwd out", out)
            out.backward(cot)
            torch.cuda.synchronize()
            cur = list(log) + [("grad " + n, t.grad.detach().clone()) for n, t in leaves] + [
                ("grad " + n, p.grad.detach().clone()) for n, p in net.named_parameters() if p.grad is not None]
            if first is None:
                first = cur
                print("recorded %d tensors per repeat: %s ..." % (len(cur), [t for t, _ in cur[:6]]), flush=True)
                continue
            assert [t for t, _ in cur] == [t for t, _ in first]
            bad = [(tag, float((a.float() - b.float()).abs().max()), float(a.float().abs().max()))
                   for (tag, a), (_, b) in zip(first, cur) if not torch.equal(a, b)]
            if bad:
                bad_runs += 1
                print("repeat %d: %d of %d tensors differ; in execution order the first are:" % (r, len(bad), len(cur)))
                for tag, d, m in bad[:8]:
                    print("    %-40s maxdiff %.4g (max |.| %.4g)" % (tag, d, m))
            else:
                print("repeat %d: bit-identical" % r)
            sys.stdout.flush()
        print("SUMMARY %s: %d of %d repeats differ from the first" % (which, bad_runs, repeats - 1))
    finally:
        if hammer is not None:
            hammer.kill()
            hammer.wait()


if __name__ == "__main__":
    main()
