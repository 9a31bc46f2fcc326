"""GPU probe: which part of the train step survives HIP-graph capture?  Each stage runs in its own process
(python -X faulthandler) so that a hard crash of one stage is reported and the others still run.

    python tools/graph_stage_probe.py            # all stages
    python tools/graph_stage_probe.py <stage>    # one stage in this process
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
for p in (os.path.join(PKG, "stylex"), PKG, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

STAGES = ["step"]


def capture(fn, warm=3, mode="thread_local"):
    import torch

    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=mode):
        out = fn()
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    return out


def small_model(size=32):
    import torch

    import ops
    import stylex_train as st

    ops.set_precision("bf16")
    torch.manual_seed(0)
    return st.StylEx(size, network_capacity=8, fmap_max=64, rank=0, capturable=True)


def trainer(size=32, **kw):
    import random

    import numpy as np
    import torch

    import ops
    import stylex_train as st

    ops.set_precision("bf16")
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(7)
    ring = [torch.rand(4, 3, size, size, generator=gen).to(dev) for _ in range(8)]
    for s in (torch.manual_seed, np.random.seed, random.seed):
        s(42)
    tr = st.Trainer(name="gsp", base_dir="/tmp/stylex_gsp", image_size=size, network_capacity=8, fmap_max=64,
                    batch_size=4, gradient_accumulate_every=2, lr=2e-4, ttur_mult=1.5, classifier_name="resnet",
                    classifier_path=None, evaluate_every=10 ** 9, save_every=10 ** 9, tensorboard_dir=None, device=dev,
                    graphs=True, graph_warmup=4, **kw)
    tr.loader = st.cycle(ring)
    tr.dataset = list(range(10 ** 6))
    tr.save = lambda *a, **k: None
    tr.evaluate = lambda *a, **k: None
    tr.init_StylEx()
    return tr


def run_stage(name):
    import torch

    import hip_backend as hb
    import ops
    import stylex_train as st

    dev = torch.device("cuda:0")
    ops.set_precision("bf16")
    if name in ("conv", "conv_bwd"):
        x = torch.randn(4, 64, 32, 32, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
        x.requires_grad_(name == "conv_bwd")

        def fn():
            hb.pack_cache_clear()
            ops.set_fast(True)
            y = ops.conv2d(x, w, None, 1, 1, lrelu=True)
            if name == "conv_bwd":
                w.grad = None
                y.float().sum().backward()
            return y

        capture(fn)
    elif name == "sidestream":
        side = torch.cuda.Stream()
        a = torch.randn(1024, 1024, device=dev)

        def fn():
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = a @ a
            c = a + 1
            main.wait_stream(side)
            b.record_stream(main)
            return b + c

        capture(fn)
    elif name in ("g_fwd", "d_fwdbwd", "adam"):
        m = small_model()
        w = torch.randn(4, m.G.num_layers, 514, device=dev)
        n = torch.rand(4, 32, 32, 1, device=dev)
        x = torch.rand(4, 3, 32, 32, device=dev)

        def g_fwd():
            hb.pack_cache_clear()
            with torch.no_grad():
                return m.G(w, n)

        def d_fwdbwd():
            hb.pack_cache_clear()
            ops.set_fast(True)
            m.D_opt.zero_grad()
            m.D(x).mean().backward()
            if name == "adam":
                m.D_opt.step()

        capture(g_fwd if name == "g_fwd" else d_fwdbwd)
    elif name in ("classifier", "lpips"):
        x = torch.rand(4, 3, 64, 64, device=dev, requires_grad=True)
        y = torch.rand(4, 3, 64, 64, device=dev)
        if name == "classifier":
            from resnet_classifier import ResNet

            net = ResNet(None, 0, image_size=64)
            fn0 = lambda: net.classify_images(x).sum()  # noqa: E731
        else:
            lp = st.get_lpips(dev)
            fn0 = lambda: st.perceptual_loss(y, x, lp)  # noqa: E731

        def fn():
            x.grad = None
            fn0().backward()

        capture(fn)
    elif name == "fork":
        tr = trainer()
        m = tr.StylEx
        x = torch.rand(4, 3, 32, 32, device=dev)

        def fn():
            hb.pack_cache_clear()
            with torch.no_grad():
                a, b = tr._fork([lambda: m.encoder(x), lambda: tr._classify(x)])
            return a.sum() + b.sum()

        capture(fn)
    elif name in ("g_fwdbwd", "g_d_bwd"):
        m = small_model()
        w = torch.randn(4, m.G.num_layers, 514, device=dev, requires_grad=True)
        n = torch.rand(4, 32, 32, 1, device=dev)

        def fn():
            hb.pack_cache_clear()
            ops.set_fast(True)
            m.G_opt.zero_grad()
            img = m.G(w, n)
            if name == "g_d_bwd":
                st.set_requires_grad(m.D, False)
                out = m.D(img).mean()
                st.set_requires_grad(m.D, True)
            else:
                out = img.mean()
            out.backward()
            ops.set_fast(False)

        capture(fn)
    elif name in ("d_phase", "g_phase", "d_phase_gp", "g_phase_nostreams", "g_phase_gae1", "g_phase_noside"):
        if name == "g_phase_nostreams":
            os.environ["STYLEX_STREAMS"] = "0"
        tr = trainer()
        if name == "g_phase_noside":  # Trainer-level forks stay, the per-block side stream inside D/encoder goes
            import networks

            networks._side_stream = lambda t, which=0: None
        if name == "g_phase_gae1":
            tr.gradient_accumulate_every = 1
        gae = tr.gradient_accumulate_every
        grp = list(range(gae))
        for _ in range(3):
            tr.graphs = False
            tr.train()
        st_ = {"encoder_input": False, "latents_fn": None}
        reals, micro_d = tr._draw_d(grp, st_, True)
        st_["encoder_input"] = False
        micro_g, _ = tr._draw_g(grp, st_, True, False)

        def fn():
            hb.pack_cache_clear()
            acc = tr._new_acc()
            if name.startswith("d_phase"):
                tr._d_phase([grp], [(reals, micro_d)], name == "d_phase_gp", gae, True, acc)
            else:
                tr._g_phase([grp], [(micro_g, [])], False, gae, True, acc)
            return tr._loss_stack(acc)

        capture(fn, warm=1)
    elif name in ("step", "step_nostreams"):
        if name == "step_nostreams":
            os.environ["STYLEX_STREAMS"] = "0"
        tr = trainer()
        for i in range(16):
            tr.train()
            print("step", i, "captured", sorted(tr._graph_cache), flush=True)
        print("losses", tr.d_loss, tr.g_loss, flush=True)
    print("STAGE_OK", name, flush=True)


def main():
    if len(sys.argv) > 1:
        run_stage(sys.argv[1])
        return
    variants = [("", {})]
    if os.environ.get("PROBE_VARIANTS") == "1":
        variants = [("default", {}), ("nopool", {"STYLEX_GRAPH_POOL": "0"}), ("global", {"STYLEX_GRAPH_MODE": "global"}),
                    ("relaxed", {"STYLEX_GRAPH_MODE": "relaxed"}), ("nodblock", {"STYLEX_DBLOCK": "0"}),
                    ("nostreams", {"STYLEX_STREAMS": "0"})]
    for s, (vname, venv) in [(s, v) for s in STAGES for v in variants]:
        env = dict(os.environ, STYLEX_GRAPH_DEBUG="1", **venv)
        p = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), s], capture_output=True,
                           text=True, timeout=600, env=env)
        s = s + ":" + vname
        ok = "STAGE_OK" in p.stdout
        print("=== %-12s rc=%d %s" % (s, p.returncode, "OK" if ok else "FAILED"), flush=True)
        if not ok:
            print("\n".join((p.stdout + "\n" + p.stderr).splitlines()[-45:]), flush=True)


if __name__ == "__main__":
    main()
