"""Host logic of the product package on CPU: module wiring, init order, state-dict layout,
losses, Trainer control flow — against the vectors captured from the reference.  The HIP
kernels cannot run here, so the product's op surface is replaced by the oracle's CPU test
double (oracle/cpu_ops.py) for the duration of this file; the numerical parity of the kernels
themselves is the job of the ``-m gpu`` tests."""
import os
import random

import numpy as np
import pytest
import torch

import ops
import stylex_train as st
from cpu_ops import CpuOracleOps
from lpips_standin import LPIPSStandIn
from standins import TinyClassifier
from conftest import load_golden
from test_oracle_vs_golden import assert_same_stats, close, close_stats, stats, build_nets_model


@pytest.fixture(autouse=True)
def cpu_double():
    prev = ops.use_impl(CpuOracleOps)
    yield
    ops.use_impl(prev)


@pytest.mark.parametrize("size", [8, 16, 32])
def test_init_parity_product(size):
    g = load_golden("init_%d" % size)
    s, cap, fmax = (int(v) for v in g["config"])
    torch.manual_seed(int(g["seed"]))
    m = st.StylEx(s, network_capacity=cap, fmap_max=fmax)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    for i, (k, v) in enumerate(sd.items()):
        assert ",".join(map(str, v.shape)) == str(g["shapes"][i]), k
        assert_same_stats(g["stats"][i], stats(v), k)


@pytest.mark.parametrize("size", [16, 32])
def test_network_parity_product(size):
    g = load_golden("nets_%d" % size)
    m = build_nets_model(g, cls=st.StylEx)
    w, inoise, x = (torch.from_numpy(g[n]) for n in ("w", "inoise", "x"))
    rgb, coords = m.G(w, inoise, get_style_coords=True)
    close(g["rgb"], rgb)
    close(g["coords"], coords)
    close(g["d_out"], m.D(x))
    close(g["enc_out"], m.encoder(x))
    close(g["s_out"], m.S(w[:, 0]))


def test_losses_product():
    g = load_golden("losses")
    s, cap, fmax = (int(v) for v in g["config"])
    torch.manual_seed(int(g["seed"]))
    m = st.StylEx(s, network_capacity=cap, fmap_max=fmax)
    real, fake = torch.from_numpy(g["hinge/real"]), torch.from_numpy(g["hinge/fake"])
    close(g["hinge/d"], st.hinge_loss(real, fake), 1e-6)
    close(g["hinge/g"], st.gen_hinge_loss(fake, None), 1e-6)
    x = torch.from_numpy(g["gp/x"]).requires_grad_()
    gp = st.gradient_penalty(x, m.D(x))
    close(g["gp/value"], gp)
    m.D.zero_grad()
    gp.backward()
    close(g["gp/grad_fc_w"], m.D.fc.weight.grad, 1e-4)
    w = torch.from_numpy(g["pl/w"]).requires_grad_()
    img = m.G(w, torch.from_numpy(g["pl/inoise"]))
    torch.manual_seed(int(g["pl/noise_seed"]))
    close(g["pl/lengths"], st.calc_pl_lengths(w, img))
    close(g["kl/value"], st.classifier_kl_loss(torch.from_numpy(g["kl/real"]), torch.from_numpy(g["kl/fake"])), 1e-6)
    lp = LPIPSStandIn(seed=int(g["rec/lpips_seed"]))
    i1, i2, w1, w2 = (torch.from_numpy(g["rec/" + n]) for n in ("i1", "i2", "w1", "w2"))
    close(g["rec/value"], st.reconstruction_loss(i1, i2, w2, w1, lp), 1e-6)
    # the product's own LPIPS implements the same published graph as the stand-in
    mine = st.LPIPS(seed=int(g["rec/lpips_seed"]))
    close(g["rec/lpips_value"], mine(st.lpips_normalize(i1), st.lpips_normalize(i2)).reshape(-1), 1e-6)


def make_trainer(g, tmp_path, device=None, trainer_cls=None):
    size, cap, fmax, bs, gae, alt, n, start = (int(v) for v in g["config"])
    cls = TinyClassifier(seed=int(g["cls_seed"]))
    gd = torch.Generator().manual_seed(int(g["data_seed"]))
    batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
    lp = LPIPSStandIn(seed=int(g["lpips_seed"]))
    if device is not None:
        cls.to(device)
        lp = lp.to(device)
    seed = int(g["seed"])
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    aug = float(g["aug_prob"]) if "aug_prob" in g.files else 0.
    lr = float(g["lr"]) if "lr" in g.files else 2e-4
    tr = (trainer_cls or st.Trainer)(name="t", base_dir=str(tmp_path), image_size=size, network_capacity=cap, fmap_max=fmax,
                                     batch_size=bs, gradient_accumulate_every=gae, alternating_training=bool(alt), lr=lr,
                                     ttur_mult=1.5, rec_scaling=1, kl_scaling=1, classifier=cls, lpips_fn=lp,
                                     classifier_name="resnet", evaluate_every=10 ** 9, save_every=10 ** 9, device=device,
                                     aug_prob=aug)
    tr.loader = st.cycle(batches)
    tr.dataset = list(range(1000))
    tr.save = lambda *a, **k: None
    tr.evaluate = lambda *a, **k: None
    tr.init_StylEx()
    tr.steps = start
    pl0 = float(g["pl_mean0"]) if "pl_mean0" in g.files else float("nan")
    tr.pl_mean = None if np.isnan(pl0) else pl0
    return tr, n


def make_cfg4_trainer(g, tmp_path, device=None):
    """BASELINE config 4 in miniature: Trainer(classifier_name='mobilenet') loading a MobileNetV2 checkpoint from
    ./trained_classifiers (seeded stand-in weights, oracle/ref_shim.py), path-length + R1 step."""
    import os

    from standins import seeded_mobilenet_state

    size, cap, fmax, bs, gae, alt, n, start = (int(v) for v in g["config"])
    os.makedirs(os.path.join(str(tmp_path), "trained_classifiers"), exist_ok=True)
    torch.save(seeded_mobilenet_state(int(g["cls_seed"])), os.path.join(str(tmp_path), "trained_classifiers", "mnv2.pth"))
    gd = torch.Generator().manual_seed(int(g["data_seed"]))
    batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
    lp = LPIPSStandIn(seed=int(g["lpips_seed"]))
    if device is not None:
        lp = lp.to(device)
    seed = int(g["seed"])
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        tr = st.Trainer(name="t", base_dir=str(tmp_path), image_size=size, network_capacity=cap, fmap_max=fmax,
                        batch_size=bs, gradient_accumulate_every=gae, alternating_training=bool(alt), lr=2e-4,
                        ttur_mult=1.5, rec_scaling=1, kl_scaling=1, lpips_fn=lp, classifier_name="mobilenet",
                        classifier_path="mnv2.pth", evaluate_every=10 ** 9, save_every=10 ** 9, device=device,
                        rank=0)
    finally:
        os.chdir(cwd)
    assert isinstance(tr.classifier, st.MobileNet)
    torch.manual_seed(seed)  # the fixture re-seeds after the classifier is built (see gen_cfg4)
    np.random.seed(seed)
    random.seed(seed)
    if device is not None:
        assert next(tr.classifier.model.parameters()).device.type == device.type
    tr.loader = st.cycle(batches)
    tr.dataset = list(range(1000))
    tr.save = lambda *a, **k: None
    tr.evaluate = lambda *a, **k: None
    tr.init_StylEx()
    tr.steps = start
    tr.pl_mean = float(g["pl_mean0"])
    return tr, n, batches


def assert_param_stats(tr, g, tol=2e-3, head_atol=1e-4):
    """post-step parameter checksums of the golden: after N Adam steps an element may legitimately differ by a
    fraction of lr (2e-4)"""
    params = dict(tr.StylEx.named_parameters())
    for name, gs in zip(g["param_names"], g["param_stats"]):
        close_stats(gs, params[str(name)].detach().cpu(), tol, head_atol=head_atol)


def run_steps(tr, n):
    rows = []
    for _ in range(n):
        tr.train()
        rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss,
                     np.nan if tr.last_gp_loss is None else tr.last_gp_loss,
                     np.nan if tr.pl_mean is None else tr.pl_mean])
    return np.array(rows, dtype=np.float64)


@pytest.mark.parametrize("tag", ["gae1_alt", "gae2_alt", "gae2_noalt", "gae2_pl", "gae2_aug"])
def test_trainer_step_parity_cpu(tag, tmp_path):
    g = load_golden("steps_" + tag)
    tr, n = make_trainer(g, tmp_path)
    rows = run_steps(tr, n)
    gold = g["scalars"]
    np.testing.assert_allclose(rows[0], gold[0], rtol=5e-5, atol=5e-6, equal_nan=True)
    np.testing.assert_allclose(rows, gold, rtol=1e-3, atol=1e-3, equal_nan=True)
    assert_param_stats(tr, g)


def test_config4_mobilenet_pl_step_parity_cpu(tmp_path):
    """BASELINE config 4 (MobileNetV2 classifier, R1 + path-length step) against the reference's own Trainer +
    MobileNet wrapper (tests/golden/steps_cfg4.npz, oracle/make_golden.py::gen_cfg4)."""
    g = load_golden("steps_cfg4")
    tr, n, batches = make_cfg4_trainer(g, tmp_path)
    close(g["logits_batch0"], tr.classifier.classify_images(batches[0]), 1e-5)
    rows = run_steps(tr, n)
    gold = g["scalars"]
    np.testing.assert_allclose(rows[0], gold[0], rtol=5e-5, atol=5e-6, equal_nan=True)
    np.testing.assert_allclose(rows, gold, rtol=1e-3, atol=1e-3, equal_nan=True)
    assert_param_stats(tr, g)


def check_evalsurface(tmp_path, device=None, tol=1e-3, head_atol=1e-4):
    """N3 — EMA / reset_parameter_averaging, truncate_style, generate_truncated and evaluate() of the product Trainer
    against tests/golden/evalsurface_16.npz, captured from the reference Trainer itself (oracle/make_golden.py::
    gen_evalsurface; reference stylex_train.py:985-999, :1508-1575, :1624-1656).  Same seeds -> same draws: the
    2000-sample W mean, the truncated styles and the three image grids evaluate() writes (regular / EMA / mixing
    regularities; with and without encoder input) must match."""
    g = load_golden("evalsurface_16")
    size, cap, fmax, bs, gae, tiles = (int(v) for v in g["config"])
    cls = TinyClassifier(seed=int(g["cls_seed"]))
    gd = torch.Generator().manual_seed(int(g["data_seed"]))
    batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
    lp = LPIPSStandIn(seed=int(g["lpips_seed"]))
    if device is not None:
        cls.to(device)
        lp = lp.to(device)
    seed = int(g["seed"])
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    tr = st.Trainer(name="ev", base_dir=str(tmp_path), image_size=size, network_capacity=cap, fmap_max=fmax, batch_size=bs,
                    gradient_accumulate_every=gae, lr=2e-4, ttur_mult=1.5, rec_scaling=1, kl_scaling=1, classifier=cls,
                    lpips_fn=lp, classifier_name="resnet", evaluate_every=10 ** 9, save_every=10 ** 9, device=device,
                    num_image_tiles=tiles)
    tr.loader = st.cycle(batches)
    tr.dataset = list(range(1000))
    tr.save = lambda *a, **k: None
    real_evaluate = tr.evaluate
    tr.evaluate = lambda *a, **k: None
    tr.init_StylEx()
    tr.steps = int(g["start_step"])
    rows = []
    for _ in range(len(g["scalars"])):  # three calls across step 20010: the EMA update fires inside the second one
        tr.train()
        rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss])
    np.testing.assert_allclose(np.array(rows, dtype=np.float64), g["scalars"], rtol=tol, atol=tol)
    assert_param_stats(tr, g, head_atol=head_atol)  # includes SE.* / GE.*: the averaged copies after update_moving_average
    params = dict(tr.StylEx.named_parameters())
    moved = max(float((params["GE." + n[2:]] - p).detach().abs().max()) for n, p in params.items() if n.startswith("G."))
    assert moved > 0, "the averaged generator must differ from the live one after training steps"
    # truncate_style: seeded 2000-sample mean of S(z), then trunc_psi * (w - mean) + mean
    torch.manual_seed(123)
    np.random.seed(123)
    random.seed(123)
    tr.av = None
    t_in = torch.randn(4, tr.StylEx.G.latent_dim)
    np.testing.assert_array_equal(t_in.numpy(), g["trunc_in"])
    dev = tr.device
    t_out = tr.truncate_style(t_in.to(dev), trunc_psi=float(g["trunc_psi"]))
    close(g["trunc_av"], torch.as_tensor(tr.av), tol)  # W mean of truncate_style
    close(g["trunc_out"], t_out.detach().cpu(), tol)
    # evaluate(): capture what the product hands to its image writer
    grids = []
    orig = st.save_image_grid
    st.save_image_grid = lambda imgs, path, nrow=8: grids.append((os.path.basename(str(path)), imgs.detach().float().cpu(), nrow))
    try:
        for k, (sd, enc) in enumerate(zip((int(v) for v in g["eval_seeds"]), (False, True))):
            torch.manual_seed(sd)
            np.random.seed(sd)
            random.seed(sd)
            tr.av = None
            real_evaluate(encoder_input=enc, num=k)
    finally:
        st.save_image_grid = orig
    assert [n for n, _, _ in grids] == [str(n) for n in g["grid_names"]]
    assert [r for _, _, r in grids] == [int(r) for r in g["grid_nrow"]]
    for i, (name, imgs, _) in enumerate(grids):
        close(g["grid/%d" % i], imgs, 5 * tol)  # evaluate grid `name`
    # reset_parameter_averaging (:997-999)
    tr.StylEx.reset_parameter_averaging()
    params = dict(tr.StylEx.named_parameters())
    for name, gs in zip(g["param_names"], g["param_stats_after_reset"]):
        close_stats(gs, params[str(name)].detach().cpu(), 2e-3, head_atol=head_atol)
    for n, p in params.items():
        if n.startswith("G."):
            assert torch.equal(params["GE." + n[2:]], p)


def test_eval_ema_truncation_surface_vs_reference_golden_cpu(tmp_path):
    check_evalsurface(tmp_path)


def test_lpips_state_dict_loading_and_published_formula():
    """LPIPS.load_lpips_state_dict takes the key layout of the lpips package (net.slice<k>.<idx>.weight / .bias,
    lin<k>.model.1.weight — reference: lpips.LPIPS(net='alex'), stylex_train.py:404) and the forward is the published
    metric: unit-normalised AlexNet taps, squared difference, non-negative 1x1 weights, spatial mean, summed."""
    import torch.nn.functional as F

    from lpips_alex import _ALEX, LPIPS

    m = LPIPS(seed=1)
    g = torch.Generator().manual_seed(9)
    sd = {}
    for i, (ci, co, k, s, p, mp, idx) in enumerate(_ALEX):
        sd["net.slice%d.%d.weight" % (i + 1, idx)] = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
        sd["net.slice%d.%d.bias" % (i + 1, idx)] = torch.randn(co, generator=g) * 0.1
        sd["lin%d.model.1.weight" % i] = torch.rand(1, co, 1, 1, generator=g) / co
    before = [p.clone() for p in m.cw]
    m.load_lpips_state_dict(sd)
    for i, (_, _, _, _, _, _, idx) in enumerate(_ALEX):
        assert torch.equal(m.cw[i], sd["net.slice%d.%d.weight" % (i + 1, idx)]) and not torch.equal(m.cw[i], before[i])
        assert torch.equal(m.cb[i], sd["net.slice%d.%d.bias" % (i + 1, idx)])
        assert torch.equal(m.lin[i], sd["lin%d.model.1.weight" % i])
    a, b = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1, torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    got = m(a, b)
    # the published computation, written out independently of the module's forward
    shift, scale = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1), torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)
    fa, fb, want = (a - shift) / scale, (b - shift) / scale, 0
    for i, (ci, co, k, s, p, mp, idx) in enumerate(_ALEX):
        if mp:
            fa, fb = F.max_pool2d(fa, 3, 2), F.max_pool2d(fb, 3, 2)
        w_, b_ = sd["net.slice%d.%d.weight" % (i + 1, idx)], sd["net.slice%d.%d.bias" % (i + 1, idx)]
        fa, fb = F.relu(F.conv2d(fa, w_, b_, stride=s, padding=p)), F.relu(F.conv2d(fb, w_, b_, stride=s, padding=p))
        na = fa / (fa.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        nb = fb / (fb.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        want = want + ((na - nb) ** 2 * sd["lin%d.model.1.weight" % i]).sum(1, keepdim=True).mean(dim=(2, 3), keepdim=True)
    assert got.shape == (2, 1, 1, 1)
    close(want, got, 1e-5)
    assert float(m(a, a).abs().max()) == 0.0 and float(got.min()) > 0.0


def test_missing_classifier_checkpoint_raises(tmp_path, monkeypatch):
    """A named classifier checkpoint that does not exist is an error (as in the reference, where torch.load raises),
    never a silent random-weight classifier; None is the explicit opt-in used by the synthetic benchmarks."""
    monkeypatch.chdir(tmp_path)
    for name in ("resnet", "mobilenet"):
        with pytest.raises(FileNotFoundError):
            st.Trainer(name="x", base_dir=str(tmp_path), image_size=32, classifier_name=name,
                       classifier_path="does-not-exist.pth", device=torch.device("cpu"))
    st.Trainer(name="x", base_dir=str(tmp_path), image_size=32, classifier_name="mobilenet", classifier_path=None,
               device=torch.device("cpu"))


def test_a20_helpers_exported():
    """reference :269-293: raise_if_nan, gradient_accumulate_contexts, loss_backwards keep their names/semantics."""
    with pytest.raises(st.NanException):
        st.raise_if_nan(torch.tensor(float("nan")))
    st.raise_if_nan(torch.tensor(1.0))
    entered = []

    class FakeDDP:
        def no_sync(self):
            from contextlib import contextmanager

            @contextmanager
            def cm():
                entered.append(1)
                yield

            return cm()

    assert sum(1 for _ in st.gradient_accumulate_contexts(3, True, [FakeDDP(), FakeDDP()])) == 3
    assert len(entered) == 4  # no_sync on every micro-step but the last, for both wrappers
    assert sum(1 for _ in st.gradient_accumulate_contexts(3, False, [FakeDDP()])) == 3 and len(entered) == 4
    w = torch.ones(2, requires_grad=True)
    st.loss_backwards(False, (w * 3).sum(), None, 0)
    assert torch.equal(w.grad, torch.full((2,), 3.0))


def test_checkpoint_roundtrip(tmp_path):
    g = load_golden("steps_gae1_alt")
    tr, _ = make_trainer(g, tmp_path)
    del tr.save
    tr.save_every = 1
    st.Trainer.save(tr, 0)
    ck = torch.load(tr.model_name(0))
    assert set(ck.keys()) == {"StylEx", "version"} and ck["version"] == "1.8.7"
    gi = load_golden("init_32")
    assert list(ck["StylEx"].keys()) == [str(k) for k in gi["keys"]]
    tr2, _ = make_trainer(g, tmp_path)
    tr2.load(0)
    for (k, a), (_, b) in zip(tr.StylEx.state_dict().items(), tr2.StylEx.state_dict().items()):
        assert torch.equal(a, b), k


def test_checkpoint_with_training_state(tmp_path):
    """Extension: with save_training_state=True the checkpoint also carries the optimiser moments,
    the step count and the path-length mean, and a new Trainer restores them; the file stays a valid reference
    checkpoint (the reference's loader reads 'StylEx' and 'version' only)."""
    g = load_golden("steps_gae1_alt")
    tr, _ = make_trainer(g, tmp_path)
    del tr.save
    tr.save_training_state = True
    tr.save_every = 10 ** 9
    for _ in range(2):
        tr.train()
    tr.pl_mean = np.float32(0.25)  # what np.mean() hands back: must not end up in the pickle (weights_only load)
    st.Trainer.save(tr, 7)
    ck = torch.load(tr.model_name(7), weights_only=True)
    assert type(ck["training_state"]["pl_mean"]) is float
    assert {"StylEx", "version", "training_state"} == set(ck.keys())
    tr2, _ = make_trainer(g, tmp_path)
    tr2.save_training_state = True
    tr2.save_every = 10 ** 9
    tr2.load(7)
    assert tr2.steps == 2 and tr2.pl_mean == 0.25
    for name in ("G_opt", "D_opt"):
        a = getattr(tr.StylEx, name).state_dict()["state"]
        b = getattr(tr2.StylEx, name).state_dict()["state"]
        assert len(a) == len(b) > 0
        for k in a:
            assert torch.equal(a[k]["exp_avg"], b[k]["exp_avg"]) and torch.equal(a[k]["exp_avg_sq"], b[k]["exp_avg_sq"])
            assert float(a[k]["step"]) == float(b[k]["step"]) == 2.0
    # without the flag the same file loads as a plain reference checkpoint
    tr3, _ = make_trainer(g, tmp_path)
    tr3.load(7)
    assert not tr3.StylEx.G_opt.state_dict()["state"]


def test_product_fails_loudly_without_gpu():
    """No CPU fallback in the product path: HIP ops refuse CPU tensors."""
    ops.use_impl(ops.HipOps)
    import hip_backend

    with pytest.raises(hip_backend.StylexHipError):
        ops.conv2d(torch.zeros(1, 4, 4, 4), torch.zeros(4, 4, 3, 3), None, 1, 1)


def check_newarch_init(device=None):
    """reference stylex_train_new.py: parameter names / shapes / init stream, conditional D, W with probabilities, the
    encoder's own lr group."""
    import stylex_train_new as stn

    g = load_golden("newarch_init")
    s, cap, fmax = (int(v) for v in g["config"])
    torch.manual_seed(int(g["seed"]))
    m = stn.StylEx(s, network_capacity=cap, fmap_max=fmax, rank=0 if device is not None else None)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    for i, (k, v) in enumerate(sd.items()):
        assert ",".join(map(str, v.shape)) == str(g["shapes"][i]), k
        assert_same_stats(g["stats"][i], stats(v.cpu()), k)
    assert [pg["lr"] for pg in m.G_opt.param_groups] == list(g["lr_groups"])
    dev = device or torch.device("cpu")
    x, probs, z = (torch.from_numpy(g[n]).to(dev) for n in ("x", "probs", "z"))
    tol = 2e-5
    close(g["d_cond"], m.D(x, probabilities=probs).cpu(), tol)
    close(g["enc_out"], m.encoder(x).cpu(), tol)
    close(g["w"], stn.latent_to_w(m.S, [(z, m.G.num_layers)], probs)[0][0].cpu(), tol)


def test_newarch_init_and_conditional_d_cpu():
    check_newarch_init()


def test_newarch_step_parity_cpu(tmp_path):
    """Trainer.train() x 3 of the reference's stylex_train_new.py (conditional D, probabilities in W, encoder lr group,
    single backward of gen + rec + kl)."""
    import stylex_train_new as stn

    g = load_golden("steps_newarch")
    tr, n = make_trainer(g, tmp_path, trainer_cls=stn.Trainer)
    assert tr.new_architecture and tr.rec_scaling == 2 and tr.kl_scaling == 2
    rows = run_steps(tr, n)
    gold = g["scalars"]
    np.testing.assert_allclose(rows[0], gold[0], rtol=5e-5, atol=5e-6, equal_nan=True)
    np.testing.assert_allclose(rows, gold, rtol=1e-3, atol=1e-3, equal_nan=True)
    assert_param_stats(tr, g)


def make_resnet_wrapper(g, tmp_path, size, norm):
    """The product's ResNet wrapper loading the A16 fixture's seeded checkpoint from ./trained_classifiers, the way
    Trainer(classifier_name='resnet') does (reference resnet_classifier.py:16-26)."""
    import os

    import resnet_classifier as rc
    from standins import seeded_resnet_state

    os.makedirs(os.path.join(str(tmp_path), "trained_classifiers"), exist_ok=True)
    torch.save(seeded_resnet_state(int(g["cls_seed"])), os.path.join(str(tmp_path), "trained_classifiers", "rn18.pth"))
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        return rc.ResNet("rn18.pth", 0, output_size=2, image_size=size, normalize=norm)
    finally:
        os.chdir(cwd)


def test_kl_rec_during_disc_is_rejected_like_the_reference_fails(tmp_path):
    """stylex_train_new.py:1392-1415.  With kl_rec_during_disc=True the reference's own train() raises on its first call
    (two backwards through one graph; transcript of the reference run in profiles/r05_kl_rec_during_disc_reference.txt), so
    there is no behaviour to reproduce: the Trainer refuses the option up front and says why; False is the default path
    the newarch goldens pin."""
    import stylex_train_new as stn
    from lpips_standin import LPIPSStandIn
    from standins import TinyClassifier

    kw = dict(name="k", base_dir=str(tmp_path), image_size=16, network_capacity=2, fmap_max=16, batch_size=2,
              classifier=TinyClassifier(seed=99), lpips_fn=LPIPSStandIn(seed=4242), device=torch.device("cpu"))
    with pytest.raises(RuntimeError, match="backwards twice through one graph"):
        stn.Trainer(kl_rec_during_disc=True, **kw)
    assert stn.Trainer(kl_rec_during_disc=False, **kw).new_architecture


def test_resnet_wrapper_vs_reference_golden_cpu(tmp_path):
    """A16 on the CPU double: ResNet.classify_images of the product against the reference wrapper's own logits and
    input gradient (tests/golden/resnet_wrapper.npz)."""
    from test_oracle_vs_golden import close, resnet_wrapper_cases

    g = load_golden("resnet_wrapper")
    for tag, size, norm, x, coef, logits, gx in resnet_wrapper_cases(g):
        clf = make_resnet_wrapper(g, tmp_path, size, norm)
        assert not clf.model.training and all(not p.requires_grad for p in clf.model.parameters())
        x = x.clone().requires_grad_(True)
        out = clf.classify_images(x)
        close(logits, out, 1e-5)
        got, = torch.autograd.grad((out * coef).sum(), x)
        close(gx, got, 1e-5)


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("tag", ["gae2_alt", "gae2_pl", "gae1_alt"])
def test_draw_ahead_worker_keeps_the_reference_draw_order(tag, mode, tmp_path, monkeypatch):
    """Round 4: the CPU-generator draws of a phase run on a worker thread under the previous phase's kernel enqueue
    (Trainer._draw_mode, the GPU default).  The random streams (loader, Python random(), torch's CPU generator) must be
    consumed in exactly the reference's order: the step goldens captured from the reference hold with the worker on —
    within a call (mode 1) and across calls (mode 2: the next call's discriminator-phase draws are prefetched)."""
    monkeypatch.setenv("STYLEX_DRAW_AHEAD", mode)
    g = load_golden("steps_" + tag)
    tr, n = make_trainer(g, tmp_path)
    assert tr._draw_mode == int(mode)
    rows = []
    for k in range(n):
        rows.append(run_steps(tr, 1)[0])
        if mode == "2" and k + 1 < n:
            done = tr.steps - 1  # a call at step 0 (or any step that evaluates / saves) ends with draws: no prefetch there
            ends_with_draws = done % 100 == 0 and done < 2500
            assert (tr._next_d is None) == ends_with_draws, "prefetch of the next call's discriminator-phase draws"
            tr._drain_draw_ahead()
    gold = g["scalars"]
    np.testing.assert_allclose(np.array(rows)[0], gold[0], rtol=5e-5, atol=5e-6, equal_nan=True)
    np.testing.assert_allclose(np.array(rows), gold, rtol=1e-3, atol=1e-3, equal_nan=True)
    assert tr._draw_worker is not None
