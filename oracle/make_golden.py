"""TEST INFRASTRUCTURE — golden-vector generator.  Runs ONLY in the build
container: it imports the reference (/root/reference) through ``ref_shim`` and
writes input/expected-output vectors to ``tests/golden/*.npz``.  The vectors are
data; no reference source travels.

    python oracle/make_golden.py [--only init,ops,nets,losses,steps,cfg4,resnet,dataset,newarch,diffaug,curve,calm,evalsurface,envelope]

Conventions: every fixture stores the seeds needed to regenerate weights
(``torch.manual_seed(seed)`` then construct ``StylEx(...)``), all inputs that are
not weights, and the reference's outputs.  Large tensors are stored as
``stats`` = (sum, abs-sum, first 8 values).
"""
import argparse
import os
import random
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def stats(t):
    t = t.detach().double().reshape(-1)
    return np.concatenate([[t.sum().item(), t.abs().sum().item()], t[:8].numpy(), np.zeros(max(0, 8 - t.numel()))])


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    clean = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        clean[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **clean)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


INIT_CONFIGS = [(8, 2, 16), (16, 4, 32), (32, 4, 64)]


def gen_init(st):
    """(1) init parity: state dict of StylEx under torch.manual_seed(seed)."""
    for (size, cap, fmax) in INIT_CONFIGS:
        seed = 100 + size
        torch.manual_seed(seed)
        m = st.StylEx(image_size=size, network_capacity=cap, fmap_max=fmax)
        sd = m.state_dict()
        out = {"seed": seed, "config": np.array([size, cap, fmax]), "keys": np.array(list(sd.keys()))}
        out["shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
        out["stats"] = np.stack([stats(v) for v in sd.values()])
        for k, v in sd.items():
            if v.numel() <= 2048:
                out["full/" + k] = v
        save("init_%d" % size, **out)


def gen_ops(st):
    """(2) op parity: Conv2DMod fwd+grads, Blur, Upsample, blocks, StyleVectorizer."""
    out = {}
    g = torch.Generator().manual_seed(11)
    # Blur / Upsample on a non-square odd-ish tensor
    x = torch.randn(2, 5, 6, 10, generator=g)
    out["blur/x"] = x
    out["blur/y"] = st.Blur()(x)
    out["up/x"] = x
    out["up/y"] = torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)(x)
    # gradient of both (they are linear; the adjoint is what the backward kernels implement)
    for nm, fn in (("blur", st.Blur()), ("up", torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))):
        xr = x.clone().requires_grad_()
        y = fn(xr)
        r = torch.randn(y.shape, generator=g)
        (y * r).sum().backward()
        out[nm + "/r"] = r
        out[nm + "/gx"] = xr.grad
    # Conv2DMod cases
    for tag, (ci, co, k, demod, hw, b, wseed) in {
        "mod3": (16, 24, 3, True, 8, 2, 21),
        "mod1": (16, 3, 1, False, 8, 2, 22),
        "mod512": (512, 512, 3, True, 4, 2, 23),
    }.items():
        torch.manual_seed(wseed)
        conv = st.Conv2DMod(ci, co, k, demod=demod)
        x = torch.randn(b, ci, hw, hw, generator=g, requires_grad=True)
        y = torch.randn(b, ci, generator=g, requires_grad=True)
        o = conv(x, y)
        r = torch.randn(o.shape, generator=g)
        (o * r).sum().backward()
        out[tag + "/cfg"] = np.array([ci, co, k, int(demod), hw, b, wseed])
        out[tag + "/x"], out[tag + "/y"], out[tag + "/r"] = x, y, r
        out[tag + "/out"], out[tag + "/gx"], out[tag + "/gy"] = o, x.grad, y.grad
        if conv.weight.numel() <= 8192:
            out[tag + "/w"], out[tag + "/gw"] = conv.weight, conv.weight.grad
        else:
            out[tag + "/w_stats"], out[tag + "/gw_stats"] = stats(conv.weight), stats(conv.weight.grad)
            out[tag + "/gw_slice"] = conv.weight.grad[:4, :4]
    # GeneratorBlock incl. transposed noise (non-zero to_noise weights so the permute matters)
    torch.manual_seed(31)
    blk = st.GeneratorBlock(20, 8, 12, upsample=True, upsample_rgb=True)
    x = torch.randn(2, 8, 4, 4, generator=g)
    prev = torch.randn(2, 3, 8, 8, generator=g)
    istyle = torch.randn(2, 20, generator=g)
    inoise = torch.rand(2, 16, 16, 1, generator=g)
    xo, rgb, sc = blk(x, prev, istyle, inoise)
    out["gblock/seed"] = 31
    out["gblock/x"], out["gblock/prev"], out["gblock/istyle"], out["gblock/inoise"] = x, prev, istyle, inoise
    out["gblock/xo"], out["gblock/rgb"], out["gblock/coords"] = xo, rgb, sc
    for k2, v in blk.state_dict().items():
        out["gblock/sd/" + k2] = v
    # DiscriminatorBlock
    torch.manual_seed(32)
    dblk = st.DiscriminatorBlock(6, 10, downsample=True)
    x = torch.randn(2, 6, 8, 8, generator=g)
    out["dblock/seed"] = 32
    out["dblock/x"], out["dblock/y"] = x, dblk(x)
    for k2, v in dblk.state_dict().items():
        out["dblock/sd/" + k2] = v
    # StyleVectorizer (small emb)
    torch.manual_seed(33)
    sv = st.StyleVectorizer(24, 8, lr_mul=0.1)
    z = torch.randn(3, 24, generator=g)
    out["svec/seed"] = 33
    out["svec/z"], out["svec/w"] = z, sv(z)
    save("ops", **out)


def gen_nets(st):
    """(3) network parity for image_size 16 & 32, cap 4."""
    for (size, cap, fmax) in [(16, 4, 32), (32, 4, 64)]:
        seed = 200 + size
        torch.manual_seed(seed)
        m = st.StylEx(image_size=size, network_capacity=cap, fmap_max=fmax)
        # make the noise path non-trivial: perturb to_noise params deterministically
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for blk in m.G.blocks:
                for lin in (blk.to_noise1, blk.to_noise2):
                    lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.3)
                    lin.bias.copy_(torch.randn(lin.bias.shape, generator=g) * 0.1)
        w = torch.randn(2, m.G.num_layers, 514, generator=g)
        inoise = torch.rand(2, size, size, 1, generator=g)
        x = torch.rand(2, 3, size, size, generator=g)
        rgb, coords = m.G(w, inoise, get_style_coords=True)
        save("nets_%d" % size, seed=seed, config=np.array([size, cap, fmax]), w=w, inoise=inoise, x=x,
             rgb=rgb, coords=coords, d_out=m.D(x), enc_out=m.encoder(x), d_of_g=m.D(rgb),
             s_out=m.S(w[:, 0]))


def gen_losses(st):
    """(4) loss parity incl. double backward (GP) and PL lengths."""
    size, cap, fmax = 16, 4, 32
    seed = 300
    torch.manual_seed(seed)
    m = st.StylEx(image_size=size, network_capacity=cap, fmap_max=fmax)
    g = torch.Generator().manual_seed(seed + 1)
    out = {"seed": seed, "config": np.array([size, cap, fmax])}
    real = torch.randn(4, generator=g)
    fake = torch.randn(4, generator=g)
    out["hinge/real"], out["hinge/fake"] = real, fake
    out["hinge/d"], out["hinge/g"] = st.hinge_loss(real, fake), st.gen_hinge_loss(fake, None)
    # gradient penalty value + resulting D grads (double backward)
    x = torch.rand(3, 3, size, size, generator=g).requires_grad_()
    d_out = m.D(x)
    gp = st.gradient_penalty(x, d_out)
    m.D.zero_grad()
    gp.backward()
    out["gp/x"], out["gp/value"] = x.detach(), gp.detach()
    names, gstats = [], []
    for n, p in m.D.named_parameters():
        if p.grad is None:  # GP does not depend on every bias
            continue
        names.append(n)
        gstats.append(stats(p.grad))
    out["gp/grad_names"], out["gp/grad_stats"] = np.array(names), np.stack(gstats)
    out["gp/grad_fc_w"] = m.D.fc.weight.grad
    out["gp/grad_b0_res_w"] = m.D.blocks[0].conv_res.weight.grad
    # path lengths (+ grads of the PL loss wrt G params)
    w = torch.randn(3, m.G.num_layers, 514, generator=g).requires_grad_()
    inoise = torch.rand(3, size, size, 1, generator=g)
    img = m.G(w, inoise)
    torch.manual_seed(seed + 2)  # pl_noise draw inside calc_pl_lengths (:309)
    pl = st.calc_pl_lengths(w, img)
    m.G.zero_grad()
    ((pl - 0.3) ** 2).mean().backward()
    out["pl/w"], out["pl/inoise"], out["pl/noise_seed"], out["pl/lengths"] = w.detach(), inoise, seed + 2, pl.detach()
    names, gstats = [], []
    for n, p in m.G.named_parameters():
        if p.grad is not None:
            names.append(n)
            gstats.append(stats(p.grad))
    out["pl/grad_names"], out["pl/grad_stats"] = np.array(names), np.stack(gstats)
    out["pl/grad_w"] = w.grad
    # KL + reconstruction (LPIPS stand-in, seed recorded)
    a = torch.randn(4, 2, generator=g)
    b = torch.randn(4, 2, generator=g)
    out["kl/real"], out["kl/fake"], out["kl/value"] = a, b, st.classifier_kl_loss(a, b)
    i1 = torch.rand(2, 3, 32, 32, generator=g)
    i2 = torch.rand(2, 3, 32, 32, generator=g)
    w1 = torch.randn(2, 512, generator=g)
    w2 = torch.randn(2, 512, generator=g)
    out["rec/i1"], out["rec/i2"], out["rec/w1"], out["rec/w2"] = i1, i2, w1, w2
    out["rec/value"] = st.reconstruction_loss(i1, i2, w2, w1)
    out["rec/lpips_seed"] = 4242
    out["rec/lpips_value"] = st.lpips_loss(st.lpips_normalize(i1), st.lpips_normalize(i2)).reshape(-1)
    save("losses", **out)


STEP_CASES = {
    # tag: (gae, alternating, n_steps, start_step, pl_mean0)
    "gae1_alt": (1, True, 5, 0, None),
    "gae2_alt": (2, True, 5, 0, None),
    "gae2_noalt": (2, False, 3, 0, None),
    "gae2_pl": (2, True, 2, 5024, 0.05),
    # discriminator augmentation on (AugWrapper.forward :558-571 + diff_augment.py, translation + cutout): every
    # micro-step is its own D pass, the random() gates and the augmentation parameters interleave with the other draws
    "gae2_aug": (2, True, 3, 0, None, 0.6),
}


def param_stats(model):
    names, st_ = [], []
    for n, p in model.named_parameters():
        names.append(n)
        st_.append(stats(p))
    return np.array(names), np.stack(st_)


def gen_steps(st, only=None):
    """(5) step parity: Trainer.train() x N on synthetic data."""
    size, cap, fmax, bs = 32, 4, 64, 2
    for tag, case in STEP_CASES.items():
        if only and tag not in only:
            continue
        (gae, alt, n, start, pl0), aug = case[:5], (case[5] if len(case) > 5 else 0.)
        cls = ref_shim.TinyClassifier(seed=99)
        gd = torch.Generator().manual_seed(7)
        batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
        seed_all(42)
        tmp = tempfile.mkdtemp()
        tr = ref_shim.make_reference_trainer(st, tmp, cls, batches, image_size=size, network_capacity=cap,
                                             fmap_max=fmax, batch_size=bs, gradient_accumulate_every=gae,
                                             alternating_training=alt, lr=2e-4, ttur_mult=1.5, rec_scaling=1,
                                             kl_scaling=1, aug_prob=aug)
        tr.init_StylEx()
        tr.steps = start
        tr.pl_mean = pl0
        rows = []
        for i in range(n):
            tr.train()
            rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss,
                         tr.last_gp_loss if tr.last_gp_loss is not None else np.nan,
                         tr.pl_mean if tr.pl_mean is not None else np.nan])
            print(tag, i, rows[-1])
        names, pst = param_stats(tr.StylEx)
        save("steps_" + tag, config=np.array([size, cap, fmax, bs, gae, int(alt), n, start]),
             pl_mean0=np.nan if pl0 is None else pl0, data_seed=7, seed=42, cls_seed=99, lpips_seed=4242,
             scalars=np.array(rows, dtype=np.float64), param_names=names, param_stats=pst, aug_prob=aug)


def _env_batches(variant, bs=2, size=32):
    gd = torch.Generator().manual_seed(7)
    batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
    if variant.startswith("eps"):
        ge = torch.Generator().manual_seed(1000 + int(variant[3:]))
        batches = [b * (1 + 1e-6 * torch.randn(b.shape, generator=ge)) for b in batches]
    return batches


def _env_rows(tr, n):
    rows = []
    for _ in range(n):
        tr.train()
        rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss,
                     tr.last_gp_loss if tr.last_gp_loss is not None else np.nan,
                     tr.pl_mean if tr.pl_mean is not None else np.nan])
    return np.array(rows, dtype=np.float64)


def gen_steps_envelope(st):
    """The reference against ITSELF on the step fixtures: every STEP_CASES run (and the config-4 and second-architecture
    runs) repeated (a) with 2 intra-op threads (CPU summation order) and (b) twice with the loader batches perturbed by
    a relative 1e-6 (two noise seeds) — the size of an fp32 kernel's summation-order difference.  The per-call spread
    of the loss scalars around the golden trajectory is what ``tests/test_hip_parity.py::assert_trajectory`` scales its
    tolerance by (instead of a free growth factor): an implementation that is correct but rounds differently cannot
    be expected inside a band narrower than the one the reference leaves around itself."""
    size, cap, fmax, bs = 32, 4, 64, 2
    keep = torch.get_num_threads()

    def build_step(tag):
        case = STEP_CASES[tag]
        (gae, alt, n, start, pl0), aug = case[:5], (case[5] if len(case) > 5 else 0.)

        def run(variant):
            seed_all(42)
            tr = ref_shim.make_reference_trainer(st, tempfile.mkdtemp(), ref_shim.TinyClassifier(seed=99),
                                                 _env_batches(variant), image_size=size, network_capacity=cap,
                                                 fmap_max=fmax, batch_size=bs, gradient_accumulate_every=gae,
                                                 alternating_training=alt, lr=2e-4, ttur_mult=1.5, rec_scaling=1,
                                                 kl_scaling=1, aug_prob=aug)
            tr.init_StylEx()
            tr.steps = start
            tr.pl_mean = pl0
            return _env_rows(tr, n)
        return run

    def run_cfg4(variant):
        mod = ref_shim.import_reference_mobilenet()
        tmp = tempfile.mkdtemp()
        os.makedirs(os.path.join(tmp, "trained_classifiers"))
        torch.save(ref_shim.seeded_mobilenet_state(77), os.path.join(tmp, "trained_classifiers", "mnv2_seed77.pth"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            st.MobileNet = mod.MobileNet
            seed_all(42)
            tr = st.Trainer(name="gold", base_dir=tmp, classifier_name="mobilenet", classifier_path="mnv2_seed77.pth",
                            tensorboard_dir=None, evaluate_every=10 ** 9, save_every=10 ** 9, image_size=size,
                            network_capacity=cap, fmap_max=fmax, batch_size=bs, gradient_accumulate_every=2,
                            alternating_training=True, lr=2e-4, ttur_mult=1.5, rec_scaling=1, kl_scaling=1)
            seed_all(42)
            tr.loader = st.cycle(_env_batches(variant))
            tr.dataset = list(range(1000))
            tr.save = tr.evaluate = lambda *a, **k: None
            tr.init_StylEx()
            tr.steps = 5024
            tr.pl_mean = 0.05
            return _env_rows(tr, 2)
        finally:
            os.chdir(cwd)

    def run_newarch(variant):
        stn = ref_shim.import_reference_new()
        seed_all(42)
        tr = ref_shim.make_reference_trainer(stn, tempfile.mkdtemp(), ref_shim.TinyClassifier(seed=99),
                                             _env_batches(variant), image_size=size, network_capacity=cap,
                                             fmap_max=fmax, batch_size=bs, gradient_accumulate_every=2,
                                             alternating_training=True, lr=2e-4, ttur_mult=1.5, rec_scaling=1,
                                             kl_scaling=1)
        tr.init_StylEx()
        rows = _env_rows(tr, 3)
        rows[:, 5] = np.nan  # the fixture records no pl_mean for this module
        return rows

    runners = {tag: build_step(tag) for tag in STEP_CASES}
    runners["cfg4"] = run_cfg4
    runners["newarch"] = run_newarch
    out = {}
    for tag, run in runners.items():
        runs = []
        for variant in ("gold", "t2", "eps1", "eps2"):
            torch.set_num_threads(2 if variant == "t2" else keep)
            runs.append(run(variant))
        torch.set_num_threads(keep)
        gold = runs[0]
        committed = np.load(os.path.join(OUT, "steps_%s.npz" % tag))["scalars"]
        assert np.array_equal(gold, committed, equal_nan=True), ("the gold variant must reproduce the committed fixture", tag)
        dev = np.stack([np.abs(r - gold) / np.maximum(np.abs(gold), 1e-2) for r in runs[1:]])  # [variant, call, scalar]
        spread = np.nanmax(dev, axis=(0, 2))
        print("envelope", tag, "per-call max relative spread:", " ".join("%.2e" % v for v in spread), flush=True)
        out["spread_" + tag] = spread
        out["dev_" + tag] = dev
    save("steps_envelope", variants=np.array(["t2", "eps1", "eps2"]), eps=1e-6, **out)


def gen_cfg4(st):
    """BASELINE config 4 in miniature: MobileNetV2 classifier through the reference's own wrapper
    (stylex/mobilenet_classifier.py:57-73: nearest interpolate to image_size, ImageNet normalise), R1 every 4th
    step + path-length regularisation (step 5024: both penalties; reference interval 32, :1272-1273)."""
    size, cap, fmax, bs, gae, n, start, pl0 = 32, 4, 64, 2, 2, 2, 5024, 0.05
    mod = ref_shim.import_reference_mobilenet()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "trained_classifiers"))
    torch.save(ref_shim.seeded_mobilenet_state(77), os.path.join(tmp, "trained_classifiers", "mnv2_seed77.pth"))
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        st.MobileNet = mod.MobileNet  # Trainer(classifier_name='mobilenet') builds it (:1104-1107)
        gd = torch.Generator().manual_seed(7)
        batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
        seed_all(42)
        tr = st.Trainer(name="gold", base_dir=tmp, classifier_name="mobilenet", classifier_path="mnv2_seed77.pth",
                        tensorboard_dir=None, evaluate_every=10 ** 9, save_every=10 ** 9, image_size=size,
                        network_capacity=cap, fmap_max=fmax, batch_size=bs, gradient_accumulate_every=gae,
                        alternating_training=True, lr=2e-4, ttur_mult=1.5, rec_scaling=1, kl_scaling=1)
        assert isinstance(tr.classifier, mod.MobileNet)
        # constructing the classifier drew from the global RNG (torch.hub's constructor does too, differently):
        # re-seed so that the fixture does not depend on how many numbers a model constructor consumes
        seed_all(42)
        tr.loader = st.cycle(batches)
        tr.dataset = list(range(1000))
        tr.save = lambda *a, **k: None
        tr.evaluate = lambda *a, **k: None
        tr.init_StylEx()
        tr.steps = start
        tr.pl_mean = pl0
        logits = tr.classifier.classify_images(batches[0])
        rows = []
        for i in range(n):
            tr.train()
            rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss,
                         tr.last_gp_loss if tr.last_gp_loss is not None else np.nan,
                         tr.pl_mean if tr.pl_mean is not None else np.nan])
            print("cfg4", i, rows[-1])
    finally:
        os.chdir(cwd)
    names, pst = param_stats(tr.StylEx)
    save("steps_cfg4", config=np.array([size, cap, fmax, bs, gae, 1, n, start]), pl_mean0=pl0, data_seed=7, seed=42,
         cls_seed=77, lpips_seed=4242, scalars=np.array(rows, dtype=np.float64), param_names=names, param_stats=pst,
         logits_batch0=logits)


def gen_resnet(st):
    """A16 — the reference's own ``ResNet`` wrapper (stylex/resnet_classifier.py:29-71: torchvision ``resize`` of the
    tensor batch to 224x224, ImageNet normalisation, ResNet-18 with a 2-logit head, eval mode, frozen parameters,
    gradient still flowing to the images) on seeded weights: logits and the input gradient of ``sum(logits * coef)``
    at 32 and 64 px, with and without normalisation."""
    mod = ref_shim.import_reference_resnet()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "trained_classifiers"))
    cls_seed = 55
    torch.save(ref_shim.seeded_resnet_state(cls_seed), os.path.join(tmp, "trained_classifiers", "rn18_seed55.pth"))
    cwd = os.getcwd()
    os.chdir(tmp)
    out = {}
    try:
        for size in (32, 64):
            for norm in (True, False):
                clf = mod.ResNet("rn18_seed55.pth", 0, output_size=2, image_size=size, normalize=norm)
                assert not clf.model.training and all(not p.requires_grad for p in clf.model.parameters())
                gd = torch.Generator().manual_seed(100 + size)
                x = (torch.rand(3, 3, size, size, generator=gd) * 1.5 - 0.25).requires_grad_(True)  # G's range is not [0,1]
                coef = torch.randn(3, 2, generator=gd)
                logits = clf.classify_images(x)
                gx, = torch.autograd.grad((logits * coef).sum(), x)
                tag = "%d_%s" % (size, "norm" if norm else "raw")
                out["logits_" + tag] = logits.detach()
                out["gx_" + tag] = gx
                out["coef_" + tag] = coef
                print("resnet", tag, logits.detach().numpy().ravel(), float(gx.abs().sum()))
    finally:
        os.chdir(cwd)
    save("resnet_wrapper", cls_seed=cls_seed, sizes=np.array([32, 64]), batch=3, data_seed_base=100, **out)


DATASET_IMAGES = [
    # (mode, height, width): the training size is 32
    ("RGB", 32, 32), ("RGB", 48, 40), ("RGB", 40, 64), ("RGB", 70, 33), ("RGB", 33, 70), ("RGB", 32, 57),
    ("RGB", 37, 32), ("RGB", 20, 14), ("L", 45, 36), ("RGBA", 40, 40), ("LA", 36, 50), ("P", 64, 64),
]


def gen_dataset(st):
    """N2 — the reference's ``Dataset`` (stylex_train.py:520-547) run on PNG files of assorted sizes and modes:
    mode conversion, ``resize_to_minimum_size`` (:480-483), ``Resize`` + ``RandomApply(aug_prob, RandomResizedCrop,
    CenterCrop)`` (one Python ``random()`` per item even at aug_prob 0), ``ToTensor``, ``expand_greyscale``.  The
    torchvision 0.11.1 calls underneath are the restatements of ``ref_shim`` (package absent offline).  The fixture
    stores the PNG FILES (bytes) as inputs and the tensors ``__getitem__`` returned, for aug_prob 0 and 1 and for
    ``transparent=True``; plus the state of Python's ``random`` after the items, which pins the draw count."""
    import io

    from PIL import Image

    size = 32
    rng = np.random.RandomState(11)
    tmp = tempfile.mkdtemp()
    out = {}
    for i, (mode, h, w) in enumerate(DATASET_IMAGES):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 100 * np.sin(xx / 5.0 + i), 127 + 100 * np.cos(yy / 4.0), (7 * xx + 3 * yy) % 256,
                         255 * ((xx + yy) % 7 > 2)], axis=-1)
        base = np.clip(base + rng.randint(-20, 20, base.shape), 0, 255).astype(np.uint8)
        img = {"RGB": lambda: Image.fromarray(base[..., :3], "RGB"), "RGBA": lambda: Image.fromarray(base, "RGBA"),
               "L": lambda: Image.fromarray(base[..., 0], "L"),
               "LA": lambda: Image.fromarray(np.ascontiguousarray(base[..., [0, 3]]), "LA"),
               "P": lambda: Image.fromarray(base[..., :3], "RGB").quantize(16)}[mode]()
        buf = io.BytesIO()
        img.save(buf, format="PNG")
        out["png_%02d" % i] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
        with open(os.path.join(tmp, "%02d.png" % i), "wb") as f:
            f.write(buf.getvalue())
    for tag, kw in (("p0", dict(aug_prob=0.)), ("p1", dict(aug_prob=1.)), ("half", dict(aug_prob=0.5)),
                    ("transparent", dict(transparent=True))):
        ds = st.Dataset(tmp, size, **kw)
        order = sorted(range(len(ds)), key=lambda k: ds.paths[k].name)
        seed_all(5)
        items = [ds[k] for k in order]
        assert all(t.shape == (4 if tag == "transparent" else 3, size, size) for t in items), [t.shape for t in items]
        out["items_" + tag] = torch.stack(items)
        out["pyrandom_after_" + tag] = random.random()
        out["torchrand_after_" + tag] = torch.rand(()).item()
        print("dataset", tag, float(out["items_" + tag].sum()))
    save("dataset_items", image_size=size, modes=np.array([m for m, _, _ in DATASET_IMAGES]),
         shapes=np.array([[h, w] for _, h, w in DATASET_IMAGES]), seed=5, **out)


def gen_newarch(st):
    """N4: the reference's second architecture (stylex/stylex_train_new.py, cli.py:17-22): init parity, conditional-D
    forward, and Trainer.train() x 3 (GAE=2 alternating; step 0 carries the gradient penalty)."""
    stn = ref_shim.import_reference_new()
    size, cap, fmax, bs, gae, n = 32, 4, 64, 2, 2, 3
    # init + nets
    torch.manual_seed(432)
    m = stn.StylEx(image_size=size, network_capacity=cap, fmap_max=fmax)
    sd = m.state_dict()
    out = {"seed": 432, "config": np.array([size, cap, fmax]), "keys": np.array(list(sd.keys())),
           "shapes": np.array([",".join(map(str, v.shape)) for v in sd.values()]),
           "stats": np.stack([stats(v) for v in sd.values()])}
    g = torch.Generator().manual_seed(433)
    x = torch.rand(3, 3, size, size, generator=g)
    probs = torch.softmax(torch.randn(3, 2, generator=g), dim=1)
    z = torch.randn(3, 512, generator=g)
    out["x"], out["probs"], out["z"] = x, probs, z
    out["d_cond"] = m.D(x, probabilities=probs)
    out["enc_out"] = m.encoder(x)
    w = stn.latent_to_w(m.S, [(z, m.G.num_layers)], probs)
    out["w"] = w[0][0]
    out["lr_groups"] = np.array([pg["lr"] for pg in m.G_opt.param_groups])
    save("newarch_init", **out)
    # steps
    cls = ref_shim.TinyClassifier(seed=99)
    gd = torch.Generator().manual_seed(7)
    batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
    seed_all(42)
    tr = ref_shim.make_reference_trainer(stn, tempfile.mkdtemp(), cls, batches, image_size=size, network_capacity=cap,
                                         fmap_max=fmax, batch_size=bs, gradient_accumulate_every=gae,
                                         alternating_training=True, lr=2e-4, ttur_mult=1.5, rec_scaling=1, kl_scaling=1)
    tr.init_StylEx()
    rows = []
    for i in range(n):
        tr.train()
        rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss,
                     tr.last_gp_loss if tr.last_gp_loss is not None else np.nan, np.nan])
        print("newarch", i, rows[-1])
    names, pst = param_stats(tr.StylEx)
    save("steps_newarch", config=np.array([size, cap, fmax, bs, gae, 1, n, 0]), pl_mean0=np.nan, data_seed=7, seed=42,
         cls_seed=99, lpips_seed=4242, scalars=np.array(rows, dtype=np.float64), param_names=names, param_stats=pst)


def gen_diffaug(st):
    """N2: DiffAugment (stylex/diff_augment.py) per augmentation type and through AugWrapper.forward (:558-571: the
    random() gate, random_hflip, DiffAugment) on a non-square batch; seeds recorded, CPU generator."""
    da = ref_shim._load_by_path("ref_diff_augment", os.path.join(ref_shim.REF_STYLEX, "diff_augment.py"))
    g = torch.Generator().manual_seed(5)
    x = torch.rand(3, 3, 16, 12, generator=g)
    r = torch.randn(3, 3, 16, 12, generator=g)
    out = {"x": x, "r": r, "types": np.array(sorted(da.AUGMENT_FNS))}
    for i, t in enumerate(sorted(da.AUGMENT_FNS)):
        for rep in range(2):
            seed_all(900 + 10 * i + rep)
            xr = x.clone().requires_grad_()
            y = da.DiffAugment(xr, types=[t])
            (y * r).sum().backward()
            out["%s/%d/y" % (t, rep)], out["%s/%d/gx" % (t, rep)] = y, xr.grad
            out["%s/%d/seed" % (t, rep)] = 900 + 10 * i + rep

    class Ident(torch.nn.Module):
        def forward(self, im):
            return im

    wrap = st.AugWrapper(Ident(), 16)
    for k in range(6):
        seed_all(950 + k)
        out["wrap/%d/y" % k] = wrap(x, prob=0.7, types=["translation", "cutout"], detach=True)
        out["wrap/%d/after" % k] = np.array([random.random(), float(torch.rand(()))])  # RNG consumption
    save("diffaug", **out)


def gen_curve(st, n=100):
    """100-step scalar trajectory, config 1 shape (64 px, B=4) at GAE=2."""
    size, cap, fmax, bs, gae = 64, 16, 512, 4, 2
    cls = ref_shim.TinyClassifier(seed=99)
    gd = torch.Generator().manual_seed(7)
    batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
    seed_all(42)
    tr = ref_shim.make_reference_trainer(st, tempfile.mkdtemp(), cls, batches, image_size=size,
                                         network_capacity=cap, fmap_max=fmax, batch_size=bs,
                                         gradient_accumulate_every=gae, lr=2e-4, ttur_mult=1.5, rec_scaling=1,
                                         kl_scaling=1)
    rows, t0 = [], time.time()
    for i in range(n):
        tr.train()
        rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss,
                     tr.last_gp_loss if tr.last_gp_loss is not None else np.nan])
        if i % 10 == 0:
            print("curve", i, rows[-1], "%.0fs" % (time.time() - t0))
    save("curve_64", config=np.array([size, cap, fmax, bs, gae]), seed=42, data_seed=7, cls_seed=99,
         lpips_seed=4242, scalars=np.array(rows, dtype=np.float64), wall_s=time.time() - t0,
         threads=torch.get_num_threads())


def gen_curve_calm(st, n=100, lr=1e-8, start=4960):
    """X1 — a 100-call scalar trajectory of the reference in a NON-chaotic regime: config-1 shape (64 px, capacity 16,
    B=4) at GAE=2 with lr = 1e-8, so that Adam's sign-like steps (<= lr per element and call) stay far below the
    amplification threshold of the untrained GAN and the trajectory is a function of the schedule and the random
    draws, not of rounding.  (Measured here first with lr = 1e-6: the reference at 8 vs 4 threads is 1e-3 apart at
    call 3 and 1e-1 apart from call 35 on — a +-1e-6 step on the noise-dominated share of 1e7 parameters already
    moves these losses by 1e-3.)  The window starts at step 4960 so that the 100 calls contain everything ``train()``
    schedules by step count (stylex_train.py:1272-1274, 1471-1479): the gradient penalty on every 4th call, the
    path-length penalty + ``pl_mean`` EMA at 5024 and 5056 (> 5000 and % 32 == 0, not 4992),
    ``reset_parameter_averaging`` at 5002, the noise / encoder alternation and the rec / KL cadence, and the RNG draw
    order of all of them.  Run twice — 8 and 4 intra-op threads — so the fixture carries the reference's own
    summation-order spread next to the trajectory (the tests hold 1e-3 on EVERY call; the spread shows that the
    reference holds it against itself here)."""
    size, cap, fmax, bs, gae = 64, 16, 512, 4, 2
    out, keep = {}, torch.get_num_threads()
    for threads in (8, 4):
        torch.set_num_threads(threads)
        cls = ref_shim.TinyClassifier(seed=99)
        gd = torch.Generator().manual_seed(7)
        batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
        seed_all(42)
        tr = ref_shim.make_reference_trainer(st, tempfile.mkdtemp(), cls, batches, image_size=size,
                                             network_capacity=cap, fmap_max=fmax, batch_size=bs,
                                             gradient_accumulate_every=gae, lr=lr, ttur_mult=1.5, rec_scaling=1,
                                             kl_scaling=1)
        tr.init_StylEx()
        tr.steps = start
        rows, t0 = [], time.time()
        for i in range(n):
            tr.train()
            rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss,
                         tr.last_gp_loss if tr.last_gp_loss is not None else np.nan,
                         tr.pl_mean if tr.pl_mean is not None else np.nan])
            if i % 10 == 0 or tr.steps - 1 in (5024, 5056):
                print("calm t%d" % threads, tr.steps - 1, rows[-1], "%.0fs" % (time.time() - t0), flush=True)
        out["scalars_t%d" % threads] = np.array(rows, dtype=np.float64)
        if threads == 8:
            names, pst = param_stats(tr.StylEx)
            out.update(param_names=names, param_stats=pst)
    torch.set_num_threads(keep)
    a, b = out["scalars_t8"], out["scalars_t4"]
    rel = np.nanmax(np.abs(a - b) / np.maximum(np.abs(a), 1e-3), axis=1)
    print("reference vs itself (8 vs 4 threads), max relative difference per call: max %.3e, at call %d"
          % (rel.max(), int(rel.argmax())))
    save("curve_64_calm", config=np.array([size, cap, fmax, bs, gae, 1, n, start]), lr=lr, seed=42, data_seed=7,
         cls_seed=99, lpips_seed=4242, scalars=out.pop("scalars_t8"), self_rel=rel, **out)


def gen_envelope(st, n=12):
    """X1 — the reference against ITSELF: the curve_64 run (config-1 shape, 8 threads) repeated with 4 and with 2
    intra-op threads.  Only the summation order of the CPU kernels changes, yet the untrained GAN amplifies it
    step by step; the spread of these trajectories is the envelope inside which any correct implementation's
    trajectory can be expected to stay (tests: HIP-vs-reference error <= c x reference-vs-reference error)."""
    size, cap, fmax, bs, gae = 64, 16, 512, 4, 2
    out = {}
    keep = torch.get_num_threads()
    for threads in (4, 2):
        torch.set_num_threads(threads)
        cls = ref_shim.TinyClassifier(seed=99)
        gd = torch.Generator().manual_seed(7)
        batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
        seed_all(42)
        tr = ref_shim.make_reference_trainer(st, tempfile.mkdtemp(), cls, batches, image_size=size,
                                             network_capacity=cap, fmap_max=fmax, batch_size=bs,
                                             gradient_accumulate_every=gae, lr=2e-4, ttur_mult=1.5, rec_scaling=1,
                                             kl_scaling=1)
        rows, t0 = [], time.time()
        for i in range(n):
            tr.train()
            rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss,
                         tr.last_gp_loss if tr.last_gp_loss is not None else np.nan])
            print("envelope t%d" % threads, i, rows[-1], "%.0fs" % (time.time() - t0))
        out["scalars_t%d" % threads] = np.array(rows, dtype=np.float64)
    torch.set_num_threads(keep)
    save("curve_64_envelope", config=np.array([size, cap, fmax, bs, gae]), seed=42, data_seed=7, cls_seed=99,
         lpips_seed=4242, threads=np.array([4, 2]), **out)


def gen_evalsurface(st):
    """N3 — evaluation / EMA / truncation surface of the reference Trainer (stylex_train.py:985-999 EMA +
    reset_parameter_averaging, :1508-1575 evaluate, :1624-1656 truncate_style / generate_truncated) at 16 px:
    three train() calls across step 20010 (the EMA update fires there), then truncate_style on a given tensor (the
    seeded 2000-sample W mean), then the three image grids evaluate() hands to save_image — with and without
    encoder input — captured from the reference's own call."""
    size, cap, fmax, bs = 16, 4, 32, 2
    cls = ref_shim.TinyClassifier(seed=99)
    gd = torch.Generator().manual_seed(7)
    batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
    seed_all(42)
    tr = ref_shim.make_reference_trainer(st, tempfile.mkdtemp(), cls, batches, image_size=size, network_capacity=cap,
                                         fmap_max=fmax, batch_size=bs, gradient_accumulate_every=1, lr=2e-4,
                                         ttur_mult=1.5, rec_scaling=1, kl_scaling=1, num_image_tiles=2)
    tr.init_StylEx()
    tr.steps = 20009
    rows = []
    for i in range(3):
        tr.train()
        rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss])
        print("evalsurface step", tr.steps - 1, rows[-1])
    names, pst = param_stats(tr.StylEx)  # S, G, D, encoder AND the averaged copies SE / GE after the EMA update
    out = dict(config=np.array([size, cap, fmax, bs, 1, 2]), seed=42, data_seed=7, cls_seed=99, lpips_seed=4242,
               start_step=20009, scalars=np.array(rows, dtype=np.float64), param_names=names, param_stats=pst)
    # truncate_style: W mean of 2000 seeded samples, then the affine pull towards it
    seed_all(123)
    tr.av = None
    t_in = torch.randn(4, tr.StylEx.G.latent_dim)
    out["trunc_in"], out["trunc_psi"] = t_in, 0.6
    out["trunc_out"] = st.Trainer.truncate_style(tr, t_in.clone(), trunc_psi=0.6)
    out["trunc_av"] = np.asarray(tr.av)
    # evaluate(): the tensors the reference passes to torchvision.utils.save_image
    grids = []
    st.torchvision.utils.save_image = lambda t, path, nrow=8, **k: grids.append((os.path.basename(str(path)), t.detach().clone(), nrow))
    for k, (seed, enc) in enumerate(((77, False), (78, True))):
        seed_all(seed)
        tr.av = None
        st.Trainer.evaluate(tr, encoder_input=enc, num=k)
    st.torchvision.utils.save_image = lambda *a, **k: None
    out["grid_names"] = np.array([g[0] for g in grids])
    out["grid_nrow"] = np.array([g[2] for g in grids])
    out["eval_seeds"] = np.array([77, 78])
    for i, g in enumerate(grids):
        out["grid/%d" % i] = g[1]
        print("grid", g[0], tuple(g[1].shape), g[2])
    # reset_parameter_averaging: the averaged copies become the live weights again
    tr.StylEx.reset_parameter_averaging()
    names2, pst2 = param_stats(tr.StylEx)
    out["param_stats_after_reset"] = pst2
    save("evalsurface_16", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="init,ops,nets,losses,steps")
    ap.add_argument("--step-cases", default="", help="comma list of STEP_CASES tags (default: all)")
    a = ap.parse_args()
    st = ref_shim.import_reference()
    todo = a.only.split(",")
    for name, fn in (("init", gen_init), ("ops", gen_ops), ("nets", gen_nets), ("losses", gen_losses),
                     ("steps", gen_steps), ("stepsenv", gen_steps_envelope), ("cfg4", gen_cfg4), ("resnet", gen_resnet), ("dataset", gen_dataset), ("newarch", gen_newarch), ("diffaug", gen_diffaug),
                     ("curve", gen_curve), ("calm", gen_curve_calm), ("evalsurface", gen_evalsurface), ("envelope", gen_envelope)):
        if name in todo:
            if name == "steps" and a.step_cases:
                fn(st, set(a.step_cases.split(",")))
            else:
                fn(st)


if __name__ == "__main__":
    main()
