"""N2: device-side input pipeline (input_pipeline.py) against the host pipeline (stylex_train.Dataset,
the restatement of the reference's transform chain stylex_train.py:520-547): identical geometry, bit-identical values
for images already at the training size, <= 1.5/255 where the host path rounds to uint8 after its resize; the prefetch
thread delivers the loader's batches in order."""
import numpy as np
import pytest
import torch

import input_pipeline as ip
import stylex_train as st


def make_folder(tmp_path, sizes):
    from PIL import Image

    rng = np.random.RandomState(3)
    d = tmp_path / "imgs"
    d.mkdir()
    for i, (h, w) in enumerate(sizes):
        # smooth content (resampling kernels differ most on noise): low-frequency pattern + mild noise
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 100 * np.sin(xx / 9.0 + i), 127 + 100 * np.cos(yy / 7.0), (xx + yy) % 256], axis=-1)
        img = np.clip(img + rng.randint(-8, 8, img.shape), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(d / f"{i:02d}.png")
    return d


def run(device, tmp_path):
    s = 32
    sizes = [(32, 32), (32, 32), (48, 40), (40, 64), (70, 33), (32, 57), (37, 32)]
    folder = make_folder(tmp_path, sizes)
    host = st.Dataset(str(folder), s)
    raw = ip.RawImageFolder(str(folder), s)
    assert [p.name for p in host.paths] == [p.name for p in raw.paths]
    pre = ip.DevicePreprocessor(s, device)
    for i in range(len(host)):
        want = host[i]
        got = pre([raw[i]])[0].cpu()
        assert got.shape == want.shape == (3, s, s)
        h, w = raw[i].shape[:2]
        if (h, w) == (s, s):
            assert torch.equal(got, want), "image at the training size must be bit-identical"
        else:
            assert float((got - want).abs().max()) <= 1.5 / 255 + 1e-6, (i, float((got - want).abs().max()))
    # geometry: the crop window of a non-square image is the host path's
    assert ip.target_geometry(70, 33, 32) == (67, 32, 18, 0)  # int(32 * 70 / 33) = 67 (truncated); round(17.5) = 18
    assert ip.target_geometry(37, 32, 32) == (37, 32, 2, 0)  # shorter side already 32: untouched; round(2.5) = 2
    for h, w in sizes:
        rh, rw, top, left = ip.target_geometry(h, w, s)
        assert (rw, rh) == st.resize_geometry(w, h, s) and (left, top) == st.center_crop_offsets(rw, rh, s)
    # prefetcher: same batches, same order as the plain loader
    batches = [[raw[i] for i in range(j, j + 2)] for j in range(0, 6, 2)]  # the first six images
    pf = ip.Prefetcher(iter(batches), pre, device, depth=2)
    got = list(pf)
    assert len(got) == 3
    for b, src in zip(got, batches):
        assert torch.equal(b.cpu(), pre(src).cpu())


def test_device_pipeline_cpu(tmp_path):
    run(torch.device("cpu"), tmp_path)


@pytest.mark.gpu
def test_device_pipeline_gpu(tmp_path):
    assert torch.cuda.is_available()
    run(torch.device("cuda:0"), tmp_path)


@pytest.mark.gpu
def test_trainer_with_device_pipeline_and_device_rng_gpu(tmp_path):
    """End to end on the GPU: cli-style Trainer with the device input pipeline and device RNG; losses finite, and the
    batches the Trainer consumes equal the host pipeline's for training-size images."""
    import ops
    from lpips_standin import LPIPSStandIn
    from standins import TinyClassifier

    dev = torch.device("cuda:0")
    folder = make_folder(tmp_path, [(32, 32)] * 8)
    prev = st._Staging.DEVICE_RNG
    try:
        tr = st.Trainer(name="p", base_dir=str(tmp_path), image_size=32, network_capacity=4, fmap_max=64, batch_size=2,
                        gradient_accumulate_every=2, classifier=TinyClassifier(seed=99).to(dev),
                        lpips_fn=LPIPSStandIn(seed=4242).to(dev), classifier_name="resnet", evaluate_every=10 ** 9,
                        save_every=10 ** 9, device=dev, device_pipeline=True, device_rng=True, num_workers=0)
        tr.set_data_src(str(folder))
        b = next(tr.loader)
        assert b.is_cuda and b.shape == (2, 3, 32, 32) and float(b.min()) >= 0 and float(b.max()) <= 1
        tr.save = lambda *a, **k: None
        tr.evaluate = lambda *a, **k: None
        for _ in range(2):
            tr.train()
        assert np.isfinite([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss]).all()
        tr.loader.close()
    finally:
        st._Staging.DEVICE_RNG = prev
        ops.set_precision("fp32")


def load_dataset_fixture(tmp_path):
    """tests/golden/dataset_items.npz -> (folder with the fixture's PNG files, fixture).  The PNG bytes are the INPUT
    data of the fixture; the expected tensors come from the reference's own Dataset class (oracle/make_golden.py
    gen_dataset)."""
    from conftest import load_golden

    g = load_golden("dataset_items")
    d = tmp_path / "fixture_imgs"
    d.mkdir()
    for i in range(len(g["modes"])):
        (d / ("%02d.png" % i)).write_bytes(g["png_%02d" % i].tobytes())
    return d, g


@pytest.mark.parametrize("tag,kw", [("p0", dict(aug_prob=0.)), ("p1", dict(aug_prob=1.)), ("half", dict(aug_prob=0.5)),
                                    ("transparent", dict(transparent=True))])
def test_dataset_getitem_vs_reference_golden(tmp_path, tag, kw):
    """N2, host leg: Dataset.__getitem__ against the tensors the REFERENCE's Dataset (stylex_train.py:520-547)
    returned for the same PNG files — RGB / greyscale / palette / alpha modes, sizes that exercise the truncated
    longer side, the round-half-even crop offset, the shorter-side-already-right shortcut and the minimum-size
    resize; aug_prob 0 / 0.5 / 1 (RandomResizedCrop parameters from torch's global generator), transparent=True —
    bit for bit, and the Python / torch generators end in the same state (same number of draws)."""
    import random

    folder, g = load_dataset_fixture(tmp_path)
    ds = st.Dataset(str(folder), int(g["image_size"]), **kw)
    order = sorted(range(len(ds)), key=lambda k: ds.paths[k].name)
    seed = int(g["seed"])
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    items = torch.stack([ds[k] for k in order])
    want = torch.from_numpy(g["items_" + tag])
    assert items.shape == want.shape
    bad = [i for i in range(len(order)) if not torch.equal(items[i], want[i])]
    assert not bad, ("items differing from the reference's", bad, [tuple(g["shapes"][i]) for i in bad])
    assert random.random() == float(g["pyrandom_after_" + tag])
    assert torch.rand(()).item() == float(g["torchrand_after_" + tag])


def check_device_pipeline_vs_reference(device, tmp_path):
    folder, g = load_dataset_fixture(tmp_path)
    s = int(g["image_size"])
    raw = ip.RawImageFolder(str(folder), s)
    order = sorted(range(len(raw)), key=lambda k: raw.paths[k].name)
    pre = ip.DevicePreprocessor(s, device)
    want = torch.from_numpy(g["items_p0"])
    for pos, k in enumerate(order):
        h, w = (int(v) for v in g["shapes"][pos])
        got = pre([raw[k]])[0].cpu()
        if min(h, w) == s:  # no resampling on either side: bit-identical to the reference's tensor
            assert torch.equal(got, want[pos]), (pos, h, w)
        else:  # the host path rounds to bytes after ITS resize; PIL's fixed-point bilinear vs the float one
            assert float((got - want[pos]).abs().max()) <= 1.5 / 255 + 1e-6, (pos, h, w)


def test_device_pipeline_vs_reference_golden_cpu(tmp_path):
    """N2, device leg against the REFERENCE's tensors (not against the product's own host Dataset)."""
    check_device_pipeline_vs_reference(torch.device("cpu"), tmp_path)


@pytest.mark.gpu
def test_device_pipeline_vs_reference_golden_gpu(tmp_path):
    assert torch.cuda.is_available()
    check_device_pipeline_vs_reference(torch.device("cuda:0"), tmp_path)
