"""The drop-in entry point end to end ON THE HIP PATH (bf16 speed mode): image folder -> Dataset/DataLoader ->
Trainer.train() with HIP streams / lazy loss scalars -> evaluate grids -> checkpoint -> reload -> generate."""
import json
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
for p in (os.path.join(PKG, "stylex"), PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


def test_train_from_folder_on_gpu_bf16(tmp_path):
    from PIL import Image

    import cli
    import ops
    import stylex_train as st

    data = tmp_path / "imgs"
    data.mkdir()
    rng = np.random.RandomState(0)
    for i in range(10):
        Image.fromarray(rng.randint(0, 255, (72, 80, 3), dtype=np.uint8)).save(data / f"{i}.png")
    kw = dict(data=str(data), results_dir=str(tmp_path / "results"), models_dir=str(tmp_path / "models"), name="g",
              image_size=64, network_capacity=8, fmap_max=128, batch_size=4, gradient_accumulate_every=2,
              num_workers=0, save_every=4, evaluate_every=4, tensorboard_dir=None, classifier_path=None,
              precision="bf16")
    try:
        cli.train_from_folder(new=True, num_train_steps=6, **kw)
        mdir = tmp_path / "models" / "g"
        assert (mdir / "model_0.pt").exists() and (mdir / "model_1.pt").exists()
        assert json.loads((mdir / ".config.json").read_text())["image_size"] == 64
        pngs = sorted(os.listdir(tmp_path / "results" / "g"))
        assert "1-from_encoder.png" in pngs and "0-from_encoder-ema.png" in pngs
        # resume from the checkpoint: two more steps, losses finite and readable (lazy properties)
        tr = st.Trainer(name="g", results_dir=str(tmp_path / "results"), models_dir=str(tmp_path / "models"),
                        image_size=64, network_capacity=8, fmap_max=128, batch_size=4, gradient_accumulate_every=2,
                        num_workers=0, save_every=4, evaluate_every=10 ** 6, tensorboard_dir=None,
                        classifier_path=None, device=torch.device("cuda:0"))
        tr.load(-1)
        assert tr.steps == 4  # checkpoint number x save_every (reference :1706)
        tr.save = lambda *a, **k: None
        tr.set_data_src(str(data))
        for _ in range(2):
            tr.train()
        vals = [tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss]
        assert all(isinstance(v, float) and math.isfinite(v) for v in vals), vals
        tr.print_log()
        # generator forward in eval through the public module API
        m = tr.StylEx
        m.eval()
        with torch.no_grad():
            w = st.styles_def_to_tensor(st.latent_to_w(m.S, st.noise_list(2, m.G.num_layers, m.G.latent_dim, tr.device)))
            img, coords = m.G(w, st.image_noise(2, 64, tr.device), get_style_coords=True)
        assert img.shape == (2, 3, 64, 64) and img.dtype == torch.float32 and torch.isfinite(img).all()
        assert coords.shape[0] == 2
    finally:
        ops.set_precision("fp32")
