"""Config 1 plumbing (BASELINE.json configs[0]): the drop-in cli.py entry point end to end on CPU —
image folder -> Dataset/DataLoader -> Trainer.train() -> evaluate PNG grids -> checkpoint + config.
Ops run on the CPU test double (no GPU here)."""
import json
import os

import numpy as np
import pytest
import torch

import cli
import ops
from cpu_ops import CpuOracleOps


@pytest.fixture(autouse=True)
def cpu_double():
    prev = ops.use_impl(CpuOracleOps)
    yield
    ops.use_impl(prev)


def test_flag_parsing_matches_fire_conventions():
    f = cli.parse_flags(["--data", "x/y", "--image_size", "128", "--new", "--aug_types", "[cutout]", "--lr_mlp=0.05",
                         "--nosample_from_encoder", "--name", "run1", "--fmap_max", "64"])
    assert f == {"data": "x/y", "image_size": 128, "new": True, "aug_types": "[cutout]", "lr_mlp": 0.05,
                 "sample_from_encoder": False, "name": "run1", "fmap_max": 64}
    with pytest.raises(TypeError):
        cli.train_from_folder(not_a_flag=1)
    # the reference's defaults (cli.py:84-171)
    d = cli.DEFAULTS
    assert (d["image_size"], d["batch_size"], d["gradient_accumulate_every"], d["ttur_mult"], d["trunc_psi"],
            d["rec_scaling"], d["classifier_name"], d["seed"]) == (64, 4, 8, 1.5, 0.75, 1, "resnet", 42)


def test_train_from_folder_cpu(tmp_path):
    from PIL import Image

    data = tmp_path / "imgs"
    data.mkdir()
    rng = np.random.RandomState(0)
    for i in range(6):
        Image.fromarray(rng.randint(0, 255, (40, 48, 3), dtype=np.uint8)).save(data / f"{i}.png")
    # positional use as python-fire / a direct caller passes them (reference cli.py:84: data, results_dir,
    # models_dir, name, new, ...)
    with pytest.raises(TypeError):
        cli.train_from_folder(str(data), data=str(data))
    cli.train_from_folder(str(data), str(tmp_path / "results"), str(tmp_path / "models"), "t", True, image_size=32, network_capacity=4, fmap_max=64, batch_size=2,
                          gradient_accumulate_every=2, num_train_steps=1, num_workers=0, save_every=1,
                          evaluate_every=1, tensorboard_dir=None, classifier_path=None)
    assert (tmp_path / "models" / "t" / "model_0.pt").exists()
    cfg = json.loads((tmp_path / "models" / "t" / ".config.json").read_text())
    assert cfg["image_size"] == 32 and cfg["network_capacity"] == 4
    pngs = sorted(os.listdir(tmp_path / "results" / "t"))
    assert pngs == ["0-from_encoder-ema.png", "0-from_encoder-mr.png", "0-from_encoder.png"]
    ck = torch.load(tmp_path / "models" / "t" / "model_1.pt") if (tmp_path / "models" / "t" / "model_1.pt").exists() \
        else torch.load(tmp_path / "models" / "t" / "model_0.pt")
    assert ck["version"] == "1.8.7"


def test_set_seed_reseeds_all_three_generators():
    """reference cli.py:35-40, including the cudnn flags (:37-38)."""
    import random

    cli.set_seed(123)
    a = (torch.rand(3), np.random.rand(3), random.random())
    cli.set_seed(123)
    b = (torch.rand(3), np.random.rand(3), random.random())
    assert torch.equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert torch.backends.cudnn.deterministic is True and torch.backends.cudnn.benchmark is False


def test_new_architecture_switch_and_evaluate(tmp_path, monkeypatch):
    """reference cli.py:16-22: USE_OLD_ARCHITECTURE=False binds Trainer to stylex_train_new; its evaluate() writes the
    three grids with class probabilities appended to W (stylex_train_new.py:1585-1661, 1738-1749)."""
    import importlib

    import stylex_train as st
    import stylex_train_new as stn
    from lpips_standin import LPIPSStandIn
    from standins import TinyClassifier

    monkeypatch.setenv("STYLEX_NEW_ARCHITECTURE", "1")
    try:
        mod = importlib.reload(cli)
        assert mod.Trainer is stn.Trainer and not mod.USE_OLD_ARCHITECTURE
    finally:
        monkeypatch.delenv("STYLEX_NEW_ARCHITECTURE")
        importlib.reload(cli)
    assert cli.Trainer is st.Trainer and cli.USE_OLD_ARCHITECTURE
    tr = stn.Trainer(name="e", base_dir=str(tmp_path), image_size=32, network_capacity=2, fmap_max=16, batch_size=2,
                     gradient_accumulate_every=2, classifier=TinyClassifier(seed=1), lpips_fn=LPIPSStandIn(seed=2),
                     classifier_name="resnet", evaluate_every=10 ** 9, save_every=10 ** 9, num_image_tiles=2,
                     device=torch.device("cpu"))
    tr.loader = st.cycle([torch.rand(2, 3, 32, 32) for _ in range(4)])
    tr.init_StylEx()
    assert tr.StylEx.S.net[0].weight.shape == (512, 512) and tr.StylEx.D.fc.out_features == 2
    tr.evaluate(encoder_input=False, num=5)
    tr.evaluate(encoder_input=True, num=6)
    assert sorted(os.listdir(tmp_path / "results" / "e")) == ["5--ema.png", "5--mr.png", "5-.png", "6-from_encoder-ema.png",
                                                               "6-from_encoder-mr.png", "6-from_encoder.png"]


@pytest.mark.timeout(900)
def test_cli_multi_gpus_through_torchrun_gloo(tmp_path):
    """`cli.py --multi_gpus` as the driver launches a multi-GPU job: `python -m torch.distributed.run` with one process
    per rank (here 2 ranks over gloo on the CPU test double; on the GPU box the same code path picks "nccl" = RCCL).
    Both ranks run the same number of train() calls with the per-rank batch, all-reduce their gradients, and only
    rank 0 writes the checkpoint / config (reference cli.py:49-51, stylex_train.py:1188-1193)."""
    import socket
    import subprocess
    import sys

    from PIL import Image

    data = tmp_path / "imgs"
    data.mkdir()
    rng = np.random.RandomState(1)
    for i in range(8):
        Image.fromarray(rng.randint(0, 255, (32, 32, 3), dtype=np.uint8)).save(data / f"{i}.png")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    launcher = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cli_cpu_double.py")
    env = dict(os.environ, OMP_NUM_THREADS="2", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), launcher, str(data), "--results_dir", str(tmp_path / "results"), "--models_dir",
           str(tmp_path / "models"), "--name", "ddp", "--new", "--multi_gpus", "--image_size", "16", "--network_capacity", "2",
           "--fmap_max", "16", "--batch_size", "4", "--gradient_accumulate_every", "1", "--num_train_steps", "2",
           "--num_workers", "0", "--save_every", "2", "--evaluate_every", "1000000", "--tensorboard_dir", "None",
           "--classifier_path", "None"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "1/2 process initialized." in out.stdout and "2/2 process initialized." in out.stdout
    mdir = tmp_path / "models" / "ddp"
    cks = sorted(f for f in os.listdir(mdir) if f.startswith("model_"))
    assert cks, os.listdir(mdir)
    ck = torch.load(mdir / cks[-1])
    assert ck["version"] == "1.8.7"
    assert all(torch.isfinite(v).all() for v in ck["StylEx"].values() if torch.is_floating_point(v))
