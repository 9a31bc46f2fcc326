"""TEST INFRASTRUCTURE — build container only (needs /root/reference).

Golden vectors for the AttFind StyleSpace sweep: the extraction cell of the reference's
``stylex/run_attfind_combined.ipynb`` (cell 5: ``attfind_extraction`` and helpers) is executed AS IS — its source is
read from the notebook at run time, never copied — on the reference's own ``StylEx`` (``stylex_train.py``, the "old
architecture" branch), on CPU, with an in-memory h5py stand-in that captures the datasets it writes.

    python oracle/make_golden_attfind.py        ->  tests/golden/attfind_16.npz
"""
import json
import math
import multiprocessing
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from make_golden import save, seed_all  # noqa: E402

NOTEBOOK = os.path.join(ref_shim.REF_STYLEX, "run_attfind_combined.ipynb")


class _FakeDataset:
    def __init__(self, store, name, shape):
        self.arr = np.zeros(shape, dtype=np.float32)
        store[name] = self.arr

    def __setitem__(self, idx, value):
        self.arr[idx] = value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value)


class _FakeFile:
    captured = {}

    def __init__(self, path, mode="r"):
        self.path = path

    def create_dataset(self, name, shape, dtype="f"):
        return _FakeDataset(_FakeFile.captured, name, shape)

    def close(self):
        pass


def load_extraction_cell(st):
    """Namespace with the notebook's extraction functions, bound to the reference module `st`."""
    nb = json.load(open(NOTEBOOK))
    src = "".join(nb["cells"][5]["source"])
    assert "def attfind_extraction(" in src
    tq = types.SimpleNamespace(tqdm=lambda it, *a, **k: it)
    ns = dict(torch=torch, np=np, F=F, os=os, math=math, multiprocessing=multiprocessing, tqdm=tq,
              h5py=types.SimpleNamespace(File=_FakeFile), USE_OLD_ARCHITECTURE=True,
              styles_def_to_tensor=st.styles_def_to_tensor, cycle=st.cycle, default=st.default, Dataset=st.Dataset,
              DistributedSampler=None, MNIST_1vA=None, DataLoader=None, make_grid=None, Image=None)
    exec(compile(src, NOTEBOOK + "#cell5", "exec"), ns)
    return ns


def main():
    st = ref_shim.import_reference()
    ns = load_extraction_cell(st)
    size, cap, fmax, seed, n_img, shift = 16, 4, 64, 5, 3, 1.0
    seed_all(seed)
    model = st.StylEx(image_size=size, network_capacity=cap, fmap_max=fmax)
    model.eval()
    # non-zero noise weights so that the fixed noise image matters (they are zero-initialised, :979-983)
    g = torch.Generator().manual_seed(seed + 1)
    for blk in model.G.blocks:
        for lin in (blk.to_noise1, blk.to_noise2):
            lin.weight.data = torch.randn(lin.weight.shape, generator=g) * 0.3
            lin.bias.data = torch.randn(lin.bias.shape, generator=g) * 0.1
    clf = ref_shim.TinyClassifier(seed=99, image_size=size)
    images = [torch.rand(1, 3, size, size, generator=g) for _ in range(n_img)]
    noise = torch.rand(1, size, size, 1, generator=g)
    n_coords = sum(b.num_style_coords for b in model.G.blocks)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    _FakeFile.captured = {}
    ns["attfind_extraction"](dataloader=list(images), num_images=n_img, results_folder="/tmp", stylex=model,
                             classifier=clf, dataset_name=None, noise=noise, num_style_coords=n_coords,
                             shift_size=shift, discriminator_threshold=-0.5, image_size=size, batch_size=1,
                             cuda_rank=0, use_discriminator=False)
    out = dict(_FakeFile.captured)
    # the sweep mutates to_style biases in place and restores them: the model must be unchanged
    for k, v in model.state_dict().items():
        assert torch.allclose(v, state[k], atol=1e-6), k
    save("attfind_16", config=np.array([size, cap, fmax]), seed=seed, shift_size=shift, n_coords=n_coords,
         images=torch.cat(images), input_noise=noise,
         noise_weights=torch.cat([torch.cat([b.to_noise1.weight.reshape(-1), b.to_noise1.bias, b.to_noise2.weight.reshape(-1),
                                             b.to_noise2.bias]) for b in model.G.blocks]),
         **{"out/" + k: v for k, v in out.items()})
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def main_select():
    """Golden for the greedy style selection (notebook cell 15, ``find_significant_styles``), fed with the
    ``style_change`` array of attfind_16.npz.  python oracle/make_golden_attfind.py select"""
    nb = json.load(open(NOTEBOOK))
    src = "".join(nb["cells"][15]["source"])
    assert "def find_significant_styles(" in src
    ns = dict(np=np)
    exec(compile(src, NOTEBOOK + "#cell15", "exec"), ns)
    g = np.load(os.path.join(os.path.dirname(HERE), "tests", "golden", "attfind_16.npz"))
    effect = g["out/style_change"]
    out = {}
    for cls in (0, 1):
        for thr in (0.2, 0.05):
            sel = ns["find_significant_styles"](effect.copy(), 6, cls, None, None, None, None, None,
                                                max_image_effect=thr, sindex_offset=0)
            out["sel/c%d_t%g" % (cls, thr)] = np.array(sel, dtype=np.int64)
    save("attfind_select_16", **out)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "select":
    main_select()
