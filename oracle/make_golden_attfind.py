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


def main_visualize():
    """Golden for the notebook's VISUALISATION cells (17, 18, 19, 20, 21: ``generate_change_image_given_dlatent``,
    ``draw_on_image``, ``generate_images_given_dlatent``, ``visualize_style``, ``visualize_style_by_distance_in_s``) and
    the post-processing of cells 11 / 12 / 14 (``filter_unstable_images``, the style-vector distances, the split by
    predicted class): the cells are executed AS THEY ARE (source read from the notebook at run time) on the reference's
    StylEx, with the datasets of attfind_16.npz as their inputs.  python oracle/make_golden_attfind.py visualize"""
    from PIL import Image, ImageDraw

    st = ref_shim.import_reference()
    nb = json.load(open(NOTEBOOK))
    # the visualisation functions hard-wire a 64 px canvas (``resolution = 64`` default of cell 19, not passed on by cells
    # 20 / 21): a 64 px model of tiny capacity, swept by the notebook's own extraction cell first
    size, cap, fmax, seed, n_img = 64, 2, 64, 11, 10
    seed_all(seed)
    model = st.StylEx(image_size=size, network_capacity=cap, fmap_max=fmax)
    model.eval()
    gen = torch.Generator().manual_seed(seed + 1)
    for blk in model.G.blocks:
        for lin in (blk.to_noise1, blk.to_noise2):
            lin.weight.data = torch.randn(lin.weight.shape, generator=gen) * 0.3
            lin.bias.data = torch.randn(lin.bias.shape, generator=gen) * 0.1
    clf = ref_shim.TinyClassifier(seed=99, image_size=size)
    images = [torch.rand(1, 3, size, size, generator=gen) for _ in range(n_img)]
    noise_in = torch.rand(1, size, size, 1, generator=gen)
    n_coords = sum(b.num_style_coords for b in model.G.blocks)
    ns = load_extraction_cell(st)  # also defines sindex_to_block_idx_and_index
    # the stand-in classifier must put generated images into BOTH classes (cell 14 fails on an empty class, as the
    # notebook's own markdown warns): centre its decision boundary on the generated images first
    with torch.no_grad():
        for _ in range(12):  # the logits are part of the latent, so the generated images move with the shift: iterate
            gen_logits = []
            for im in images:
                w = torch.cat((model.encoder(im).reshape(1, -1), clf.classify_images(im)), dim=1)
                gen_logits.append(clf.classify_images(model.G(st.styles_def_to_tensor([(w, model.G.num_layers)]), noise_in))[0])
            gen_logits = torch.stack(gen_logits)
            labels = gen_logits.argmax(dim=1)
            if 3 <= int(labels.sum()) <= n_img - 3:
                break
            clf.b2 = clf.b2 - 0.7 * torch.tensor([float((gen_logits[:, 0] - gen_logits[:, 1]).median()), 0.0])
        assert 3 <= int(labels.sum()) <= n_img - 3, labels
    _FakeFile.captured = {}
    ns["attfind_extraction"](dataloader=list(images), num_images=n_img, results_folder="/tmp", stylex=model,
                             classifier=clf, dataset_name=None, noise=noise_in, num_style_coords=n_coords, shift_size=1.0,
                             discriminator_threshold=-0.5, image_size=size, batch_size=1, cuda_rank=0,
                             use_discriminator=False)
    g = {"out/" + k: v for k, v in _FakeFile.captured.items()}
    print("labels of the generated images:", np.argmax(g["out/base_prob"].reshape(n_img, -1), axis=1))
    fixture_inputs = dict(config=np.array([size, cap, fmax]), model_seed=seed, cls_b2=clf.b2, images=torch.cat(images), input_noise=noise_in,
                          noise_weights=torch.cat([torch.cat([b.to_noise1.weight.reshape(-1), b.to_noise1.bias,
                                                              b.to_noise2.weight.reshape(-1), b.to_noise2.bias])
                                                   for b in model.G.blocks]),
                          **g)
    ns.update(Image=Image, ImageDraw=ImageDraw, ImageFont=types.SimpleNamespace(truetype=lambda *a, **k: None),
              stylex=model, plt=None)
    for ci in (11, 17, 18, 19, 20, 21):
        exec(compile("".join(nb["cells"][ci]["source"]), NOTEBOOK + "#cell%d" % ci, "exec"), ns)
    effect, w_values = g["out/style_change"], g["out/latents"]
    coords, smin, smax = g["out/style_coordinates"], np.squeeze(g["out/minima"]), np.squeeze(g["out/maxima"])
    noise = noise_in
    # cell 12 (tail) and cell 14, executed on the golden's arrays
    cell12 = "".join(nb["cells"][12]["source"])
    tail = cell12[cell12.index("all_style_vectors_distances = np.zeros"):]
    env = dict(np=np, all_style_vectors=coords, style_min=smin, style_max=smax)
    exec(compile(tail, NOTEBOOK + "#cell12tail", "exec"), env)
    dist_s = env["all_style_vectors_distances"]
    env14 = dict(np=np, base_probs=g["out/base_prob"], style_change_effect=effect, W_values=w_values,
                 all_style_vectors_distances=dist_s, all_style_vectors=coords, print=lambda *a, **k: None)
    exec(compile("".join(nb["cells"][14]["source"]), NOTEBOOK + "#cell14", "exec"), env14)
    out = {"dist": dist_s}
    for c in (0, 1):
        out["class%d/effect" % c] = env14["style_effect_classes"][c]
        out["class%d/w" % c] = env14["W_classes"][c]
        out["class%d/dist" % c] = env14["style_vectors_distances_classes"][c]
    out["filtered"] = ns["filter_unstable_images"](effect.copy(), effect_threshold=0.05, num_indices_threshold=20)
    # the coordinate / direction with the largest mean effect on class 0
    mean_eff = effect[:, :, :, 0].mean(axis=0)
    direction, sindex = np.unravel_index(np.argmax(mean_eff), mean_eff.shape)
    out["pick"] = np.array([direction, sindex], dtype=np.int64)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        yy = ns["visualize_style"](model.G, clf, w_values, effect, smin, smax, int(sindex), int(direction), max_images=3,
                                   shift_size=2, font_file="x", noise=noise, label_size=2, class_index=0,
                                   effect_threshold=1e-6, seed=3, draw_results_on_image=True)
        zz = ns["visualize_style_by_distance_in_s"](model.G, clf, w_values, dist_s, smin, smax, int(sindex), int(direction),
                                                    max_images=3, shift_size=2, font_file="x", noise=noise, label_size=2,
                                                    class_index=0, draw_results_on_image=True, cuda_rank=0)
        probs = []
        for i in range(w_values.shape[0]):
            img, change_prob, base_prob = ns["generate_images_given_dlatent"](
                dlatent=w_values[i:i + 1], generator=model.G, classifier=clf, class_index=0, sindex=int(sindex),
                s_style_min=smin[sindex], s_style_max=smax[sindex], style_direction_index=int(direction), font_file="x",
                noise=noise, shift_size=2, label_size=2, draw_results_on_image=True, resolution=size, cuda_rank=0,
                gen_num_layers=model.G.num_layers)
            probs.append([base_prob, change_prob])
            out["pair_%d" % i] = img
    for k, v in model.state_dict().items():
        assert torch.allclose(v, state[k], atol=1e-6), k
    assert yy.size > 0 and zz.size > 0, (yy.shape, zz.shape)
    out["visualize_style"], out["visualize_by_distance"], out["probs"] = yy, zz, np.array(probs)
    out.update({k: v for k, v in fixture_inputs.items() if k not in out})
    save("attfind_visualize_64", shift_size=2, seed=3, **out)
    print({k: tuple(getattr(v, "shape", ())) for k, v in out.items()})


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "visualize":
    main_visualize()
