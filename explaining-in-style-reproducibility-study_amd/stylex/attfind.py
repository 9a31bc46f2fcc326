"""Batched AttFind StyleSpace sweep — drop-in for ``attfind_extraction`` of the reference's
``stylex/run_attfind_combined.ipynb`` (cell 5, :246-417; "old architecture" branch), producing the same datasets
(``style_change, latents, base_prob, minima, maxima, style_coordinates, original_images, noise, discriminator``).

The notebook perturbs ONE style coordinate at a time by mutating ``to_style{1,2}.bias`` in place and re-running
the whole generator at batch 1: 2 x N_coords generator + classifier evaluations per image (2 x 2464 at 64 px).
Here the same arithmetic is restructured for the GPU:

* a perturbation is a per-sample additive offset on the block's style vector (``GeneratorBlock.forward_main(...,
  styles=)``) — no parameter is mutated, so perturbations of different coordinates batch together;
* a coordinate of block k only changes blocks k..L-1: the feature map and RGB entering block k are computed once
  per image and shared by all perturbations of that block (prefix caching);
* ``chunk`` perturbations (both directions of chunk/2 coordinates) run as one generator-suffix + classifier pass;
* under ``torch.distributed`` the sweep shards over images: every rank runs the cheap first pass on
  all images (1 evaluation each, so minima / maxima need no collective), sweeps images ``rank::world`` and one
  ``all_reduce(SUM)`` over the effect tensor (zero for foreign images) assembles the result on every rank.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

DATASETS = ("style_change", "latents", "base_prob", "minima", "maxima", "style_coordinates", "original_images",
            "noise", "discriminator")


def styles_def_to_tensor(styles_def):
    return torch.cat([t[:, None, :].expand(-1, n, -1) for t, n in styles_def], dim=1)


def _prefix_states(G, w_tensor, noise):
    """Feature map / RGB entering every block for one image (batch 1), plus the unperturbed block styles."""
    x = G.initial_conv(G.initial_block.expand(1, -1, -1, -1))
    rgb, states = None, []
    for li, block in enumerate(G.blocks):
        istyle = w_tensor[:, li]
        s1, s2 = block.to_style1(istyle), block.to_style2(istyle)
        states.append((x, rgb, s1, s2))
        x, _ = block.forward_main(x, istyle, noise, styles=(s1, s2))
        rgb = block.to_rgb(x, rgb, istyle)
    return states


def _suffix(G, k, x, rgb, w_tensor, noise, styles_k):
    """Blocks k..L-1 for a batch of perturbations of block k (styles_k = per-sample (style1, style2))."""
    p = styles_k[0].shape[0]
    w = w_tensor.expand(p, -1, -1)
    nz = noise.expand(p, -1, -1, -1)
    x = x.expand(p, -1, -1, -1)
    rgb = None if rgb is None else rgb.expand(p, -1, -1, -1)
    for li in range(k, len(G.blocks)):
        block = G.blocks[li]
        x, _ = block.forward_main(x, w[:, li], nz, styles=styles_k if li == k else None)
        rgb = block.to_rgb(x, rgb, w[:, li])
    return rgb.float()


@torch.no_grad()
def attfind_extraction(stylex, classifier, images, num_images, noise, shift_size=1.0, discriminator_threshold=None,
                       use_discriminator=False, chunk=256, results_folder=None):
    """images: iterable of [1,3,S,S] batches (the notebook's batch-size-1 loader).  Returns a dict of CPU tensors
    with the notebook's dataset names; with `results_folder` also writes style_change_records.{hdf5|npz}."""
    G = stylex.G
    dev = next(G.parameters()).device
    noise = noise.to(dev)
    n_coords = sum(b.num_style_coords for b in G.blocks)
    latents, base_logits, coords, disc, originals = [], [], [], [], []
    for batch in images:
        if len(latents) >= num_images:
            break
        batch = batch.to(dev)
        enc = stylex.encoder(batch).reshape(1, -1)
        w = torch.cat((enc, classifier.classify_images(batch)), dim=1)
        generated, sc = G(styles_def_to_tensor([(w, G.num_layers)]), noise, get_style_coords=True)
        d_out = stylex.D(generated).reshape(1)
        if use_discriminator and discriminator_threshold is not None and float(d_out) < discriminator_threshold:
            continue
        originals.append(batch[0])
        latents.append(w[0])
        coords.append(sc[0])
        disc.append(d_out)
        base_logits.append(classifier.classify_images(generated)[0])
    if not latents:
        raise ValueError("No images pass the threshold check")
    n = len(latents)
    coords = torch.stack(coords)
    minima, maxima = coords.min(dim=0)[0], coords.max(dim=0)[0]
    effects = torch.zeros(n, 2, n_coords, 2, device=dev)
    half = max(1, chunk // 2)
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if sharded else (0, 1)
    for i in range(rank, n, world):
        w_tensor = styles_def_to_tensor([(latents[i].unsqueeze(0), G.num_layers)])
        states = _prefix_states(G, w_tensor, noise)
        base = 0
        for k, block in enumerate(G.blocks):
            x_k, rgb_k, s1, s2 = states[k]
            c1 = block.input_channels
            for lo in range(0, block.num_style_coords, half):
                idx = torch.arange(lo, min(lo + half, block.num_style_coords), device=dev)
                sidx = base + idx
                cur = coords[i, sidx]
                # rows: [down for every coordinate of the chunk, then up]
                delta = torch.cat(((minima[sidx] - cur) * shift_size, (maxima[sidx] - cur) * shift_size))
                cols = torch.cat((idx, idx))
                rows = torch.arange(cols.numel(), device=dev)
                st1 = s1.expand(cols.numel(), -1).clone()
                st2 = s2.expand(cols.numel(), -1).clone()
                in1 = cols < c1
                st1[rows[in1], cols[in1]] += delta[in1]
                st2[rows[~in1], cols[~in1] - c1] += delta[~in1]
                logits = classifier.classify_images(_suffix(G, k, x_k, rgb_k, w_tensor, noise, (st1, st2)))
                diff = logits - base_logits[i][None]
                m = idx.numel()
                effects[i, 0, sidx] = diff[:m]
                effects[i, 1, sidx] = diff[m:]
            base += block.num_style_coords
    if sharded:
        dist.all_reduce(effects, op=dist.ReduceOp.SUM)
    out = {"style_change": effects, "latents": torch.stack(latents), "base_prob": torch.stack(base_logits),
           "minima": minima[None], "maxima": maxima[None], "style_coordinates": coords,
           "original_images": torch.stack(originals), "noise": noise, "discriminator": torch.stack(disc)}
    out = {k: v.detach().float().cpu() for k, v in out.items()}
    if results_folder is not None and rank == 0:
        write_records(out, results_folder)
    return out


def write_records(out, results_folder):
    """style_change_records.hdf5 with the notebook's dataset names (:392-416); .npz when h5py is not installed."""
    try:
        import h5py
    except ImportError:
        np.savez(os.path.join(results_folder, "style_change_records.npz"), **{k: v.numpy() for k, v in out.items()})
        return
    with h5py.File(os.path.join(results_folder, "style_change_records.hdf5"), "w") as f:
        for k in DATASETS:
            f.create_dataset(k, data=out[k].numpy(), dtype="f")


def find_significant_styles(style_change_effect, num_indices, class_index, max_image_effect=0.2, sindex_offset=0):
    """Greedy selection of the style coordinates that most raise `class_index` (notebook cell 15): repeatedly take the
    coordinate/direction with the largest mean positive effect over the images not yet explained (accumulated effect
    < max_image_effect).  style_change_effect: [images, 2 directions, coords, classes].  Returns
    [(direction, coordinate + sindex_offset)].  (The notebook's generator / classifier / latent arguments are unused.)"""
    effect = np.array(style_change_effect, copy=True)
    n_img, _, n_coords, _ = effect.shape
    direction = np.maximum(0, effect[:, :, :, class_index].reshape((n_img, -1)))
    images_effect = np.zeros(n_img)
    chosen = []
    while len(chosen) < num_indices:
        nxt = int(np.argmax(np.mean(direction[images_effect < max_image_effect], axis=0)))
        chosen.append(nxt)
        images_effect += direction[:, nxt]
        direction[:, nxt] = 0
    return [(x // n_coords, (x % n_coords) + sindex_offset) for x in chosen]


# ---- post-processing and visualisation cells of the notebook (cells 11, 12, 14, 17-21) ------------------------------


def filter_unstable_images(style_change_effect, effect_threshold=0.3, num_indices_threshold=150):
    """Cell 11: zero the effects of images on which more than `num_indices_threshold` coordinates move the classifier by
    more than `effect_threshold` (in place, like the notebook)."""
    unstable = np.sum(np.abs(style_change_effect) > effect_threshold, axis=(1, 2, 3)) > num_indices_threshold
    style_change_effect[unstable] = 0
    return style_change_effect


def style_vector_distances(all_style_vectors, style_min, style_max):
    """Cell 12 (tail): [images, coords, 2] = distance of every style coordinate to the minimum / to the maximum."""
    d = np.zeros((all_style_vectors.shape[0], all_style_vectors.shape[1], 2))
    d[:, :, 0] = all_style_vectors - style_min[None]
    d[:, :, 1] = style_max[None] - all_style_vectors
    return d


def split_by_class(base_probs, style_change_effect, W_values, all_style_vectors_distances, all_style_vectors):
    """Cell 14: the sweep's arrays split by the class the classifier assigns to the generated image."""
    labels = np.argmax(base_probs, axis=1)
    out = {}
    for c in range(2):
        idx = np.nonzero(labels == c)[0]
        out[c] = dict(effect=style_change_effect[idx].astype(np.float64), w=W_values[idx].astype(np.float64),
                      dist=all_style_vectors_distances[idx], style_vectors=all_style_vectors[idx], index=idx)
    return out


def _block_of(G, sindex):
    base = 0
    for k, block in enumerate(G.blocks):
        if sindex < base + block.num_style_coords:
            return k, sindex - base
        base += block.num_style_coords
    raise IndexError(sindex)


@torch.no_grad()
def change_images(G, classifier, dlatents, sindex, style_direction_index, s_style_min, s_style_max, shift_size, noise,
                  class_index=0):
    """Cells 17 + 19 for a BATCH of latents: the generated image of every latent, the image with style coordinate
    `sindex` moved towards its minimum (direction 0) or maximum (1) by `shift_size` times the distance, and the
    classifier's probability of `class_index` for both.  The notebook mutates ``to_style{1,2}.bias`` and runs the
    generator twice per image; here the shift is a per-sample offset on the block's style vector and all latents run
    as one pass.  Returns (base [n,3,S,S], changed [n,3,S,S], base_prob [n], change_prob [n])."""
    dev = next(G.parameters()).device
    w = torch.as_tensor(np.asarray(dlatents), dtype=torch.float32, device=dev)
    n = w.shape[0]
    noise = torch.as_tensor(noise).to(dev).expand(n, -1, -1, -1)
    k, widx = _block_of(G, int(sindex))
    base_img, coords = G(styles_def_to_tensor([(w, G.num_layers)]), noise, get_style_coords=True)
    target = float(s_style_min) if style_direction_index == 0 else float(s_style_max)
    delta = (target - coords[:, int(sindex)]) * shift_size
    x = G.initial_conv(G.initial_block.expand(n, -1, -1, -1))
    rgb = None
    for li, block in enumerate(G.blocks):
        styles = None
        if li == k:
            s1, s2 = block.to_style1(w), block.to_style2(w)
            if widx < block.input_channels:
                s1 = s1.clone()
                s1[:, widx] += delta
            else:
                s2 = s2.clone()
                s2[:, widx - block.input_channels] += delta
            styles = (s1, s2)
        x, _ = block.forward_main(x, w, noise, styles=styles)
        rgb = block.to_rgb(x, rgb, w)
    changed = rgb.float()
    base_img = base_img.float()
    p0 = torch.softmax(classifier.classify_images(base_img), dim=1)[:, class_index]
    p1 = torch.softmax(classifier.classify_images(changed), dim=1)[:, class_index]
    return base_img, changed, p0.cpu().numpy(), p1.cpu().numpy()


def pair_image(base, changed):
    """Cells 18 + 19: [S, 2S, 3] uint8, generated image left, changed image right (``draw_on_image``: clip to [0, 1],
    times 255, truncated to bytes; the probability text the notebook once drew is commented out there)."""
    def u8(img):
        return (np.clip(np.transpose(img.detach().cpu().numpy(), (1, 2, 0)), 0, 1) * 255).astype(np.uint8)
    return np.concatenate((u8(base), u8(changed)), axis=1)


def visualize_style(G, classifier, all_dlatents, style_change_effect, style_min, style_max, sindex, style_direction_index,
                    max_images, shift_size, noise, class_index=0, effect_threshold=0.3, seed=None,
                    allow_both_directions_change=False):
    """Cell 20: images on which moving coordinate `sindex` in direction `style_direction_index` changed the
    classifier's `class_index` output by more than `effect_threshold` in the sweep, shuffled (numpy global generator,
    seeded like the notebook), up to 10 x `max_images` candidates re-generated, those whose probability moves by at
    least `effect_threshold` kept, the first `max_images` stacked vertically; an empty array if fewer than three."""
    eff = style_change_effect[:, style_direction_index, sindex, class_index]
    idx = (np.abs(eff) > effect_threshold).nonzero()[0] if allow_both_directions_change else (eff > effect_threshold).nonzero()[0]
    if idx.size == 0:
        return np.array([])
    if seed is not None:
        np.random.seed(seed)
    np.random.shuffle(idx)
    idx = idx[:min(max_images * 10, len(idx))]
    base, changed, p0, p1 = change_images(G, classifier, np.asarray(all_dlatents)[idx], sindex, style_direction_index,
                                          style_min[sindex], style_max[sindex], shift_size, noise, class_index)
    rows = [pair_image(base[i], changed[i]) for i in range(len(idx)) if not np.abs(p1[i] - p0[i]) < effect_threshold]
    rows = rows[:max_images]
    return np.concatenate(rows, axis=0) if len(rows) >= 3 else np.array([])


def visualize_style_by_distance_in_s(G, classifier, all_dlatents, all_style_vectors_distances, style_min, style_max, sindex,
                                     style_sign_index, max_images, shift_size, noise, class_index=0):
    """Cell 21: the images whose coordinate `sindex` is farthest from the end it is moved to (largest distance first),
    base | changed pairs of the first `max_images`; an empty array if fewer than three."""
    idx = np.argsort(all_style_vectors_distances[:, sindex, style_sign_index])[::-1]
    if idx.size == 0:
        return np.array([])
    idx = idx[:min(max_images * 10, len(idx))]
    base, changed, _, _ = change_images(G, classifier, np.asarray(all_dlatents)[idx], sindex, style_sign_index,
                                        style_min[sindex], style_max[sindex], shift_size, noise, class_index)
    rows = [pair_image(base[i], changed[i]) for i in range(len(idx))]
    return np.concatenate(rows[:max_images], axis=0) if len(rows) >= 3 else np.array([])
