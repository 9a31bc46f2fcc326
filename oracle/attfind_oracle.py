"""TEST INFRASTRUCTURE — CPU restatement of the reference's AttFind StyleSpace sweep
(stylex/run_attfind_combined.ipynb, cell 5: ``sindex_to_block_idx_and_index`` :194-209,
``get_min_max_style_vectors`` :212-228, ``discriminator_filter`` :231-243, ``attfind_extraction`` :246-417, "old
architecture" branch).  Literal and sequential: one generator evaluation per (image, coordinate, direction), the
perturbation applied by mutating ``to_style{1,2}.bias`` in place exactly as the notebook does.

Pinned against ``tests/golden/attfind_16.npz`` (made by ``oracle/make_golden_attfind.py``, which executes the
notebook cell itself on the reference's StylEx).  Only ``tests/`` may import this file.
"""
import torch


def styles_def_to_tensor(styles_def):  # stylex_train.py:352-353
    return torch.cat([t[:, None, :].expand(-1, n, -1) for t, n in styles_def], dim=1)


def sindex_to_block_idx_and_index(generator, sindex):
    tmp = sindex
    for idx, block in enumerate(generator.blocks):
        if tmp < block.num_style_coords:
            return idx, tmp
        tmp -= block.num_style_coords
    return None, None


@torch.no_grad()
def attfind_extraction(stylex, classifier, images, noise, shift_size=1.0):
    """images: list of [1,3,S,S] tensors.  Returns the datasets the notebook writes to style_change_records.hdf5."""
    G = stylex.G
    n = len(images)
    n_coords = sum(b.num_style_coords for b in G.blocks)
    latents, base_logits, coords, disc, originals = [], [], [], [], []
    for batch in images:
        enc = stylex.encoder(batch).unsqueeze(0)
        logits = classifier.classify_images(batch)
        w = torch.cat((enc, logits), dim=1)
        generated, sc = G(styles_def_to_tensor([(w, G.num_layers)]), noise, get_style_coords=True)
        disc.append(stylex.D(generated).reshape(1))
        originals.append(batch[0])
        latents.append(w[0])
        coords.append(sc[0])
        base_logits.append(classifier.classify_images(generated)[0])
    coords = torch.stack(coords)
    minima, maxima = coords.min(dim=0)[0], coords.max(dim=0)[0]
    effects = torch.zeros(n, 2, n_coords, 2)
    for i in range(n):
        w_tensor = styles_def_to_tensor([(latents[i].unsqueeze(0), G.num_layers)])
        for s in range(n_coords):
            bi, wi = sindex_to_block_idx_and_index(G, s)
            block = G.blocks[bi]
            if wi < block.input_channels:
                layer, width = block.to_style1, block.input_channels
            else:
                wi -= block.input_channels
                layer, width = block.to_style2, block.filters
            one_hot = torch.zeros(width)
            one_hot[wi] = 1
            for d, target in enumerate((minima, maxima)):
                shift = one_hot * ((target[s] - coords[i, s]) * shift_size)
                layer.bias += shift
                logits = classifier.classify_images(G(w_tensor, noise))
                layer.bias -= shift
                effects[i, d, s] = logits[0] - base_logits[i]
    return {"style_change": effects, "latents": torch.stack(latents), "base_prob": torch.stack(base_logits),
            "minima": minima[None], "maxima": maxima[None], "style_coordinates": coords,
            "original_images": torch.stack(originals), "noise": noise, "discriminator": torch.stack(disc)}
