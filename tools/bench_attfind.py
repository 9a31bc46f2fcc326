"""AttFind sweep throughput (BASELINE config 5): coordinate evaluations per second at 64 px on one MI355X —
the batched, prefix-cached engine (attfind.py) vs the notebook's procedure (one batch-1 generator + classifier
evaluation per coordinate and direction, bias mutated in place) on the same HIP modules.
Usage (GPU box): python tools/bench_attfind.py [--images 2] [--chunk 512] [--precision bf16]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402

import attfind  # noqa: E402
import hip_backend as hb  # noqa: E402
import ops  # noqa: E402
import stylex_train as st  # noqa: E402
from resnet_classifier import ResNet  # noqa: E402


@torch.no_grad()
def notebook_style(m, clf, w_tensor, noise, n_eval):
    """The reference procedure for the first n_eval coordinates of one image (both directions)."""
    G, done = m.G, 0
    for block in G.blocks:
        for layer, width in ((block.to_style1, block.input_channels), (block.to_style2, block.filters)):
            for j in range(width):
                if done >= n_eval:
                    return done
                for sign in (-0.5, 0.5):
                    layer.bias[j] += sign
                    clf.classify_images(G(w_tensor, noise))
                    layer.bias[j] -= sign
                done += 1
    return done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=512)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    hb.load_library()
    ops.set_precision(a.precision)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = st.StylEx(a.size, rank=0).eval()
    clf = ResNet(None, 0, output_size=2, image_size=a.size)
    n_coords = sum(b.num_style_coords for b in m.G.blocks)
    imgs = [torch.rand(1, 3, a.size, a.size, device=dev) for _ in range(a.images)]
    noise = st.image_noise(1, a.size, dev)
    attfind.attfind_extraction(m, clf, imgs[:1], 1, noise, chunk=a.chunk)  # warm-up
    torch.cuda.synchronize()
    t = time.perf_counter()
    attfind.attfind_extraction(m, clf, imgs, a.images, noise, chunk=a.chunk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    evals = a.images * n_coords * 2
    w = torch.cat((m.encoder(imgs[0]).reshape(1, -1), clf.classify_images(imgs[0])), dim=1)
    w_tensor = attfind.styles_def_to_tensor([(w, m.G.num_layers)])
    notebook_style(m, clf, w_tensor, noise, 8)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = notebook_style(m, clf, w_tensor, noise, 100)
    torch.cuda.synchronize()
    dt_nb = time.perf_counter() - t
    print({"size": a.size, "precision": a.precision, "style_coords": n_coords, "images": a.images, "chunk": a.chunk,
           "batched_generator_evals_per_s": round(evals / dt, 1), "seconds_per_image": round(dt / a.images, 3),
           "notebook_procedure_evals_per_s": round(2 * n / dt_nb, 1),
           "reference_notebook_progress_bar_it_per_s": "49-50 it/s (=98-100 evals/s, unknown GPU; BASELINE.md)"})


if __name__ == "__main__":
    main()
