"""DiffAugment for the discriminator input — same entry point, augmentation names and random-number
consumption as the reference's ``stylex/diff_augment.py`` (DiffAugment :7-11, AUGMENT_FNS :99-113), re-expressed
for device execution:

* the random PARAMETERS of an augmentation (one scalar / integer pair per image) are drawn exactly like the
  reference draws them — same calls, same order, same shapes — but always from the CPU generator and then uploaded
  (a few bytes).  On a CPU run that is literally what the reference does (``device=x.device``), so fixtures captured
  from it are reproduced bit for bit; on the GPU the draw does not depend on a device RNG stream, like the rest of
  the Trainer's inputs (stylex_train.noise / image_noise).
* the image arithmetic runs on the tensor's device without the reference's [B,H,W] advanced-index grids:
  translation is two masked gathers along H and W (zero fill outside), cutout a comparison mask.
"""
import random
from functools import partial

import torch


def DiffAugment(x, types=[]):
    for p in types:
        for f in AUGMENT_FNS[p]:
            x = f(x)
    return x.contiguous()


def _rand(b, x):
    """torch.rand(B,1,1,1) of the reference, CPU generator, on x's device/dtype."""
    return torch.rand(b, 1, 1, 1, dtype=torch.float32).to(device=x.device, dtype=x.dtype)


def rand_brightness(x, scale):  # :22-24
    return x + (_rand(x.size(0), x) - 0.5) * scale


def rand_saturation(x, scale):  # :26-29
    x_mean = x.mean(dim=1, keepdim=True)
    return (x - x_mean) * (((_rand(x.size(0), x) - 0.5) * 2.0 * scale) + 1.0) + x_mean


def rand_contrast(x, scale):  # :31-34
    x_mean = x.mean(dim=[1, 2, 3], keepdim=True)
    return (x - x_mean) * (((_rand(x.size(0), x) - 0.5) * 2.0 * scale) + 1.0) + x_mean


def rand_translation(x, ratio=0.125):
    """:36-49 — out[b,:,i,j] = x[b,:,i+tx,j+ty] where that exists, else 0 (the reference gathers from a zero-padded
    copy with clamped indices: rows 0 and H+1 of the padded image are the zero border)."""
    b, _, h, w = x.shape
    shift_x, shift_y = int(h * ratio + 0.5), int(w * ratio + 0.5)
    tx = torch.randint(-shift_x, shift_x + 1, size=[b, 1, 1]).to(x.device)
    ty = torch.randint(-shift_y, shift_y + 1, size=[b, 1, 1]).to(x.device)
    src_h = torch.arange(h, device=x.device).view(1, h, 1) + tx  # [B,H,1]
    src_w = torch.arange(w, device=x.device).view(1, 1, w) + ty  # [B,1,W]
    ok = ((src_h >= 0) & (src_h < h)) & ((src_w >= 0) & (src_w < w))  # [B,H,W]
    rows = x.gather(2, src_h.clamp(0, h - 1).view(b, 1, h, 1).expand(-1, x.size(1), -1, w))
    out = rows.gather(3, src_w.clamp(0, w - 1).view(b, 1, 1, w).expand(-1, x.size(1), h, -1))
    return out * ok.unsqueeze(1).to(x.dtype)


def rand_offset(x, ratio=1, ratio_h=1, ratio_v=1):
    """:51-71 — a circular shift of every image by its own random (horizontal, vertical) amount.  The amounts are
    drawn exactly as the reference draws them (Python `random.randint`, per image, horizontal first; note its naming:
    the horizontal range is derived from size(2) and applied along the last axis), then ALL images are shifted by one
    batched gather per axis on the tensor's device — out[b, :, i, j] = x[b, :, (i - v_b) mod H, (j - h_b) mod W] —
    instead of the reference's per-image torch.roll + stack."""
    b, _, hh, ww = x.shape
    max_h, max_v = int(hh * ratio * ratio_h), int(ww * ratio * ratio_v)
    shifts = [(random.randint(0, max_h) * 2 - max_h, random.randint(0, max_v) * 2 - max_v) for _ in range(b)]
    sh = torch.tensor([s[0] for s in shifts], dtype=torch.long).to(x.device).view(b, 1)
    sv = torch.tensor([s[1] for s in shifts], dtype=torch.long).to(x.device).view(b, 1)
    src_w = (torch.arange(ww, device=x.device).view(1, ww) - sh) % ww  # [B, W]
    src_h = (torch.arange(hh, device=x.device).view(1, hh) - sv) % hh  # [B, H]
    out = x.gather(3, src_w.view(b, 1, 1, ww).expand(-1, x.size(1), hh, -1))
    return out.gather(2, src_h.view(b, 1, hh, 1).expand(-1, x.size(1), -1, ww))


def rand_offset_h(x, ratio=1):
    return rand_offset(x, ratio=1, ratio_h=ratio, ratio_v=0)


def rand_offset_v(x, ratio=1):
    return rand_offset(x, ratio=1, ratio_h=0, ratio_v=ratio)


def rand_cutout(x, ratio=0.5):
    """:79-92 — zero a cutout_size window centred at (offset - size//2 ...), clamped to the image: the reference
    scatters zeros at clamp(grid + offset - size//2, 0, H-1); the set of rows hit is the contiguous clamped range
    [max(0, ox - s//2), min(H-1, ox - s//2 + s - 1)] (clamping only ever maps onto rows 0 / H-1, which the range
    then contains)."""
    b, _, h, w = x.shape
    ch, cw = int(h * ratio + 0.5), int(w * ratio + 0.5)
    ox = torch.randint(0, h + (1 - ch % 2), size=[b, 1, 1]).to(x.device)
    oy = torch.randint(0, w + (1 - cw % 2), size=[b, 1, 1]).to(x.device)
    if ch == 0 or cw == 0:
        return x * 1
    lo_h, hi_h = (ox - ch // 2).clamp(0, h - 1), (ox - ch // 2 + ch - 1).clamp(0, h - 1)
    lo_w, hi_w = (oy - cw // 2).clamp(0, w - 1), (oy - cw // 2 + cw - 1).clamp(0, w - 1)
    ih = torch.arange(h, device=x.device).view(1, h, 1)
    iw = torch.arange(w, device=x.device).view(1, 1, w)
    hole = ((ih >= lo_h) & (ih <= hi_h)) & ((iw >= lo_w) & (iw <= hi_w))  # [B,H,W]
    return x * (~hole).unsqueeze(1).to(x.dtype)


AUGMENT_FNS = {
    "brightness": [partial(rand_brightness, scale=1.)],
    "lightbrightness": [partial(rand_brightness, scale=.65)],
    "contrast": [partial(rand_contrast, scale=.5)],
    "lightcontrast": [partial(rand_contrast, scale=.25)],
    "saturation": [partial(rand_saturation, scale=1.)],
    "lightsaturation": [partial(rand_saturation, scale=.5)],
    "color": [partial(rand_brightness, scale=1.), partial(rand_saturation, scale=1.), partial(rand_contrast, scale=0.5)],
    "lightcolor": [partial(rand_brightness, scale=0.65), partial(rand_saturation, scale=.5),
                   partial(rand_contrast, scale=0.5)],
    "offset": [rand_offset],
    "offset_h": [rand_offset_h],
    "offset_v": [rand_offset_v],
    "translation": [rand_translation],
    "cutout": [rand_cutout],
}
