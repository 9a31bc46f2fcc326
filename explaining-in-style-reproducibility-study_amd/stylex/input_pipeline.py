"""Device-side input pipeline for ``Trainer.set_data_src`` — opt-in with
``Trainer(device_pipeline=True)`` / ``STYLEX_DEVICE_PIPELINE=1``.

The reference's ``Dataset`` (stylex/stylex_train.py:520-556) decodes, resizes (shorter side -> image_size, antialiased
bilinear), centre-crops and converts every image on the host, inside DataLoader workers, and the training loop then
uploads float32 batches (32 x 3 x 256 x 256 x 4 B = 25 MB per micro-step) with a blocking ``.cuda()``.  Once the step
is tens of milliseconds that is the first non-kernel bottleneck.  Here:

* workers only DECODE (``RawImageFolder``: PIL -> uint8 HWC tensor, no resampling, 4x fewer bytes than float32);
* a prefetch thread (``Prefetcher``) pins the decoded images, uploads them on its own HIP stream and keeps ``depth``
  preprocessed batches ready, so ``next(loader)`` in ``train()`` never waits for PCIe or the host;
* resize / centre-crop / [0,1] scaling run on the GPU (``DevicePreprocessor``): images already at the training
  resolution (FFHQ-256 resized, the benchmark config) are bit-identical to the host path, others agree to the uint8
  rounding the host path applies after its resize (<= 1.5/255, tests/test_input_pipeline.py).

The default (host) pipeline stays the parity path: fixtures and step goldens never go through this file.
"""
import queue
import threading
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils import data

EXTS = ["jpg", "jpeg", "png"]


class RawImageFolder(data.Dataset):
    """Decode only: returns the image as a uint8 [H, W, C] tensor (C = 3, or 4 with transparent=True)."""

    def __init__(self, folder, image_size, transparent=False, aug_prob=0.):
        super().__init__()
        self.folder, self.image_size, self.transparent = folder, image_size, transparent
        self.paths = [p for ext in EXTS for p in Path(f"{folder}").glob(f"**/*.{ext}")]
        assert len(self.paths) > 0, f"No images were found in {folder} for training"

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        from PIL import Image

        img = Image.open(self.paths[index]).convert("RGBA" if self.transparent else "RGB")
        return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())


def collate_raw(items):
    return list(items)  # images of a batch may differ in size until the device has resized them


def target_geometry(h, w, s):
    """(resized_h, resized_w, top, left) of Resize(s) + CenterCrop(s) — the arithmetic of stylex_train.Dataset
    (= torchvision 0.11.1 on PIL images: shorter side -> s, longer side int(s * long / short) TRUNCATED, an image whose
    shorter side already is s untouched; crop offsets int(round((side - s) / 2.0)), round-half-to-even)."""
    short, long = (w, h) if w <= h else (h, w)
    if short == s:
        rw, rh = w, h
    else:
        new_long = int(s * long / short)
        rw, rh = (s, new_long) if w <= h else (new_long, s)
    return rh, rw, int(round((rh - s) / 2.0)), int(round((rw - s) / 2.0))


class DevicePreprocessor:
    """uint8 HWC host images -> float [B, C, S, S] in [0, 1] on `device` (resize, centre crop, scaling on the GPU)."""

    def __init__(self, image_size, device):
        self.s, self.device = image_size, device
        # value/255 for the 256 byte values, computed on the HOST in fp32 exactly as the host pipeline does: a device
        # division by a scalar is a multiplication by the rounded reciprocal (1 ulp off for some values)
        self.lut = (torch.arange(256, dtype=torch.float32) / 255.0).to(device)

    def __call__(self, images):
        s, out = self.s, []
        cuda = self.device.type == "cuda"
        staged = []
        for im in images:
            if cuda:
                im = im.pin_memory()
            staged.append(im.to(self.device, non_blocking=True))
        for im in staged:
            h, w = im.shape[0], im.shape[1]
            rh, rw, top, left = target_geometry(h, w, s)
            if (rh, rw) == (h, w):  # no resampling: table lookup, bit-identical to the host path
                x = self.lut[im[top:top + s, left:left + s].long()].permute(2, 0, 1).unsqueeze(0)
            else:
                x = F.interpolate(im.permute(2, 0, 1).unsqueeze(0).float(), size=(rh, rw), mode="bilinear",
                                  align_corners=False, antialias=True)
                x = x[:, :, top:top + s, left:left + s] / 255.0
            out.append(x)
        return torch.cat(out, dim=0).contiguous()


class Prefetcher:
    """Iterator over preprocessed device batches, produced `depth` ahead by a background thread on its own HIP stream.
    `source` is any iterator of host batches (lists of uint8 images, or ready tensors), `prepare` turns one into a
    device tensor.  next() hands the batch to the caller's current stream (event wait + record_stream)."""

    def __init__(self, source, prepare, device, depth=3):
        self.source, self.prepare, self.device = source, prepare, device
        self.q = queue.Queue(maxsize=depth)
        self.stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        self._stop = False
        self.thread = threading.Thread(target=self._run, daemon=True, name="stylex-prefetch")
        self.thread.start()

    def _run(self):
        try:
            if self.stream is not None:
                torch.cuda.set_device(self.device)
            for host in self.source:
                if self._stop:
                    return
                if self.stream is not None:
                    with torch.cuda.stream(self.stream):
                        batch = self.prepare(host)
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                else:
                    batch, ev = self.prepare(host), None
                self.q.put((batch, ev))
            self.q.put((None, None))
        except BaseException as e:  # noqa: BLE001 — surfaced on the consumer side
            self.q.put((e, None))

    def __iter__(self):
        return self

    def __next__(self):
        batch, ev = self.q.get()
        if batch is None:
            raise StopIteration
        if isinstance(batch, BaseException):
            raise batch
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            batch.record_stream(cur)
        return batch

    def close(self):
        self._stop = True
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass


def cycle(iterable):
    while True:
        for i in iterable:
            yield i


def make_device_loader(folder, image_size, batch_size, device, num_workers=0, transparent=False, sampler=None,
                       shuffle=True, depth=3):
    """DataLoader over decoded images -> Prefetcher of preprocessed device batches; returns (iterator, dataset).
    `folder` may be an already built RawImageFolder (the Trainer sizes its DistributedSampler from it first)."""
    ds = folder if isinstance(folder, RawImageFolder) else RawImageFolder(folder, image_size, transparent=transparent)
    loader = data.DataLoader(ds, num_workers=num_workers, batch_size=batch_size, sampler=sampler,
                             shuffle=shuffle and sampler is None, drop_last=True, collate_fn=collate_raw)
    return Prefetcher(cycle(loader), DevicePreprocessor(image_size, device), device, depth=depth), ds
