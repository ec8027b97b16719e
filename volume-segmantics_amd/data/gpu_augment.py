"""Device side of the training augmentations (csrc/augment.hip): the host draws one sample_params() per slice, packs them into the
small tables vs_augment_batch takes, and the whole batch is augmented and normalised in HBM - the reference's per-sample
albumentations calls in 4 DataLoader workers (data/datasets.py:42-60, utilities/config.py:33) cost ~50-300 ms per slice in
NumPy, against a training step of a few milliseconds for 32 slices."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import augmentations as A


def pack_params(params: list, size: int):
    """(ctypes array of vs_aug_params, intensity LUTs (n, 256) uint8, grid coordinate tables (n, 2, size) float32)."""
    from .. import _lib
    n = len(params)
    arr = (_lib.AugParams * n)()
    luts = np.empty((n, 256), np.uint8)
    tables = np.zeros((n, 2, size), np.float32)
    for i, p in enumerate(params):
        a = arr[i]
        if p["crop"] is not None:
            ch, cw, hs, ws = p["crop"]
            ch, cw = min(ch, size), min(cw, size)
            a.crop, a.ch, a.cw, a.y1, a.x1 = 1, ch, cw, int((size - ch) * hs), int((size - cw) * ws)
        a.flip_v, a.rot_k, a.transpose = int(p["flip_v"]), int(p["rot_k"]), int(p["transpose"])
        d = p["distort"]
        if d is not None:
            if d[0] == "elastic":
                a.distort = 1
                inv = A.elastic_inverse_affine(size, size, d[1])
                for k in range(6):
                    a.inv_affine[k] = float(inv.reshape(-1)[k])
                a.noise_seed = int(d[2]) & 0xFFFFFFFF
            elif d[0] == "grid":
                a.distort = 2
                tables[i, 0], tables[i, 1] = A.grid_distortion_maps(size, size, d[1], d[2])
            else:
                a.distort = 3
                a.k, a.cx, a.cy = float(d[1]), float(np.float32(size * 0.5 + d[2])), float(np.float32(size * 0.5 + d[3]))
        if p["clahe_clip"]:
            a.clahe_clip, a.clahe_limit = float(p["clahe_clip"]), A.clahe_limit(p["clahe_clip"], size)
        luts[i] = A.intensity_lut(p["intensity"])
    return arr, luts, tables


def augment_batch(images_u8: torch.Tensor, masks_u8: torch.Tensor, rng: np.random.Generator | None, params: list | None = None,
                  want_fields: bool = False):
    """images_u8 / masks_u8: (n, size, size) uint8 on the GPU -> (x (n, 1, size, size) fp32 normalised, masks (n, size, size)
    uint8).  ``rng`` draws the reference pipeline's parameters per sample (None or ``params=[]``-style identity: normalise only);
    ``params`` overrides the draws (tests).  With ``want_fields`` the elastic displacement fields are returned as a third value."""
    from .. import _lib
    from .._lib import check, lib, ptr
    n, size = images_u8.shape[0], images_u8.shape[-1]
    if images_u8.dtype != torch.uint8 or masks_u8.dtype != torch.uint8 or images_u8.shape[-2] != size:
        raise ValueError("augment_batch takes square uint8 images and masks")
    if size % 8:
        raise ValueError("augment_batch: the slice size must be a multiple of 8 (CLAHE's 8 x 8 tiles)")
    if params is None:
        identity = dict(size=size, crop=None, flip_v=False, rot_k=0, transpose=False, distort=None, clahe_clip=0.0, intensity=None)
        params = [A.sample_params(rng, size) if rng is not None else identity for _ in range(n)]
    arr, luts, tables = pack_params(params, size)
    dev = images_u8.device
    pd = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    ld, td = torch.from_numpy(luts).to(dev), torch.from_numpy(tables).to(dev)
    images_u8, masks_u8 = images_u8.reshape(n, size, size).contiguous(), masks_u8.reshape(n, size, size).contiguous()
    x = torch.empty((n, 1, size, size), dtype=torch.float32, device=dev)
    m = torch.empty((n, size, size), dtype=torch.uint8, device=dev)
    nbytes = lib.vs_augment_workspace(n, size)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    fields = torch.zeros((n, 2, size, size), dtype=torch.float32, device=dev) if want_fields else None
    check(lib.vs_augment_batch(ptr(images_u8), ptr(masks_u8), n, size, ptr(pd), ptr(ld), ptr(td), ptr(x), ptr(m), ptr(ws), nbytes,
                               ptr(fields), _lib.stream_ptr()))
    return (x, m, fields, params) if want_fields else (x, m)
