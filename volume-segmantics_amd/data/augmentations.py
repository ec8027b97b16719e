"""The reference's training augmentations (volume_segmantics/data/augmentations.py:68-101: an albumentations ^1.1 pipeline),
restated in NumPy - albumentations and OpenCV are not installed next to torch here, and nothing of the pipeline is arithmetic
the GPU path depends on, so this is host-side data preparation with the reference's transforms, probabilities and parameter
ranges:

    RandomSizedCrop(min_max_height=(s/2, s), height=s, width=s, p=0.5)
    VerticalFlip(p=0.5); RandomRotate90(p=0.5); Transpose(p=0.5)
    OneOf([ElasticTransform(alpha=120, sigma=8.4, alpha_affine=4.8), GridDistortion(), OpticalDistortion(distort_limit=1,
           shift_limit=0.5)], p=0.5)
    CLAHE(p=0.5)                                                   (clip limit U(1, 4), 8 x 8 tiles)
    OneOf([RandomBrightnessContrast(), RandomGamma()], p=0.5)      (limits 0.2 / 0.2, brightness by max; gamma U{80..120} / 100)

Images are uint8 (H, W); masks follow every geometric transform with nearest-neighbour sampling and skip the intensity ones.
Borders reflect without repeating the edge sample (cv2.BORDER_REFLECT_101, albumentations' default).  The random draws come from
the generator passed in, so loader workers and epochs get their own streams (datasets.VolSeg2dDataset._worker_rng).
What cannot be promised without the libraries: bit equality with cv2's fixed-point bilinear interpolation and its CLAHE
rounding - the transforms have the same geometry and value maps, not the same last bits."""
from __future__ import annotations

import numpy as np


# ---- sampling ----------------------------------------------------------------------------------------------------------------
def _reflect101(idx: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    idx = np.mod(idx, period)
    return np.where(idx >= n, period - idx, idx)


def remap(image: np.ndarray, map_x: np.ndarray, map_y: np.ndarray, nearest: bool = False) -> np.ndarray:
    """cv2.remap(image, map_x, map_y, INTER_LINEAR | INTER_NEAREST, borderMode=BORDER_REFLECT_101)."""
    h, w = image.shape
    if nearest:
        xi = _reflect101(np.rint(map_x).astype(np.int64), w)
        yi = _reflect101(np.rint(map_y).astype(np.int64), h)
        return image[yi, xi]
    x0, y0 = np.floor(map_x), np.floor(map_y)
    fx, fy = (map_x - x0).astype(np.float32), (map_y - y0).astype(np.float32)
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
    xa, xb, ya, yb = _reflect101(x0, w), _reflect101(x0 + 1, w), _reflect101(y0, h), _reflect101(y0 + 1, h)
    img = image.astype(np.float32)
    top = img[ya, xa] * (1 - fx) + img[ya, xb] * fx
    bot = img[yb, xa] * (1 - fx) + img[yb, xb] * fx
    out = top * (1 - fy) + bot * fy
    return np.clip(np.rint(out), 0, 255).astype(image.dtype) if np.issubdtype(image.dtype, np.integer) else out.astype(image.dtype)


def resize(image: np.ndarray, height: int, width: int, nearest: bool = False) -> np.ndarray:
    """cv2.resize(..., INTER_LINEAR / INTER_NEAREST): pixel centres map as (dst + 0.5) * scale - 0.5, no antialiasing."""
    h, w = image.shape
    if (h, w) == (height, width):
        return image
    if nearest:       # cv2's nearest: floor(dst * scale)
        yi = np.minimum((np.arange(height, dtype=np.float32) * np.float32(np.float32(h) / np.float32(height))).astype(np.int64), h - 1)
        xi = np.minimum((np.arange(width, dtype=np.float32) * np.float32(np.float32(w) / np.float32(width))).astype(np.int64), w - 1)
        return image[np.ix_(yi, xi)]
    my = np.clip((np.arange(height, dtype=np.float32) + 0.5) * np.float32(np.float32(h) / np.float32(height)) - np.float32(0.5), 0, h - 1)
    mx = np.clip((np.arange(width, dtype=np.float32) + 0.5) * np.float32(np.float32(w) / np.float32(width)) - np.float32(0.5), 0, w - 1)
    gx, gy = np.meshgrid(mx, my)
    return remap(image, gx, gy)


# ---- transforms (image uint8 (H, W), mask or None; parameters drawn by the callers below) ------------------------------------------
def random_sized_crop(image, mask, rng, min_max_height, height, width, w2h_ratio=1.0):
    crop_h = int(rng.integers(min_max_height[0], min_max_height[1] + 1))
    crop_w = int(crop_h * w2h_ratio)
    h, w = image.shape
    crop_h, crop_w = min(crop_h, h), min(crop_w, w)
    y1 = int((h - crop_h) * rng.random())
    x1 = int((w - crop_w) * rng.random())
    image = resize(image[y1:y1 + crop_h, x1:x1 + crop_w], height, width)
    if mask is not None:
        mask = resize(mask[y1:y1 + crop_h, x1:x1 + crop_w], height, width, nearest=True)
    return image, mask


def _affine_from_points(src, dst):
    """cv2.getAffineTransform: the 2 x 3 matrix M with dst = M [src; 1]."""
    a = np.hstack([src, np.ones((3, 1))]).astype(np.float64)
    return np.linalg.solve(a, dst.astype(np.float64)).T


def elastic_transform(image, mask, rng, alpha=120.0, sigma=120 * 0.07, alpha_affine=120 * 0.04):
    """albumentations.ElasticTransform (Simard et al.): a small random affine map about the centre, then a displacement field of
    Gaussian-smoothed uniform noise scaled by alpha."""
    from scipy.ndimage import gaussian_filter
    h, w = image.shape
    centre = np.array([h, w], dtype=np.float32) // 2
    side = min(h, w) // 3
    pts1 = np.float32([centre + side, [centre[0] + side, centre[1] - side], centre - side])
    pts2 = pts1 + rng.uniform(-alpha_affine, alpha_affine, size=pts1.shape).astype(np.float32)
    m = _affine_from_points(pts1, pts2)                       # forward map; sampling needs its inverse
    full = np.vstack([m, [0, 0, 1]])
    inv = np.linalg.inv(full)[:2]
    dx = gaussian_filter(rng.random((h, w)).astype(np.float32) * 2 - 1, sigma) * alpha
    dy = gaussian_filter(rng.random((h, w)).astype(np.float32) * 2 - 1, sigma) * alpha
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    ax = inv[0, 0] * gx + inv[0, 1] * gy + inv[0, 2]
    ay = inv[1, 0] * gx + inv[1, 1] * gy + inv[1, 2]
    def apply(img, nearest):
        warped = remap(img, ax.astype(np.float32), ay.astype(np.float32), nearest)     # warpAffine
        return remap(warped, gx + dx, gy + dy, nearest)
    return apply(image, False), (apply(mask, True) if mask is not None else None)


def grid_distortion_maps(h, w, xsteps, ysteps, num_steps=5):
    """albumentations.functional.grid_distortion's coordinate tables: cell i of width n // num_steps is stretched by steps[i]
    (np.linspace includes its end point, so even all-ones steps are a slight stretch, as in the library)."""
    def axis_map(n, steps):
        step = n // num_steps
        out = np.zeros(n, np.float32)
        prev = 0.0
        for idx in range(num_steps + 1):
            start = idx * step
            end = start + step
            if end > n:
                end, cur = n, float(n)
            else:
                cur = prev + step * steps[idx]
            if end > start:
                out[start:end] = np.linspace(prev, cur, end - start)
            prev = cur
        return out
    return axis_map(w, xsteps), axis_map(h, ysteps)


def grid_distortion(image, mask, rng, num_steps=5, distort_limit=0.3):
    h, w = image.shape
    xsteps = 1 + rng.uniform(-distort_limit, distort_limit, size=num_steps + 1)
    ysteps = 1 + rng.uniform(-distort_limit, distort_limit, size=num_steps + 1)
    mx, my = grid_distortion_maps(h, w, xsteps, ysteps, num_steps)
    gx, gy = np.meshgrid(mx, my)
    return remap(image, gx, gy), (remap(mask, gx, gy, nearest=True) if mask is not None else None)


def optical_distortion(image, mask, rng, distort_limit=1.0, shift_limit=0.5):
    """cv2.initUndistortRectifyMap with camera matrix [[w, 0, w/2 + dx], [0, h, h/2 + dy]] and distortion (k, k, 0, 0, 0)."""
    h, w = image.shape
    k = rng.uniform(-distort_limit, distort_limit)
    dx, dy = round(rng.uniform(-shift_limit, shift_limit)), round(rng.uniform(-shift_limit, shift_limit))
    cx, cy = w * 0.5 + dx, h * 0.5 + dy
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    x, y = (gx - cx) / w, (gy - cy) / h
    r2 = x * x + y * y
    f = 1 + k * r2 + k * r2 * r2
    mx, my = (x * f * w + cx).astype(np.float32), (y * f * h + cy).astype(np.float32)
    return remap(image, mx, my), (remap(mask, mx, my, nearest=True) if mask is not None else None)


def clahe_limit(clip_limit: float, size: int, tiles: int = 8) -> int:
    area = (size // tiles) ** 2
    return max(int(clip_limit * area / 256), 1)


def clahe(image, clip_limit=2.0, tile_grid=(8, 8)):
    """cv2.createCLAHE(clipLimit, tileGridSize).apply on a uint8 image: per-tile clipped histogram equalisation, bilinear blend
    of the four surrounding tile maps."""
    h, w = image.shape
    ty, tx = tile_grid
    ph, pw = (-h) % ty, (-w) % tx
    img = np.pad(image, ((0, ph), (0, pw)), mode="reflect") if (ph or pw) else image
    th, tw = img.shape[0] // ty, img.shape[1] // tx
    area = th * tw
    limit = max(int(clip_limit * area / 256), 1)
    luts = np.empty((ty, tx, 256), np.float32)
    for i in range(ty):
        for j in range(tx):
            hist = np.bincount(img[i * th:(i + 1) * th, j * tw:(j + 1) * tw].ravel(), minlength=256).astype(np.int64)
            excess = int(np.maximum(hist - limit, 0).sum())
            hist = np.minimum(hist, limit)
            hist += excess // 256
            rest = excess % 256
            if rest:
                step = max(256 // rest, 1)
                hist[np.arange(0, 256, step)[:rest]] += 1
            luts[i, j] = np.clip(np.rint(np.cumsum(hist).astype(np.float32) * (np.float32(255.0) / np.float32(area))), 0, 255)
    yy = (np.arange(h, dtype=np.float32) + np.float32(0.5)) / np.float32(th) - np.float32(0.5)
    xx = (np.arange(w, dtype=np.float32) + np.float32(0.5)) / np.float32(tw) - np.float32(0.5)
    y0, x0 = np.floor(yy).astype(np.int64), np.floor(xx).astype(np.int64)
    wy, wx = (yy - y0)[:, None], (xx - x0)[None, :]
    y0c, y1c = np.clip(y0, 0, ty - 1)[:, None], np.clip(y0 + 1, 0, ty - 1)[:, None]
    x0c, x1c = np.clip(x0, 0, tx - 1)[None, :], np.clip(x0 + 1, 0, tx - 1)[None, :]
    v = image.astype(np.int64)
    out = ((luts[y0c, x0c, v] * (1 - wx) + luts[y0c, x1c, v] * wx) * (1 - wy) +
           (luts[y1c, x0c, v] * (1 - wx) + luts[y1c, x1c, v] * wx) * wy)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def brightness_contrast_lut(alpha, beta) -> np.ndarray:
    """albumentations' uint8 path (brightness_by_max=True): LUT of clip(v * alpha + beta * 255)."""
    return np.clip(np.arange(256, dtype=np.float32) * alpha + beta * 255.0, 0, 255).astype(np.uint8)


def gamma_lut(gamma) -> np.ndarray:
    return (np.power(np.arange(256, dtype=np.float32) / 255.0, gamma) * 255.0).astype(np.uint8)


def intensity_lut(intensity) -> np.ndarray:
    """The 256-entry map of a sample's RandomBrightnessContrast / RandomGamma draw (identity for None)."""
    if intensity is None:
        return np.arange(256, dtype=np.uint8)
    return brightness_contrast_lut(intensity[1], intensity[2]) if intensity[0] == "bc" else gamma_lut(intensity[1])


def brightness_contrast(image, alpha, beta):
    return brightness_contrast_lut(alpha, beta)[image]


def gamma_transform(image, gamma):
    return gamma_lut(gamma)[image]


# ---- the pipeline: parameters are drawn once (host), then applied on the host (below) or on the device (gpu_augment.py) ------------
def sample_params(rng: np.random.Generator, img_size: int) -> dict:
    """One sample's draws for get_train_augs(img_size): every transform's coin and parameters, in pipeline order."""
    p = {"size": img_size, "crop": None, "flip_v": False, "rot_k": 0, "transpose": False, "distort": None, "clahe_clip": 0.0,
         "intensity": None}
    if rng.random() < 0.5:       # RandomSizedCrop(min_max_height=(s/2, s), w2h_ratio=1)
        ch = int(rng.integers(img_size // 2, img_size + 1))
        p["crop"] = (ch, int(ch * 1.0), float(rng.random()), float(rng.random()))       # height, width, h_start, w_start
    p["flip_v"] = bool(rng.random() < 0.5)
    if rng.random() < 0.5:
        p["rot_k"] = int(rng.integers(0, 4))
    p["transpose"] = bool(rng.random() < 0.5)
    if rng.random() < 0.5:       # OneOf: members chosen in proportion to their own p (all 0.5 -> uniformly)
        kind = ("elastic", "grid", "optical")[int(rng.integers(0, 3))]
        if kind == "elastic":
            p["distort"] = ("elastic", rng.uniform(-120 * 0.04, 120 * 0.04, size=(3, 2)).astype(np.float32), int(rng.integers(0, 2 ** 31)))
        elif kind == "grid":
            p["distort"] = ("grid", 1 + rng.uniform(-0.3, 0.3, size=6), 1 + rng.uniform(-0.3, 0.3, size=6))
        else:
            p["distort"] = ("optical", float(rng.uniform(-1.0, 1.0)), round(rng.uniform(-0.5, 0.5)), round(rng.uniform(-0.5, 0.5)))
    if rng.random() < 0.5:
        p["clahe_clip"] = float(rng.uniform(1, 4))
    if rng.random() < 0.5:
        if rng.integers(0, 2) == 0:
            p["intensity"] = ("bc", float(1.0 + rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2)))
        else:
            p["intensity"] = ("gamma", int(rng.integers(80, 121)) / 100.0)
    return p


def elastic_inverse_affine(h, w, jitter):
    centre = np.array([h, w], dtype=np.float32) // 2
    side = min(h, w) // 3
    pts1 = np.float32([centre + side, [centre[0] + side, centre[1] - side], centre - side])
    m = _affine_from_points(pts1, pts1 + jitter)
    return np.linalg.inv(np.vstack([m, [0, 0, 1]]))[:2].astype(np.float32)


def elastic_fields(h, w, seed, alpha=120.0, sigma=120 * 0.07):
    from scipy.ndimage import gaussian_filter
    r = np.random.default_rng(seed)
    dx = gaussian_filter(r.random((h, w)).astype(np.float32) * 2 - 1, sigma) * alpha
    dy = gaussian_filter(r.random((h, w)).astype(np.float32) * 2 - 1, sigma) * alpha
    return dx.astype(np.float32), dy.astype(np.float32)


def apply_params(image: np.ndarray, mask: np.ndarray, p: dict, fields=None):
    """The pipeline for one (image, mask) pair (uint8, (size, size)) with the draws ``p``; ``fields`` = (dx, dy) displacement
    fields for the elastic transform (default: generated from the seed in ``p``)."""
    s = p["size"]
    if p["crop"] is not None:
        ch, cw, hs, ws = p["crop"]
        h, w = image.shape
        ch, cw = min(ch, h), min(cw, w)
        y1, x1 = int((h - ch) * hs), int((w - cw) * ws)
        image = resize(image[y1:y1 + ch, x1:x1 + cw], s, s)
        mask = resize(mask[y1:y1 + ch, x1:x1 + cw], s, s, nearest=True)
    if p["flip_v"]:
        image, mask = image[::-1], mask[::-1]
    if p["rot_k"]:
        image, mask = np.rot90(image, p["rot_k"]), np.rot90(mask, p["rot_k"])
    if p["transpose"]:
        image, mask = image.T, mask.T
    image, mask = np.ascontiguousarray(image), np.ascontiguousarray(mask)
    d = p["distort"]
    if d is not None:
        h, w = image.shape
        gx, gy = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
        if d[0] == "elastic":
            inv = elastic_inverse_affine(h, w, d[1])
            ax = inv[0, 0] * gx + inv[0, 1] * gy + inv[0, 2]
            ay = inv[1, 0] * gx + inv[1, 1] * gy + inv[1, 2]
            dx, dy = fields if fields is not None else elastic_fields(h, w, d[2])
            image = remap(remap(image, ax, ay), gx + dx, gy + dy)
            mask = remap(remap(mask, ax, ay, True), gx + dx, gy + dy, True)
        elif d[0] == "grid":
            mx, my = grid_distortion_maps(h, w, d[1], d[2])
            mx, my = np.meshgrid(mx, my)
            image, mask = remap(image, mx, my), remap(mask, mx, my, True)
        else:
            k, sx, sy = d[1], d[2], d[3]
            cx, cy = np.float32(w * 0.5 + sx), np.float32(h * 0.5 + sy)
            x, y = (gx - cx) / np.float32(w), (gy - cy) / np.float32(h)
            r2 = x * x + y * y
            f = 1 + np.float32(k) * r2 + np.float32(k) * r2 * r2
            mx, my = x * f * np.float32(w) + cx, y * f * np.float32(h) + cy
            image, mask = remap(image, mx, my), remap(mask, mx, my, True)
    if p["clahe_clip"]:
        image = clahe(image, clip_limit=p["clahe_clip"])
    it = p["intensity"]
    if it is not None:
        image = brightness_contrast(image, it[1], it[2]) if it[0] == "bc" else gamma_transform(image, it[1])
    return image, mask


def train_augment(image: np.ndarray, mask: np.ndarray, img_size: int, rng: np.random.Generator):
    """get_train_augs(img_size) of the reference applied to one (image, mask) pair (both (img_size, img_size) uint8)."""
    return apply_params(image, mask, sample_params(rng, img_size))
