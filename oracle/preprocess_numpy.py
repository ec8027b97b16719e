"""CPU oracle of the volume pre-processing step (TEST INFRASTRUCTURE - never imported by the product path).

Restates, for parity tests of csrc/preprocess.hip:
  * BaseDataManager._preprocess_data (volume_segmantics/data/base_data_manager.py:29-42): data_mean = np.nanmean(volume),
    then clip_to_uint8 when settings.clip_data;
  * clip_to_uint8 (volume_segmantics/utilities/base_data_utils.py:243-287): bounds = mean -+ factor * np.nanstd(volume),
    NaN -> mean, clip, subtract, divide, clip to [0, 1], * 255, truncate to uint8 - in place, in the volume's float type
    (integers are converted to float64 first, :277-281).

The statistics are NumPy's (the reference's dependency, numpy ^1.18; 2.2.6 in this image), and the bounds decide which side
of a truncation boundary a voxel lands on, so the summation ORDER is part of the contract.  `np_order_sum` spells that order
out (numpy/_core/src/umath/loops_utils.h.src `pairwise_sum`, applied to consecutive 8192-element buffers whose results are
added in sequence) and `nanmean` / `nanstd` compose it the way numpy/lib/_nanfunctions_impl.py (inexact dtypes) and
numpy/_core/_methods.py `_mean` / `_var` (integer dtypes) do.  Pinned two ways: tests/test_oracle_goldens.py checks these
against np.nanmean / np.nanstd themselves (bit-equal) and against tests/golden/g8_clip_to_uint8.npz, which the reference's
own BaseDataManager produced (oracle/gen_goldens.py).
"""
from __future__ import annotations

import numpy as np

BUF = 8192      # elements per reduction buffer (NPY_BUFSIZE)
BLOCK = 128     # PW_BLOCKSIZE


def _pairwise(a: np.ndarray):
    """pairwise_sum of a 1-D array of at most BUF elements, in a's own dtype."""
    n, T = len(a), a.dtype.type
    if n < 8:
        res = T(0)
        for x in a:
            res = T(res + x)
        return res
    if n <= BLOCK:
        m = n - n % 8
        rows = a[:m].reshape(-1, 8)
        r = rows[0].copy()
        for row in rows[1:]:
            r = r + row                                   # 8 independent accumulators
        res = T(T(T(r[0] + r[1]) + T(r[2] + r[3])) + T(T(r[4] + r[5]) + T(r[6] + r[7])))
        for x in a[m:]:
            res = T(res + x)
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return T(_pairwise(a[:n2]) + _pairwise(a[n2:]))


def np_order_sum(a: np.ndarray):
    """np.add.reduce over a C-contiguous array of an inexact dtype, with the order of additions written out."""
    flat = np.ascontiguousarray(a).reshape(-1)
    T = flat.dtype.type
    nfull = len(flat) // BUF
    acc = T(0)
    if nfull:
        full = flat[:nfull * BUF].reshape(nfull, BUF // BLOCK, BLOCK // 8, 8)
        r = full[:, :, 0, :].copy()
        for i in range(1, BLOCK // 8):
            r = r + full[:, :, i, :]
        s = ((r[..., 0] + r[..., 1]) + (r[..., 2] + r[..., 3])) + ((r[..., 4] + r[..., 5]) + (r[..., 6] + r[..., 7]))
        while s.shape[1] > 1:                             # the halving of an 8192-element buffer is a balanced tree
            s = s[:, 0::2] + s[:, 1::2]
        for v in s[:, 0]:
            acc = T(acc + v)
    if len(flat) % BUF:
        acc = T(acc + _pairwise(flat[nfull * BUF:]))
    return acc


def nanmean(data: np.ndarray):
    """np.nanmean(data) over all axes."""
    if np.issubdtype(data.dtype, np.inexact):             # nanmean: NaN -> 0, sum / count of non-NaN (_divide_by_count)
        mask = np.isnan(data)
        tot = np_order_sum(np.where(mask, data.dtype.type(0), data))
        with np.errstate(invalid="ignore", divide="ignore"):
            return tot.dtype.type(tot / np.intp(data.size - mask.sum()))
    tot = np_order_sum(data.astype(np.float64))           # _mean: integers are summed as float64
    return np.float64(tot / np.intp(data.size))


def nanstd(data: np.ndarray):
    """np.nanstd(data) over all axes (ddof = 0)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        if np.issubdtype(data.dtype, np.inexact):         # nanvar
            T = data.dtype.type
            mask = np.isnan(data)
            arr = np.where(mask, T(0), data)
            cnt = np.intp(data.size - mask.sum())
            avg = T(np_order_sum(arr) / cnt)
            dev = arr - avg
            dev[mask] = 0
            var = np_order_sum(dev * dev)
            var = T(var / cnt)
            return T(np.sqrt(var))
        avg = nanmean(data)                               # _var with dtype f8
        dev = data - avg
        var = np.float64(np_order_sum(dev * dev) / np.intp(data.size))
        return np.float64(np.sqrt(var))


def clip_to_uint8(data: np.ndarray, data_mean, st_dev_factor: float) -> np.ndarray:
    """base_data_utils.py:243-287 (the input is not modified)."""
    st_dev = nanstd(data)
    lower = data_mean - (st_dev * st_dev_factor)
    upper = data_mean + (st_dev * st_dev_factor)
    work = data.astype(np.float64) if np.issubdtype(data.dtype, np.integer) else data.copy()
    if np.isnan(work).any():
        work[np.isnan(work)] = data_mean
    work = np.clip(work, lower, upper)
    work = work - lower
    work = work / (upper - lower)
    work = np.clip(work, 0.0, 1.0)
    work = work * 255
    return work.astype(np.uint8)


def preprocess(data: np.ndarray, st_dev_factor: float, clip_data: bool = True):
    """base_data_manager.py:29-42 without the optional downsample: (data_mean, volume)."""
    mean = nanmean(data)
    return mean, (clip_to_uint8(data, mean, st_dev_factor) if clip_data else data)
