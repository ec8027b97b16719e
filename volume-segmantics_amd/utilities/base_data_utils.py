"""Enums and small helpers of the hot path, mirroring volume_segmantics/utilities/base_data_utils.py
(the reference lines each item follows are cited inline).  Disk I/O helpers import their third-party
library lazily: HDF5 goes through h5py where it is installed and through `hdf5_lite` (ctypes on libhdf5) where it is not."""
from __future__ import annotations

import logging
import math
import sys
from enum import Enum
from itertools import chain, product
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

from . import config as cfg


class Quality(Enum):  # base_data_utils.py:21-32 - number of (axis, rotation) directions predicted
    LOW = 1
    MEDIUM = 3
    HIGH = 12


class Axis(Enum):  # :35-39
    Z = 0
    Y = 1
    X = 2
    ALL = 4


class ModelType(Enum):  # :42-50
    U_NET = 1
    U_NET_PLUS_PLUS = 2
    FPN = 3
    DEEPLABV3 = 4
    DEEPLABV3_PLUS = 5
    MA_NET = 6
    LINKNET = 7
    PAN = 8


def create_enum_from_setting(setting_str, enum):
    """:53-64 - case-insensitive lookup; an unknown name is a user error -> log + exit(1)."""
    if isinstance(setting_str, Enum):
        return setting_str
    try:
        return enum[str(setting_str).upper()]
    except KeyError:
        logging.error(f"{enum.__name__}: {setting_str} is not valid. Options are {[k.name for k in enum]}.")
        sys.exit(1)


def get_prediction_quality(settings: SimpleNamespace) -> Quality:
    return create_enum_from_setting(settings.quality, Quality)


def get_model_type(settings: SimpleNamespace) -> ModelType:
    return create_enum_from_setting(settings.model["type"], ModelType)


def get_training_axis(settings: SimpleNamespace) -> Axis:  # :77-83, default "All"
    return create_enum_from_setting(getattr(settings, "training_axes", "All"), Axis)


def get_prediction_axis(settings: SimpleNamespace) -> Axis:  # :86-92, default "Z"
    return create_enum_from_setting(getattr(settings, "prediction_axis", "Z"), Axis)


def setup_path_if_exists(input_param):
    if isinstance(input_param, (str, Path)):
        return Path(input_param)
    return None


def get_batch_size(settings: SimpleNamespace, prediction: bool = False) -> int:
    """:104-122 with two changes: an explicit ``batch_size`` / ``prediction_batch_size`` settings key wins,
    and prediction on the HIP engine defaults to a large batch (eval-mode results are batch independent;
    a 288 GB MI355X is idle at the reference's 4).  Training keeps 12: train-mode BN statistics depend on it."""
    key = "prediction_batch_size" if prediction else "batch_size"
    override = getattr(settings, key, None)
    if override:
        return int(override)
    if prediction:
        return cfg.HIP_PRED_BATCH
    dev = int(getattr(settings, "cuda_device", 0))
    free_gb = cfg.BIG_CUDA_THRESHOLD
    if torch.cuda.is_available():
        total = torch.cuda.get_device_properties(dev).total_memory
        free_gb = (total - torch.cuda.memory_allocated(dev)) / 1024 ** 3
    batch = cfg.SMALL_CUDA_BATCH if free_gb < cfg.BIG_CUDA_THRESHOLD else cfg.BIG_CUDA_TRAIN_BATCH
    logging.info(f"Free GPU memory is {free_gb:0.2f} GB. Batch size will be {batch}.")
    return batch


def get_padded_dimension(dimension: int) -> int:
    """data/augmentations.py:30-43 (KATs 32->32, 33->64, 13->32, 0->0)."""
    d = cfg.IM_SIZE_DIVISOR
    return dimension if dimension % d == 0 else (math.floor(dimension / d) + 1) * d


def pad_crop_offsets(dim: int) -> tuple[int, int, int]:
    """(padded size, pad offset, crop offset) of one slice dimension.  The pad is albumentations' centred
    PadIfNeeded (floor(d/2) leading pixels); the crop is torchvision's center_crop
    (round-half-even(d/2)), so for d = 3 (mod 4) the output is shifted by one pixel - reproduced, not fixed
    (crop: base_data_utils.py:125-129)."""
    padded = get_padded_dimension(dim)
    d = padded - dim
    return padded, int(d / 2.0), int(round(d / 2.0))


def rotate_array_to_axis(array: np.ndarray, axis: Axis = Axis.Z) -> np.ndarray:
    """:132-138 - a view with the slicing axis first (self-inverse)."""
    if axis == Axis.Z:
        return array
    return array.swapaxes(0, axis.value)


def one_hot_encode_array(input_array: np.ndarray, num_labels: int) -> np.ndarray:
    """:141-147 -> (num_labels, *shape) uint8."""
    out = np.zeros((num_labels, input_array.size), dtype=np.uint8)
    out[input_array.ravel(), np.arange(input_array.size)] = 1
    return out.reshape((num_labels,) + input_array.shape)


def prepare_training_batch(batch, device, num_labels: int, augment_rng=None):
    """:150-158 - images to the device, masks to one-hot (B,K,H,W) uint8 (built on the device here).  A batch of raw uint8
    images (datasets.VolSeg2dDataset(augment="device")) is augmented - with the reference pipeline's draws from ``augment_rng``,
    or not at all when it is None - and normalised on the device (data/gpu_augment.py)."""
    if batch[1].numel() and not batch[1].is_cuda and int(batch[1].max()) >= num_labels:
        # torch.nn.functional.one_hot in the reference raises here (e.g. 0/255 PNG masks with 2 labels); the device kernel
        # would silently give such pixels an all-zero target
        raise RuntimeError("Class values must be smaller than num_classes.")
    inputs = batch[0].to(device, non_blocking=True)
    masks = batch[1].to(device, non_blocking=True)
    if inputs.dtype == torch.uint8:
        if not inputs.is_cuda:
            raise RuntimeError("raw uint8 training batches are augmented / normalised on the GPU: use augment='host' without one")
        from ..data.gpu_augment import augment_batch
        inputs, masks = augment_batch(inputs.reshape(inputs.shape[0], inputs.shape[-2], inputs.shape[-1]), masks.to(torch.uint8), augment_rng)
    if masks.is_cuda and masks.dtype == torch.uint8 and num_labels <= 255:
        from .._lib import check, lib, ptr, stream_ptr   # one HIP sweep instead of int64 one_hot + permute + cast
        masks = masks.contiguous()
        n, hw = masks.shape[0], masks[0].numel()
        targets = torch.empty((n, num_labels) + tuple(masks.shape[1:]), dtype=torch.uint8, device=masks.device)
        check(lib.vs_onehot_u8(ptr(masks), n, num_labels, hw, ptr(targets), stream_ptr()))
        return inputs, targets
    targets = torch.nn.functional.one_hot(masks.to(torch.int64), num_classes=num_labels).permute(0, 3, 1, 2).to(torch.uint8)
    return inputs, targets


def clip_to_uint8(data: np.ndarray, data_mean: float, st_dev_factor: float) -> np.ndarray:
    """:243-287 - clip to mean +- factor * std, rescale to [0, 255], truncate to uint8 (float64 NumPy semantics)."""
    logging.info("Clipping data and converting to uint8.")
    st_dev = np.nanstd(data)
    lower, upper = data_mean - st_dev * st_dev_factor, data_mean + st_dev * st_dev_factor
    with np.errstate(invalid="ignore"):
        above, below = int((data > upper).sum()), int((data < lower).sum())
    logging.info(f"Lower bound: {lower}, upper bound: {upper}; clipping {above} voxels above "
                 f"({above / data.size * 100:.3f}%) and {below} below ({below / data.size * 100:.3f}%).")
    if np.isnan(data).any():
        data = np.nan_to_num(data, copy=False, nan=data_mean)
    if np.issubdtype(data.dtype, np.integer):
        data = data.astype(float)
    data = np.clip(data, lower, upper, out=data)
    data -= lower
    data /= (upper - lower)
    data = np.clip(data, 0.0, 1.0, out=data)
    data *= 255
    return data.astype(np.uint8)


# ---- device path of BaseDataManager._preprocess_data (base_data_manager.py:29-42) --------------------------------------
# The same three results the reference takes with NumPy - np.nanmean, np.nanstd, the clipped uint8 volume - from
# libvolseg_hip (csrc/preprocess.hip): sums in NumPy's own reduction order and accumulation type, so mean, standard
# deviation, clip bounds and every uint8 voxel equal the NumPy ones bit for bit.  The scalar steps between the device passes
# are written with the NumPy scalar types the reference's expressions produce (numpy/_core/_methods.py _mean / _var,
# numpy/lib/_nanfunctions_impl.py nanmean / nanvar / _divide_by_count).
def volume_to_device(data: np.ndarray, device):
    """C-contiguous copy of a supported volume on the device (torch has no uint16/32/64: same-width signed views)."""
    import torch
    from .. import _lib

    if data.dtype.name not in _lib.VS_VOL:
        raise TypeError(f"volume dtype {data.dtype} has no device pre-processing path")
    view = {"uint16": np.int16, "uint32": np.int32, "uint64": np.int64}.get(data.dtype.name)
    host = np.ascontiguousarray(data)
    return torch.from_numpy(host.view(view) if view else host).to(device), _lib.VS_VOL[data.dtype.name]


def _device_sum(dev_data, vtype, n, op, avg):
    import torch
    from .. import _lib

    ws = torch.empty(_lib.lib.vs_volume_sum_workspace(n), dtype=torch.uint8, device=dev_data.device)
    out = torch.empty(2, dtype=torch.float64, device=dev_data.device)
    _lib.check(_lib.lib.vs_volume_sum(vtype, _lib.ptr(dev_data), n, op, float(avg), _lib.ptr(ws), ws.numel(), _lib.ptr(out), _lib.stream_ptr()))
    total, nans = out.cpu().numpy()
    return total, int(nans)


def device_nanmean_nanstd(dev_data, vtype: int, n: int, want_std: bool = True):
    """(np.nanmean, np.nanstd) of the volume, as NumPy scalars of the types NumPy returns."""
    acc = np.float32 if vtype == 0 else np.float64
    total, nans = _device_sum(dev_data, vtype, n, 0, 0.0)
    total = acc(total)
    with np.errstate(invalid="ignore", divide="ignore"):
        if vtype <= 1:                                  # inexact: nanmean / nanvar (always the masked path)
            cnt = np.intp(n - nans)
            mean = acc(total / cnt)                     # _divide_by_count: float64 division, result cast back
        else:                                           # integers: np.mean / np.var with dtype f8
            cnt = np.intp(n)
            mean = acc(total / cnt)
        if not want_std:
            return mean, None
        sq, _ = _device_sum(dev_data, vtype, n, 1, mean)
        var = acc(acc(sq) / cnt)
        return mean, acc(np.sqrt(var))


def clip_to_uint8_device(data: np.ndarray, data_mean, st_dev_factor: float, device="cuda:0", uploaded=None) -> np.ndarray:
    """clip_to_uint8 (:243-287) with the reductions and the map on the device; returns the uint8 volume on the host.
    `uploaded`: the (tensor, dtype code) volume_to_device already returned for `data`, to skip a second upload."""
    import torch
    from .. import _lib

    dev, vtype = uploaded if uploaded is not None else volume_to_device(data, device)
    n = data.size
    _, st_dev = device_nanmean_nanstd(dev, vtype, n)
    lower = data_mean - (st_dev * st_dev_factor)        # NumPy scalar arithmetic, as in the reference (:257-258)
    upper = data_mean + (st_dev * st_dev_factor)
    out = torch.empty(n, dtype=torch.uint8, device=dev.device)
    counts = torch.zeros(2, dtype=torch.int64, device=dev.device)
    # the map runs in float32 for float32 volumes and in float64 otherwise (integers: data.astype(float), :277-281)
    comp = np.float32 if vtype == 0 else np.float64
    _lib.check(_lib.lib.vs_clip_to_uint8(vtype, _lib.ptr(dev), n, float(comp(data_mean)), float(comp(lower)), float(comp(upper)),
                                         _lib.ptr(out), _lib.ptr(counts), _lib.stream_ptr()))
    above, below = (int(v) for v in counts.cpu())
    logging.info(f"Lower bound: {lower}, upper bound: {upper}; clipping {above} voxels above "
                 f"({above / n * 100:.3f}%) and {below} below ({below / n * 100:.3f}%).")
    return out.cpu().numpy().reshape(data.shape)


def nanmean_device(data: np.ndarray, device="cuda:0"):
    dev, vtype = volume_to_device(data, device)
    return device_nanmean_nanstd(dev, vtype, data.size, want_std=False)[0]


def downsample_data_device(data: np.ndarray, device="cuda:0") -> np.ndarray:
    """`downsample_data` for integer volumes on the GPU (vs_downsample2x_mean): the same float64 values, bit for bit."""
    import torch
    from .. import _lib
    dev, vtype = volume_to_device(data, device)
    d, h, w = data.shape
    out = torch.empty(((d + 1) // 2, (h + 1) // 2, (w + 1) // 2), dtype=torch.float64, device=dev.device)
    _lib.check(_lib.lib.vs_downsample2x_mean(vtype, _lib.ptr(dev), _lib.ptr(out), d, h, w, _lib.stream_ptr()))
    return out.cpu().numpy()


def downsample_data(data: np.ndarray, factor: int = 2) -> np.ndarray:
    """:161-163 - block nan-mean (skimage.measure.block_reduce semantics: edge blocks are zero padded)."""
    pads = [(0, (-s) % factor) for s in data.shape]
    padded = np.pad(data.astype(float), pads, mode="constant", constant_values=0)
    z, y, x = (s // factor for s in padded.shape)
    return np.nanmean(padded.reshape(z, factor, y, factor, x, factor), axis=(1, 3, 5))


def get_num_of_ims(vol_shape, axis_enum: Axis) -> int:
    return sum(vol_shape) if axis_enum == Axis.ALL else vol_shape[axis_enum.value]


def get_axis_index_pairs(vol_shape, axis_enum: Axis):
    if axis_enum == Axis.ALL:
        return chain(*(product(name, range(n)) for name, n in zip("zyx", vol_shape)))
    return product(axis_enum.name.lower(), range(vol_shape[axis_enum.value]))


def axis_index_to_slice(vol, axis: str, index: int):
    return vol[index] if axis == "z" else vol[:, index] if axis == "y" else vol[:, :, index]


def sequential_labels(unique_labels: np.ndarray) -> bool:
    return not np.where(np.diff(unique_labels) != 1)[0].size


# ---- disk I/O (outside the hot path; third-party libraries imported lazily) -----------------------------
def _h5py():
    """h5py when this interpreter has it, else None (then `hdf5_lite`, a ctypes binding to libhdf5, does the same two jobs)."""
    try:
        import h5py
        return h5py
    except ImportError:
        return None


def _h5lite():
    from . import hdf5_lite
    if not hdf5_lite.available():
        raise ImportError("HDF5 input/output needs h5py or the HDF5 C library (libhdf5; set VOLSEG_LIBHDF5): neither was found - "
                          "pass a numpy array / .npy file instead")
    return hdf5_lite


def numpy_from_hdf5(path, hdf5_path="/data", nexus=False):
    h5 = _h5py()
    if h5 is None:
        lite = _h5lite()
        if nexus:
            for cand in ("processed/result/data", "entry/final_result_tomo/data"):
                if lite.exists(path, cand):
                    hdf5_path = cand
                    break
            else:
                logging.error("NXS file: could not find a data entry, exiting!")
                sys.exit(1)
        return lite.read_dataset(path, hdf5_path)
    with h5.File(path, "r") as f:
        if nexus:
            for cand in ("processed/result/data", "entry/final_result_tomo/data"):
                if cand in f:
                    ds = f[cand]
                    break
            else:
                logging.error("NXS file: could not find a data entry, exiting!")
                sys.exit(1)
        else:
            ds = f[hdf5_path]
        return ds[()], ds.chunks


def numpy_from_tiff(path):
    from PIL import Image, ImageSequence
    with Image.open(path) as im:
        return np.stack([np.array(page) for page in ImageSequence.Iterator(im)])


def get_numpy_from_path(path: Path, internal_path: str = "/data"):
    """:218-236 plus .npy."""
    if path.suffix in cfg.TIFF_SUFFIXES:
        return numpy_from_tiff(path), True
    if path.suffix in cfg.HDF5_SUFFIXES:
        return numpy_from_hdf5(path, hdf5_path=internal_path, nexus=path.suffix == ".nxs")
    if path.suffix in cfg.NUMPY_SUFFIXES:
        return np.load(path), True
    logging.error(f"Unsupported volume file type {path.suffix}")
    sys.exit(1)


def save_data_to_hdf5(data, file_path, internal_path="/data", chunking=True):
    """:351-356; a ``.npy`` target is written with numpy instead."""
    file_path = Path(file_path)
    logging.info(f"Saving data of shape {data.shape} to {file_path}.")
    if file_path.suffix in cfg.NUMPY_SUFFIXES:
        np.save(file_path, data)
        return
    h5 = _h5py()
    if h5 is None:
        _h5lite().write_dataset(file_path, internal_path, np.asarray(data), chunks=chunking, compression=cfg.HDF5_COMPRESSION)
        return
    with h5.File(file_path, "w") as f:
        f.create_dataset(internal_path, data=data, chunks=chunking, compression=cfg.HDF5_COMPRESSION)


# checkpoints pickle ModelType under the reference's module path (see checkpoint_compat)
from ..checkpoint_compat import install_reference_aliases as _install_reference_aliases  # noqa: E402

_install_reference_aliases()
