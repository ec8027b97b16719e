"""-m gpu: volume pre-processing on the device (csrc/preprocess.hip through the C ABI) against the CPU oracle
(oracle/preprocess_numpy.py, pinned to NumPy and to the reference's BaseDataManager goldens): mean, standard deviation and
every uint8 voxel must be EQUAL - the statistics follow NumPy's order of additions, the map NumPy's per-step rounding."""
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hip_helpers import DEV
from oracle import preprocess_numpy as Q

pytestmark = pytest.mark.gpu
G8_CASES = ("f32", "f64", "u16", "i16", "u8", "f32b")


def _stats(vol):
    from volume_segmantics_amd.utilities import base_data_utils as U
    dev, vtype = U.volume_to_device(vol, DEV)
    return U.device_nanmean_nanstd(dev, vtype, vol.size)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.int64, np.uint64])
def test_statistics_equal_numpy_order(dtype):
    rng = np.random.default_rng(3)
    for shape in [(1, 1, 5), (1, 3, 43), (2, 5, 8191), (3, 8192), (5, 77, 131), (2, 8192 * 3 + 9)]:
        if np.issubdtype(dtype, np.floating):
            vol = (rng.standard_normal(shape) * 700 + 90).astype(dtype)
            vol.reshape(-1)[::13] = np.nan
        else:
            info = np.iinfo(dtype)
            vol = rng.integers(max(info.min, -40000), min(info.max, 40000) + 1, shape).astype(dtype)
        mean, std = _stats(vol)
        assert mean == Q.nanmean(vol) and type(mean) is type(Q.nanmean(vol)), (dtype, shape)
        assert std == Q.nanstd(vol) and type(std) is type(Q.nanstd(vol)), (dtype, shape)


@pytest.mark.parametrize("case", G8_CASES)
def test_data_manager_on_device_equals_reference_golden(golden, case):
    """BaseDataManager with a GPU present takes the device path: same stored mean, same uint8 volume as the reference's."""
    from volume_segmantics_amd.data.base_data_manager import BaseDataManager
    g = golden("g8_clip_to_uint8.npz")
    s = SimpleNamespace(st_dev_factor=float(g[case + "__factor"]), downsample=False, clip_data=True, data_hdf5_path="/data", cuda_device=0)
    vol = g[case + "__in"].copy()
    dm = BaseDataManager(vol, s)
    assert dm.data_mean == g[case + "__mean"] and np.asarray(dm.data_mean).dtype == g[case + "__mean"].dtype
    assert dm.data_vol.dtype == np.uint8 and np.array_equal(dm.data_vol, g[case + "__out"])
    assert np.array_equal(vol, g[case + "__in"], equal_nan=True)       # the caller's array is left alone


def test_clip_counts_and_all_nan_free_float_volume():
    from volume_segmantics_amd import _lib as L
    from volume_segmantics_amd.utilities import base_data_utils as U
    rng = np.random.default_rng(5)
    vol = (rng.standard_normal((9, 64, 65)) * 3 + 1).astype(np.float32)
    mean = U.nanmean_device(vol, DEV)
    assert mean == Q.nanmean(vol)
    out = U.clip_to_uint8_device(vol, mean, 1.25, DEV)
    assert np.array_equal(out, Q.clip_to_uint8(vol, mean, 1.25))
    dev, vtype = U.volume_to_device(vol, DEV)
    std = Q.nanstd(vol)
    lo, hi = mean - std * 1.25, mean + std * 1.25
    counts = torch.zeros(2, dtype=torch.int64, device=DEV)
    o = torch.empty(vol.size, dtype=torch.uint8, device=DEV)
    L.check(L.lib.vs_clip_to_uint8(vtype, L.ptr(dev), vol.size, float(mean), float(lo), float(hi), L.ptr(o), L.ptr(counts), L.stream_ptr()))
    assert counts.tolist() == [int((vol > hi).sum()), int((vol < lo).sum())]


def test_full_size_uint16_volume_512_cube():
    """BASELINE config-3 size (512^3) as a 16-bit tomogram: 16 384 summation buffers, every voxel of the uint8 result."""
    from volume_segmantics_amd.utilities import base_data_utils as U
    rng = np.random.default_rng(11)
    vol = np.clip(rng.gamma(2.0, 5000.0, (512, 512, 512)), 0, 65535).astype(np.uint16)
    mean, std = _stats(vol)
    assert mean == Q.nanmean(vol) and std == Q.nanstd(vol)
    out = U.clip_to_uint8_device(vol, mean, 2.575, DEV)
    ref = Q.clip_to_uint8(vol, mean, 2.575)
    assert np.array_equal(out, ref)
    order = np.argsort(vol[100].reshape(-1), kind="stable")             # the map is monotone
    assert (np.diff(out[100].reshape(-1)[order].astype(np.int16)) >= 0).all()


# ---- the reference's own data-manager tests (tests/test_base_data_manager.py:57-96), on the device path -----------------
def _settings(**kw):
    base = dict(st_dev_factor=2.575, downsample=False, clip_data=False, data_hdf5_path="/data", cuda_device=0)
    base.update(kw)
    return SimpleNamespace(**base)


@pytest.fixture()
def rand_size():
    return np.random.default_rng(17).integers(10, 120, size=3)


def test_preprocess_like_reference_suite(rand_size):
    from volume_segmantics_amd.data.base_data_manager import BaseDataManager
    rng = np.random.default_rng(23)
    rand_int_volume = rng.integers(256, size=rand_size)                     # int64, as np.random.randint gives
    rand_float_volume = rng.uniform(-1, 1, size=rand_size)
    for vol in (rand_int_volume, rand_float_volume):                        # test_preprocess_clip_int / _float
        dm = BaseDataManager(vol.copy(), _settings(clip_data=True))
        host = BaseDataManager(vol.copy(), _settings(clip_data=True, device_preprocess=False))
        assert dm._preprocess_device() == "cuda:0" or dm.data_vol.dtype == np.uint8
        assert dm.data_vol.dtype == np.uint8 and dm.data_mean == host.data_mean and np.array_equal(dm.data_vol, host.data_vol)
    dm = BaseDataManager(rand_int_volume.copy(), _settings(downsample=True))    # test_preprocess_downsampled
    assert dm.data_vol.shape == tuple(int(np.ceil(s / 2)) for s in rand_int_volume.shape)
    nan_vol = rand_float_volume.copy()
    nan_vol[rng.integers(min(nan_vol.shape), size=3)] = np.nan                  # whole planes, as the reference's fixture does
    assert np.isnan(nan_vol).any()
    dm = BaseDataManager(nan_vol.copy(), _settings())                           # test_preprocess_replace_nan
    assert not np.isnan(dm.data_vol).any() and dm.data_mean == np.nanmean(nan_vol)
    dm = BaseDataManager(nan_vol.copy(), _settings(clip_data=True))             # test_preprocess_replace_nan_clip
    host = BaseDataManager(nan_vol.copy(), _settings(clip_data=True, device_preprocess=False))
    assert dm.data_vol.dtype == np.uint8 and np.array_equal(dm.data_vol, host.data_vol)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.int32, np.int8])
def test_downsample_on_device_equals_block_reduce_nanmean(dtype):
    """`downsample: True` (data/base_data_manager.py:36-38 -> utilities/base_data_utils.py:161-163: skimage block_reduce with
    np.nanmean over 2 x 2 x 2 blocks, odd edges zero padded): vs_downsample2x_mean gives the host restatement's float64 values bit
    for bit for integer volumes (even, odd and size-1 dimensions), and BaseDataManager takes that path with a GPU present."""
    from volume_segmantics_amd.data.base_data_manager import BaseDataManager
    from volume_segmantics_amd.utilities import base_data_utils as U
    rng = np.random.default_rng(9)
    info = np.iinfo(dtype)
    for shape in [(8, 10, 12), (9, 11, 13), (1, 7, 2), (33, 64, 65), (2, 2, 2)]:
        vol = rng.integers(max(info.min, -30000), min(info.max, 30000) + 1, shape).astype(dtype)
        ref = U.downsample_data(vol)
        got = U.downsample_data_device(vol, DEV)
        assert got.dtype == np.float64 and got.shape == ref.shape == tuple((s + 1) // 2 for s in shape)
        assert np.array_equal(got, ref), (dtype, shape)
    vol = rng.integers(0, 200, (40, 50, 61)).astype(dtype) if dtype != np.int8 else rng.integers(-100, 100, (40, 50, 61)).astype(dtype)
    s = SimpleNamespace(st_dev_factor=2.575, downsample=True, clip_data=True, data_hdf5_path="/data", cuda_device=0)
    on_device = BaseDataManager(vol.copy(), s)
    on_host = BaseDataManager(vol.copy(), SimpleNamespace(**{**vars(s), "device_preprocess": False}))
    assert on_device.data_vol.shape == (20, 25, 31) and on_device.data_mean == on_host.data_mean
    assert on_device.data_vol.dtype == np.uint8 and np.array_equal(on_device.data_vol, on_host.data_vol)
