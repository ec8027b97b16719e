"""Volume loading + optional downsample / clip-to-uint8 (volume_segmantics/data/base_data_manager.py:10-42)."""
import logging
from pathlib import Path
from types import SimpleNamespace
from typing import Union

import numpy as np

from ..utilities import base_data_utils as utils


class BaseDataManager:
    def __init__(self, data_vol: Union[Path, str, np.ndarray], settings: SimpleNamespace) -> None:
        self.settings = settings
        self.st_dev_factor = settings.st_dev_factor
        self.downsample = settings.downsample
        self.data_vol_path = utils.setup_path_if_exists(data_vol)
        if self.data_vol_path is not None:
            self.data_vol, self.input_data_chunking = utils.get_numpy_from_path(
                self.data_vol_path, internal_path=settings.data_hdf5_path)
        elif isinstance(data_vol, np.ndarray):
            self.data_vol, self.input_data_chunking = data_vol, True
        else:
            raise TypeError("data_vol must be a path or a numpy array")
        self._preprocess_data()

    def _preprocess_device(self):
        """The GPU to pre-process on, or None for the reference's NumPy path (hosts without a GPU, dtypes the device path
        does not take).  With a GPU present a missing libvolseg_hip is an error, not a reason to fall back."""
        import torch

        if not torch.cuda.is_available() or not getattr(self.settings, "device_preprocess", True):
            return None
        from .. import _lib

        if self.data_vol.dtype.name not in _lib.VS_VOL:
            return None
        return f"cuda:{getattr(self.settings, 'cuda_device', 0)}"

    def _preprocess_data(self):
        if self.downsample:
            device = self._preprocess_device()
            if device is not None and self.data_vol.ndim == 3 and np.issubdtype(self.data_vol.dtype, np.integer):
                self.data_vol = utils.downsample_data_device(self.data_vol, device)      # exact for integer volumes (csrc/preprocess.hip)
            else:
                self.data_vol = utils.downsample_data(self.data_vol)
        self.data_vol_shape = self.data_vol.shape
        device = self._preprocess_device()
        if device is not None:   # statistics and the uint8 map on the GPU: the same numbers, bit for bit (csrc/preprocess.hip)
            uploaded = utils.volume_to_device(self.data_vol, device)
            self.data_mean = utils.device_nanmean_nanstd(*uploaded, self.data_vol.size, want_std=False)[0]
            logging.info(f"Mean value: {self.data_mean}")
            if self.settings.clip_data:
                self.data_vol = utils.clip_to_uint8_device(self.data_vol, self.data_mean, self.st_dev_factor, device, uploaded)
        else:
            self.data_mean = np.nanmean(self.data_vol)
            logging.info(f"Mean value: {self.data_mean}")
            if self.settings.clip_data:
                self.data_vol = utils.clip_to_uint8(self.data_vol, self.data_mean, self.st_dev_factor)
        if np.isnan(self.data_vol).any():
            self.data_vol = np.nan_to_num(self.data_vol, copy=False)
