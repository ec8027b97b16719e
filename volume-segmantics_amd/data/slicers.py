"""TrainingDataSlicer - 3-D data / label volumes to 2-D PNG slices on disk for training
(reference: volume_segmantics/data/slicers.py:14-149).  Host-side data preparation next to the accelerated path: label classes
are made sequential from zero (:48-71), slices are written along z / y / x per ``training_axes`` as
``<prefix>_<axis>_stack_<index>.png`` (:100-118), volumes that are not uint8 go through skimage's ``img_as_ubyte`` (restated
below: scikit-image is not installed next to torch here) and binary labels are clamped to {0, 1} (:120-135)."""
from __future__ import annotations

import logging
import os
from pathlib import Path
from types import SimpleNamespace
from typing import Union

import numpy as np

from ..utilities import base_data_utils as utils
from .base_data_manager import BaseDataManager


def _scale_bits(a: np.ndarray, n: int, m: int) -> np.ndarray:
    """skimage.util.dtype._scale: integers of n significant bits to m bits (a copy)."""
    a = a.copy()
    if n > m:                       # downscale with precision loss
        a //= 2 ** (n - m)
        return a
    if n == m:
        return a
    if m % n == 0:                  # exact upscale: replicate the bit pattern
        a *= (2 ** m - 1) // (2 ** n - 1)
        return a
    o = (m // n + 1) * n            # upscale to a multiple of n bits, then down to m
    a = a.astype(np.int64 if a.dtype.kind == "i" else np.uint64)
    a *= (2 ** o - 1) // (2 ** n - 1)
    a //= 2 ** (o - m)
    return a


def img_as_ubyte(image: np.ndarray) -> np.ndarray:
    """skimage.img_as_ubyte (scikit-image 0.18 ``util.dtype._convert(image, np.uint8)``), restated; known answers from the real
    function are in tests/test_host_logic.py.  NOTE the behaviour the reference inherits: integers are RESCALED by bit depth, so
    an int64 / int32 / uint32 label volume with values 0..3 becomes all zeros and uint16 1000 becomes 3 - the reference's label
    volumes are uint8 in practice (HDF5 / TIFF), which pass through untouched."""
    image = np.asarray(image)
    dt = image.dtype
    if dt == np.uint8:
        return image
    if dt == np.bool_:
        return image.astype(np.uint8) * 255
    if dt.kind == "f":
        if image.size and (np.min(image) < -1.0 or np.max(image) > 1.0):
            raise ValueError("Images of type float must be between -1 and 1.")
        out = np.multiply(image, 255, dtype=np.float64 if dt.itemsize > 4 else np.float32)
        np.rint(out, out=out)
        np.clip(out, 0, 255, out=out)
        return out.astype(np.uint8)
    if dt.kind == "u":
        return _scale_bits(image, 8 * dt.itemsize, 8).astype(np.uint8)
    if dt.kind == "i":
        scaled = _scale_bits(image, 8 * dt.itemsize - 1, 8)
        return np.maximum(scaled, 0).astype(np.uint8)
    raise ValueError(f"img_as_ubyte: unsupported dtype {dt}")


def imsave_png(path: Union[str, Path], data: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(data)).save(str(path))


class TrainingDataSlicer(BaseDataManager):
    """Image pre-processing (BaseDataManager) + slicing of the data and label volumes to PNG files in the xy (z), xz (y) and
    yz (x) planes."""

    def __init__(self, data_vol: Union[str, np.ndarray], label_vol: Union[str, np.ndarray], settings: SimpleNamespace):
        super().__init__(data_vol, settings)
        self.data_im_out_dir = None
        self.seg_im_out_dir = None
        self.multilabel = False
        self.settings = settings
        self.label_vol_path = utils.setup_path_if_exists(label_vol)
        if self.label_vol_path is not None:
            self.seg_vol, _ = utils.get_numpy_from_path(self.label_vol_path, internal_path=settings.seg_hdf5_path)
        elif isinstance(label_vol, np.ndarray):
            self.seg_vol = label_vol
        else:
            raise TypeError("label_vol must be a path or a numpy array")
        self._preprocess_labels()

    def _preprocess_labels(self):
        seg_classes = np.unique(self.seg_vol)
        self.num_seg_classes = len(seg_classes)
        if self.num_seg_classes > 2:
            self.multilabel = True
        logging.info(f"Number of classes in segmentation dataset: {self.num_seg_classes}")
        logging.info(f"These classes are: {seg_classes}")
        if seg_classes[0] != 0 or not utils.sequential_labels(seg_classes):
            logging.info("Fixing label classes.")
            self._fix_label_classes(seg_classes)
        self.codes = [f"label_val_{i}" for i in seg_classes]

    def _fix_label_classes(self, seg_classes):
        """Relabel so that the classes are 0 .. K-1 (ascending input order), in place as the reference does."""
        for idx, current in enumerate(seg_classes):
            self.seg_vol[self.seg_vol == current] = idx

    def output_data_slices(self, data_dir: Path, prefix: str) -> None:
        self.data_im_out_dir = Path(data_dir)
        logging.info("Slicing data volume and saving slices to disk")
        os.makedirs(data_dir, exist_ok=True)
        self._output_slices_to_disk(self.data_vol, Path(data_dir), prefix)

    def output_label_slices(self, data_dir: Path, prefix: str) -> None:
        self.seg_im_out_dir = Path(data_dir)
        logging.info("Slicing label volume and saving slices to disk")
        os.makedirs(data_dir, exist_ok=True)
        self._output_slices_to_disk(self.seg_vol, Path(data_dir), prefix, label=True)

    def _output_slices_to_disk(self, data_arr, output_path, name_prefix, label=False):
        axis_enum = utils.get_training_axis(self.settings)
        for axis, index in utils.get_axis_index_pairs(data_arr.shape, axis_enum):
            self._output_im(utils.axis_index_to_slice(data_arr, axis, index), output_path / f"{name_prefix}_{axis}_stack_{index}", label)

    def _output_im(self, data, path, label=False):
        if data.dtype != np.uint8:
            data = img_as_ubyte(data)
        if label and not self.multilabel:
            data = data.copy()
            data[data > 1] = 1
        imsave_png(f"{path}.png", data)

    def _delete_image_dir(self, im_dir_path):
        if im_dir_path is not None and Path(im_dir_path).exists():
            ims = list(Path(im_dir_path).glob("*.png"))
            logging.info(f"Deleting {len(ims)} images.")
            for im in ims:
                im.unlink()
            logging.info("Deleting the empty directory.")
            Path(im_dir_path).rmdir()

    def clean_up_slices(self) -> None:
        """Deletes data and label image slices created by the slicer."""
        self._delete_image_dir(self.data_im_out_dir)
        self._delete_image_dir(self.seg_im_out_dir)
