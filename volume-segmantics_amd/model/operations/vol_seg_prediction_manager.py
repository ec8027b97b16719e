"""VolSeg2DPredictionManager - volume in, label volume out / to disk
(reference: volume_segmantics/model/operations/vol_seg_prediction_manager.py:12-100)."""
from pathlib import Path
from types import SimpleNamespace
from typing import Union

import numpy as np

from ...data.base_data_manager import BaseDataManager
from ...utilities import base_data_utils as utils
from .vol_seg_2d_predictor import VolSeg2dPredictor


class VolSeg2DPredictionManager(BaseDataManager):
    def __init__(self, model_file_path: str, data_vol: Union[str, np.ndarray], settings: SimpleNamespace) -> None:
        super().__init__(data_vol, settings)
        self.predictor = VolSeg2dPredictor(model_file_path, settings)
        self.settings = settings

    def get_label_codes(self) -> dict:
        return self.predictor.label_codes

    def predict_volume_to_path(self, output_path: Union[Path, None], quality: Union[utils.Quality, None] = None) -> np.ndarray:
        """LOW = one axis, MEDIUM = 3 axes, HIGH = 3 axes x 4 rotations, merged by maximum probability;
        ``one_hot`` returns per-class vote counts instead (:43-89)."""
        one_hot = self.settings.one_hot
        axis = utils.get_prediction_axis(self.settings)
        if quality is None:
            quality = utils.get_prediction_quality(self.settings)
        p, probs = self.predictor, None
        if quality == utils.Quality.LOW:
            if one_hot:
                prediction = p._predict_single_axis_to_one_hot(self.data_vol, axis=axis)
            else:
                prediction, probs = p._predict_single_axis(self.data_vol, axis=axis)
        elif quality == utils.Quality.MEDIUM:
            if one_hot:
                prediction = p._predict_3_ways_one_hot(self.data_vol)
            else:
                prediction, probs = p._predict_3_ways_max_probs(self.data_vol)
        elif quality == utils.Quality.HIGH:
            if one_hot:
                prediction = p._predict_12_ways_one_hot(self.data_vol)
            else:
                prediction, probs = p._predict_12_ways_max_probs(self.data_vol)
        else:
            raise ValueError(f"unknown quality {quality}")
        if output_path is not None:
            output_path = Path(output_path)
            utils.save_data_to_hdf5(prediction, output_path, chunking=self.input_data_chunking)
            if probs is not None and self.settings.output_probs:
                # the reference hard-codes the name: "<stem>_probs.h5" whatever the label file's suffix (:94-98)
                utils.save_data_to_hdf5(probs, f"{output_path.parent / output_path.stem}_probs.h5", chunking=self.input_data_chunking)
        return prediction
