"""MI355X-native (gfx950) engine for the volume-segmantics 2D-slice segmentation hot path.

Public surface mirrors the reference (volume_segmantics/model/__init__.py:1-6,
volume_segmantics/data/__init__.py:1-4, volume_segmantics/utilities/__init__.py:1-8);
submodules are imported lazily so that ``import volume_segmantics_amd`` stays cheap.
"""
__version__ = "0.1.0"

_LAZY = {
    "VolSeg2dTrainer": "volume_segmantics_amd.model.operations.vol_seg_2d_trainer",
    "VolSeg2dPredictor": "volume_segmantics_amd.model.operations.vol_seg_2d_predictor",
    "VolSeg2DPredictionManager": "volume_segmantics_amd.model.operations.vol_seg_prediction_manager",
    "create_model_on_device": "volume_segmantics_amd.model.model_2d",
    "create_model_from_file": "volume_segmantics_amd.model.model_2d",
    "get_settings_data": "volume_segmantics_amd.data.settings_data",
    "TrainingDataSlicer": "volume_segmantics_amd.data.slicers",
    "Quality": "volume_segmantics_amd.utilities.base_data_utils",
    "Axis": "volume_segmantics_amd.utilities.base_data_utils",
    "ModelType": "volume_segmantics_amd.utilities.base_data_utils",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        return getattr(importlib.import_module(_LAZY[name]), name)
    raise AttributeError(name)
