"""Public classes of the path (volume_segmantics/model/__init__.py:1-6), imported lazily."""


def __getattr__(name):
    if name == "VolSeg2dTrainer":
        from .operations.vol_seg_2d_trainer import VolSeg2dTrainer
        return VolSeg2dTrainer
    if name == "VolSeg2DPredictionManager":
        from .operations.vol_seg_prediction_manager import VolSeg2DPredictionManager
        return VolSeg2DPredictionManager
    if name == "VolSeg2dPredictor":
        from .operations.vol_seg_2d_predictor import VolSeg2dPredictor
        return VolSeg2dPredictor
    raise AttributeError(name)
