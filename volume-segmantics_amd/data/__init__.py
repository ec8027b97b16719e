from .settings_data import get_settings_data  # noqa: F401
from .slicers import TrainingDataSlicer  # noqa: F401

__all__ = ["get_settings_data", "TrainingDataSlicer"]
