from .base_data_utils import Axis, ModelType, Quality  # noqa: F401
