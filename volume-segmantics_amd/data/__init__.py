from .settings_data import get_settings_data  # noqa: F401
