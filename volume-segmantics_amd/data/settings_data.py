"""YAML / dict -> SimpleNamespace (volume_segmantics/data/settings_data.py:10-27); the settings surface
(volseg-settings/*.yaml keys) is unchanged, new keys are optional."""
import logging
import sys
from pathlib import Path
from types import SimpleNamespace
from typing import Union

import yaml


def get_settings_data(data: Union[Path, dict, None]) -> SimpleNamespace:
    if data is None:
        return SimpleNamespace()
    if isinstance(data, dict):
        return SimpleNamespace(**data)
    if isinstance(data, Path):
        logging.info(f"Loading settings from {data}")
        if not data.exists():
            logging.error("Couldn't find settings file... Exiting!")
            sys.exit(1)
        with open(data, "r") as stream:
            return SimpleNamespace(**yaml.safe_load(stream))
    raise TypeError(f"settings must be a Path, dict or None, got {type(data).__name__}")
