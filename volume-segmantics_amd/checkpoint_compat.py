"""Interchange of ``.pytorch`` checkpoints with the reference.

``EarlyStopping.save_checkpoint`` pickles ``model_struc_dict["type"]`` as the Enum object
``volume_segmantics.utilities.base_data_utils.ModelType`` (reference utilities/early_stopping.py:50-63).  To read
such files without the reference installed, and to write files the reference can read, this module registers
this package's enums under the reference's module path when (and only when) the real package is absent."""
import importlib.util
import sys
import types


def install_reference_aliases() -> bool:
    name = "volume_segmantics.utilities.base_data_utils"
    if name in sys.modules:
        return False
    try:
        if importlib.util.find_spec("volume_segmantics") is not None:
            importlib.import_module(name)   # the real reference is installed: import it so that pickle (and
            return False                    # reference_pickle_enum) find its enums in sys.modules
    except Exception:
        # absent, or installed without its third-party dependencies (h5py, smp, ...): drop whatever the failed import
        # left behind and register the aliases instead
        for k in [k for k in sys.modules if k == "volume_segmantics" or k.startswith("volume_segmantics.")]:
            del sys.modules[k]
    from .utilities import base_data_utils as ours
    root = types.ModuleType("volume_segmantics")
    util = types.ModuleType("volume_segmantics.utilities")
    bdu = types.ModuleType("volume_segmantics.utilities.base_data_utils")
    for name in ("ModelType", "Quality", "Axis"):
        enum = getattr(ours, name)
        enum.__module__ = "volume_segmantics.utilities.base_data_utils"  # pickles name the reference's path
        setattr(bdu, name, enum)
    root.utilities, util.base_data_utils = util, bdu
    root.__path__, util.__path__ = [], []
    sys.modules.update({"volume_segmantics": root, "volume_segmantics.utilities": util,
                        "volume_segmantics.utilities.base_data_utils": bdu})
    return True


def reference_pickle_enum(model_type):
    """ModelType member whose pickle names the reference's module path (for checkpoints the reference will read)."""
    install_reference_aliases()
    ref = sys.modules["volume_segmantics.utilities.base_data_utils"].ModelType
    return ref[model_type.name]
