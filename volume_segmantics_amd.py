"""Import shim: the package directory is ``volume-segmantics_amd/`` (not a valid Python identifier),
so this module exposes it under the importable name ``volume_segmantics_amd``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "volume-segmantics_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _os, _f
