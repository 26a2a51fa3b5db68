"""Importable alias of the package directory ``oct-image-segmentation-models_amd/``.

The product package lives in the hyphenated directory the project layout asks
for; a hyphen cannot appear in a Python module name, so this shim points the
package search path there and runs its ``__init__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "oct-image-segmentation-models_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
