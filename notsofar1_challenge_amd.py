"""Import alias: ``import notsofar1_challenge_amd`` -> the package directory ``notsofar1-challenge_amd/``.

The package directory carries the repository's name (with a hyphen, which Python's ``import``
statement cannot spell).  This shim makes the package -- and every submodule -- importable under a
valid identifier, always resolving to the ONE module object of the real package (so that e.g. ctypes
structure classes are not duplicated under two module names).
"""
import importlib
import importlib.abc
import importlib.util
import sys

_ALIAS = "notsofar1_challenge_amd"
_REAL = "notsofar1-challenge_amd"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _ALIAS or fullname.startswith(_ALIAS + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        real = importlib.import_module(_REAL + spec.name[len(_ALIAS):])
        return real

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

_real_pkg = importlib.import_module(_REAL)
sys.modules[_ALIAS] = _real_pkg
