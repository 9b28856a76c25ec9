"""Importable alias of the package directory `fhe-study_amd/` (hyphenated by repo
convention, so `import fhe-study_amd` is not valid Python syntax)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("fhe-study_amd")
sys.modules[__name__] = _pkg
