"""Import alias: `import stochastic_muzero_amd` -> the package in ./stochastic-muzero_amd/ (hyphenated directory)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("stochastic-muzero_amd")
sys.modules[__name__] = _pkg
