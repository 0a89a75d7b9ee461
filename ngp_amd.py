"""Import alias: the product package lives in `instant-ngp-pp_amd/` (a directory name Python
cannot spell in an `import` statement), so `import ngp_amd` loads it through importlib and
re-exports it.  `ngp_amd.vren`, `ngp_amd.tinycudann`, `ngp_amd.rendering`, ... are the
sub-modules."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_pkg = importlib.import_module("instant-ngp-pp_amd")
sys.modules[__name__] = _pkg
