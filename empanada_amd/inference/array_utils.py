"""`empanada.inference.array_utils` is star-imported by scripts/inference3d_multigpu.py:28 but does not exist in
the reference; it is an alias of `empanada.array_utils`."""
from ..array_utils import *  # noqa: F401,F403
from ..array_utils import __all__  # noqa: F401
