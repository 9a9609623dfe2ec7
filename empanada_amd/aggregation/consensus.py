"""`empanada.aggregation.consensus` is imported by scripts/inference3d_multigpu.py:30 but does not exist in the
reference; this module provides the name it expects."""
from ..consensus import merge_objects_from_trackers as merge_objects3d  # noqa: F401
from ..consensus import *  # noqa: F401,F403
