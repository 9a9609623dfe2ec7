"""top-level `config_utils` of scripts/inference3d_multigpu.py:33 (absent from the reference)"""
from empanada_amd.config_utils import *      # noqa: F401,F403
from empanada_amd.config_utils import load_config, load_inference_config, load_train_config      # noqa: F401
