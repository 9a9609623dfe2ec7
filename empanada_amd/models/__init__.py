"""Dense path (D1): encoder-decoder models with the reference's constructor arguments, module tree and
state-dict keys (``empanada/models``), so that reference checkpoints load with ``strict=True``.
The conv stacks stay in PyTorch-ROCm (MIOpen / hipBLASLt on MFMA); see ``prepare_for_inference``.

``models.__dict__[arch](**config['MODEL'])`` works like in the reference (scripts/inference3d_multigpu.py:288).
"""
from .export import load_checkpoint, load_exported, model_from_state_dict
from .graphed import GraphedForward
from .panoptic_bifpn import PanopticBiFPN, PanopticBiFPNPR
from .panoptic_deeplab import (PanopticDeepLab, PanopticDeepLabPR, fuse_bn_act, prepare_for_inference,
                               synthesize_weights, tune_fused_convs)

__all__ = ['PanopticDeepLab', 'PanopticDeepLabPR', 'PanopticBiFPN', 'PanopticBiFPNPR', 'prepare_for_inference',
           'synthesize_weights', 'fuse_bn_act', 'tune_fused_convs', 'GraphedForward', 'load_exported', 'load_checkpoint',
           'model_from_state_dict']
