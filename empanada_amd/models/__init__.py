"""Dense path (D1): encoder-decoder models with the reference's constructor arguments, module tree and
state-dict keys (``empanada/models``), so that reference checkpoints load with ``strict=True``.
The conv stacks stay in PyTorch-ROCm (MIOpen / hipBLASLt on MFMA); see ``prepare_for_inference``.
"""
from .panoptic_deeplab import PanopticDeepLab, prepare_for_inference, synthesize_weights

__all__ = ['PanopticDeepLab', 'prepare_for_inference', 'synthesize_weights']
