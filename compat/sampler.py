"""top-level `sampler` of scripts/inference3d_multigpu.py:34 (absent from the reference)"""
from empanada_amd.sampler import ContiguousShardSampler, DistributedEvalSampler      # noqa: F401
