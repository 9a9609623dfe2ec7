"""Loading exported models.

The reference's inference scripts take a yaml descriptor written by scripts/export_model.py:182-194
(`model`, `model_quantized`, `norms`, `padding_factor`, `thing_list`, `labels`, `class_names`, `FINETUNE`) whose
`model` entry is a TorchScript archive of the scripted (Quantizable)Panoptic{DeepLab,BiFPN}[PR] module
(scripts/pdl_inference3d.py:69-72: `torch.jit.load`), or they rebuild `models.__dict__[arch]` and load a training
checkpoint `{'state_dict', 'norms', 'run_id'}` with the `module.` prefix stripped
(scripts/inference3d_multigpu.py:288-300).

A TorchScript archive runs through the TorchScript interpreter with library kernels only.  To run it on this
repository's kernels the archive is used for what it is on this path -- a container of named tensors: its state dict
is read (state-dict keys of the exported classes equal the training classes', and both equal this package's), the
architecture is inferred from the archive's class name and tensor shapes (the descriptor does not name the encoder),
this package's module of the same architecture is built, loaded with `strict=True`, and handed to
`prepare_for_inference`.
"""
import os

import torch

from . import panoptic_bifpn as BF
from . import panoptic_deeplab as DL

__all__ = ['load_exported', 'load_checkpoint', 'infer_architecture', 'model_from_state_dict']

_CLASSES = {'PanopticDeepLab': DL.PanopticDeepLab, 'PanopticDeepLabPR': DL.PanopticDeepLabPR,
            'PanopticBiFPN': BF.PanopticBiFPN, 'PanopticBiFPNPR': BF.PanopticBiFPNPR}


def _shapes(sd):
    return {k: tuple(v.shape) for k, v in sd.items()}


def infer_architecture(state_dict, class_name=None):
    """(arch name, constructor kwargs) whose module has exactly the keys and shapes of `state_dict`.
    class_name: the archive's class (`original_name`, with or without the `Quantizable` prefix), if known."""
    want = _shapes(state_dict)
    has_pr = any(k.startswith('semantic_pr.') for k in want)
    is_bifpn = any(k.startswith('semantic_fpn.') for k in want)
    arch = ('PanopticBiFPN' if is_bifpn else 'PanopticDeepLab') + ('PR' if has_pr else '')
    if class_name:
        named = class_name.replace('Quantizable', '')
        if named in _CLASSES and named != arch:
            raise ValueError(f"archive class {class_name} does not match its tensors (look like {arch})")
    num_classes = want['semantic_head.head.1.weight'][0]
    base = dict(num_classes=num_classes, ins_decoder=any(k.startswith(('instance_decoder.', 'instance_fpn.')) for k in want))
    if is_bifpn:
        base['fpn_dim'] = want['semantic_head.head.1.weight'][1]
        base['fpn_layers'] = 1 + max(int(k.split('.')[2]) for k in want if k.startswith('semantic_fpn.bifpns.')) \
            if any(k.startswith('semantic_fpn.bifpns.') for k in want) else 3
        encoders = list(BF.REGNETS) + list(DL._RESNETS)
    else:
        base['decoder_channels'] = want['semantic_head.head.1.weight'][1]
        base['aspp_channels'] = want['semantic_decoder.aspp.project.0.weight'][0]
        proj = sorted(int(k.split('.')[2]) for k in want if k.startswith('semantic_decoder.project.') and k.endswith('.0.weight'))
        base['low_level_channels_project'] = tuple(want[f'semantic_decoder.project.{i}.0.weight'][0] for i in proj)
        base['low_level_stages'] = tuple(range(len(proj), 0, -1))
        encoders = list(DL._RESNETS)
    errors = []
    for enc in encoders:
        kw = dict(base, encoder=enc)
        try:
            with torch.device('meta'):
                cand = _CLASSES[arch](**kw)
        except Exception as e:                     # a candidate the constructor rejects is simply not the one
            errors.append(f'{enc}: {e}')
            continue
        if _shapes(cand.state_dict()) == want:
            return arch, kw
    raise ValueError(f"no {arch} configuration of this package has the archive's {len(want)} tensors "
                     f"(tried encoders {encoders}); pass the constructor arguments explicitly")


def model_from_state_dict(state_dict, class_name=None, arch=None, **model_kwargs):
    """module of this package holding `state_dict` (strict); arch / model_kwargs skip the inference"""
    sd = {k[len('module.'):] if k.startswith('module.') else k: v for k, v in state_dict.items()}
    if arch is None:
        arch, kw = infer_architecture(sd, class_name)
        kw.update(model_kwargs)
    else:
        kw = dict(model_kwargs)
    model = _CLASSES[arch](**kw)
    model.load_state_dict(sd, strict=True)
    return model.eval(), arch, kw


def load_exported(descriptor, device='cuda', prepare=True, **model_kwargs):
    """descriptor: path of the yaml written by scripts/export_model.py (or the dict) -> (model, descriptor dict).
    The model is this package's module with the archive's weights; with prepare=True it is moved to `device` and set
    up for inference (`prepare_for_inference`: NHWC, fused HIP kernels)."""
    from ..config_utils import load_config
    desc = load_config(descriptor) if isinstance(descriptor, (str, os.PathLike)) else dict(descriptor)
    path = desc['model']
    if not os.path.isfile(path):
        raise FileNotFoundError(f"{path}: the descriptor's `model` must be a local TorchScript archive "
                                "(there is no network: model URLs cannot be fetched)")
    archive = torch.jit.load(path, map_location='cpu')
    model, arch, kw = model_from_state_dict(archive.state_dict(), getattr(archive, 'original_name', None), **model_kwargs)
    desc['arch'], desc['model_kwargs'] = arch, kw
    if prepare:
        model = DL.prepare_for_inference(model, device)
    return model, desc


def load_checkpoint(path, arch=None, device='cuda', prepare=True, **model_kwargs):
    """training checkpoint {'state_dict', 'norms', 'run_id'} (scripts/inference3d_multigpu.py:288-300; `module.`
    prefixes stripped, strict) -> (model, norms).  The file must hold tensors only (`weights_only=True`)."""
    ckpt = torch.load(path, map_location='cpu', weights_only=True)
    model, arch, kw = model_from_state_dict(ckpt['state_dict'], None, arch, **model_kwargs)
    if prepare:
        model = DL.prepare_for_inference(model, device)
    return model, ckpt.get('norms')
