"""Loading exported models.

The reference's inference scripts take a yaml descriptor written by scripts/export_model.py:182-194
(`model`, `model_quantized`, `norms`, `padding_factor`, `thing_list`, `labels`, `class_names`, `FINETUNE`) whose
`model` entry is a TorchScript archive of the scripted (Quantizable)Panoptic{DeepLab,BiFPN}[PR] module
(scripts/pdl_inference3d.py:69-72: `torch.jit.load`), or they rebuild `models.__dict__[arch]` and load a training
checkpoint `{'state_dict', 'norms', 'run_id'}` with the `module.` prefix stripped
(scripts/inference3d_multigpu.py:288-300).

A TorchScript archive runs through the TorchScript interpreter with library kernels only.  To run it on this
repository's kernels the archive is used for what it is on this path -- a container of named tensors: its state dict
is read, the architecture is inferred from the archive's class name and the shapes of its convolution weights (the
descriptor does not name the encoder), this package's module of the same architecture is built, loaded with
`strict=True`, and handed to `prepare_for_inference`.

Two key layouts occur.  A training checkpoint (and an archive scripted without fusing) has the training classes' keys,
which equal this package's.  scripts/export_model.py:115-120 calls `model.fuse_model()` before `torch.jit.script`:
`torch.quantization.fuse_modules` folds every eval-mode BatchNorm of the encoder, the ASPP / first projection of the
decoders and the PointRend MLP into the preceding convolution (folded weight + a new conv bias; the BatchNorm becomes
an Identity and its tensors disappear) and wraps conv + ReLU pairs in a `ConvReLU2d`, which moves the convolution's
keys one level down (`encoder.conv1.weight` -> `encoder.conv1.0.weight`).  `unfuse_state_dict` maps such a state dict
back onto this package's modules: the wrapper level is stripped, and a BatchNorm whose tensors are gone is loaded as
the identity plus the folded bias (weight 1, bias = the convolution's folded bias, running_mean 0, running_var 1 and
eps 1e-30 -- torch insists on eps > 0, and 1 + 1e-30 == 1 -- so that it computes x + bias exactly).
"""
import os

import torch

from . import panoptic_bifpn as BF
from . import panoptic_deeplab as DL

__all__ = ['load_exported', 'load_checkpoint', 'infer_architecture', 'model_from_state_dict', 'unfuse_state_dict']

_CLASSES = {'PanopticDeepLab': DL.PanopticDeepLab, 'PanopticDeepLabPR': DL.PanopticDeepLabPR,
            'PanopticBiFPN': BF.PanopticBiFPN, 'PanopticBiFPNPR': BF.PanopticBiFPNPR}


def _shapes(sd):
    return {k: tuple(v.shape) for k, v in sd.items()}


def _conv_key(want, key):
    """name under which the weight / bias `key` of one of this package's convolutions is stored in `want`: the key
    itself, or one level down inside the ConvReLU wrapper fuse_modules puts around a conv + ReLU pair"""
    if key in want:
        return key
    stem, leaf = key.rsplit('.', 1)
    wrapped = f'{stem}.0.{leaf}'
    return wrapped if wrapped in want else None


def _conv_weights(shapes):
    return {k: v for k, v in shapes.items() if k.endswith('.weight') and len(v) >= 3}


def _bn_partners(model):
    """{BatchNorm module name: name of the convolution it normalises} from the module tree: the convolution is the
    BatchNorm's previous sibling (conv1, bn1, conv2, bn2, ... of a bottleneck; (conv, bn[, relu]) of a Sequential) or,
    if that sibling is a block, the last convolution inside it"""
    convs = (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.ConvTranspose2d)
    out = {}
    for pname, parent in model.named_modules():
        prev = None
        for cname, child in parent.named_children():
            full = f'{pname}.{cname}' if pname else cname
            if isinstance(child, torch.nn.modules.batchnorm._BatchNorm) and prev is not None:
                out[full] = prev
            if isinstance(child, convs):
                prev = full
            else:
                inner = [n for n, m in child.named_modules() if isinstance(m, convs)]
                if inner:
                    prev = f'{full}.{inner[-1]}'
    return out


def unfuse_state_dict(model, state_dict):
    """`state_dict` in either layout (module docstring) -> (state dict with exactly `model`'s keys, names of the
    BatchNorm modules that were folded away and must compute x + bias: the caller makes their eps vanish)."""
    own = model.state_dict()
    partners = _bn_partners(model)
    out, folded, used = {}, [], set()
    for key, ref in own.items():
        stem, leaf = key.rsplit('.', 1)
        src = _conv_key(state_dict, key)
        if src is not None:
            out[key] = state_dict[src]
            used.add(src)
            continue
        if stem in partners:                                     # a BatchNorm that fuse_modules folded away
            bias = _conv_key(state_dict, partners[stem] + '.bias')
            if bias is None:
                raise KeyError(f'{key}: neither the BatchNorm tensor nor a folded bias of {partners[stem]} is in the archive')
            used.add(bias)
            if stem not in folded:
                folded.append(stem)
            b = state_dict[bias]
            out[key] = {'weight': torch.ones_like(b), 'bias': b, 'running_mean': torch.zeros_like(b),
                        'running_var': torch.ones_like(b),
                        'num_batches_tracked': torch.zeros((), dtype=torch.long)}[leaf].to(ref.dtype)
            continue
        raise KeyError(f'{key} is not in the archive (fused or not)')
    extra = sorted(set(state_dict) - used)
    if extra:
        raise KeyError(f'archive tensors without a place in {type(model).__name__}: {extra[:5]}{" ..." if len(extra) > 5 else ""}')
    return out, folded


def infer_architecture(state_dict, class_name=None):
    """(arch name, constructor kwargs) whose module has exactly the convolutions (names up to the ConvReLU wrapper
    level, shapes) of `state_dict`.  Convolution weights only: BatchNorm tensors are absent from a fused export.
    class_name: the archive's class (`original_name`, with or without the `Quantizable` prefix), if known."""
    want = _shapes(state_dict)

    def shape(key):
        k = _conv_key(want, key)
        if k is None:
            raise KeyError(key)
        return want[k]

    has_pr = any(k.startswith('semantic_pr.') for k in want)
    is_bifpn = any(k.startswith('semantic_fpn.') for k in want)
    arch = ('PanopticBiFPN' if is_bifpn else 'PanopticDeepLab') + ('PR' if has_pr else '')
    if class_name:
        named = class_name.replace('Quantizable', '')
        if named in _CLASSES and named != arch:
            raise ValueError(f"archive class {class_name} does not match its tensors (look like {arch})")
    num_classes = shape('semantic_head.head.1.weight')[0]
    base = dict(num_classes=num_classes, ins_decoder=any(k.startswith(('instance_decoder.', 'instance_fpn.')) for k in want))
    if is_bifpn:
        base['fpn_dim'] = shape('semantic_head.head.1.weight')[1]
        base['fpn_layers'] = 1 + max(int(k.split('.')[2]) for k in want if k.startswith('semantic_fpn.bifpns.')) \
            if any(k.startswith('semantic_fpn.bifpns.') for k in want) else 3
        encoders = list(BF.REGNETS) + list(DL._RESNETS)
    else:
        base['decoder_channels'] = shape('semantic_head.head.1.weight')[1]
        base['aspp_channels'] = shape('semantic_decoder.aspp.project.0.weight')[0]
        proj = sorted({int(k.split('.')[2]) for k in want if k.startswith('semantic_decoder.project.')})
        base['low_level_channels_project'] = tuple(shape(f'semantic_decoder.project.{i}.0.weight')[0] for i in proj)
        base['low_level_stages'] = tuple(range(len(proj), 0, -1))
        encoders = list(DL._RESNETS)
    n_convs = len(_conv_weights(want))
    errors = []
    for enc in encoders:
        kw = dict(base, encoder=enc)
        try:
            with torch.device('meta'):
                cand = _CLASSES[arch](**kw)
        except Exception as e:                     # a candidate the constructor rejects is simply not the one
            errors.append(f'{enc}: {e}')
            continue
        convs = _conv_weights(_shapes(cand.state_dict()))
        # aliased parameters (PanopticBiFPN's shared after_combines) are listed once per alias on both sides
        if len(convs) == n_convs and all(_conv_key(want, k) is not None and want[_conv_key(want, k)] == v
                                         for k, v in convs.items()):
            return arch, kw
    raise ValueError(f"no {arch} configuration of this package has the archive's {n_convs} convolutions "
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
    sd, folded = unfuse_state_dict(model, sd)
    model.load_state_dict(sd, strict=True)
    mods = dict(model.named_modules())
    for name in folded:
        mods[name].eps = 1e-30                      # weight 1, mean 0, var 1 (+ 1e-30 == 1): exactly x + folded bias
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
