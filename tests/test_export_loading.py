"""CPU: loading exported models (empanada_amd/models/export.py) -- a TorchScript archive + yaml descriptor as
scripts/export_model.py:182-194 writes them, and a training checkpoint as scripts/inference3d_multigpu.py:288-300 reads
it.  The archive is treated as a container of named tensors: architecture inferred from names and shapes, this
package's module built and loaded strictly; outputs must equal the archive's own forward."""
import os
import sys

import pytest
import torch
import yaml

from empanada_amd import models
from empanada_amd.models import export as EX


def _descriptor(tmp_path, archive, **extra):
    desc = {'model': str(archive), 'model_quantized': None, 'norms': {'mean': 0.508979, 'std': 0.148561},
            'padding_factor': 128, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'FINETUNE': {}}
    desc.update(extra)
    p = tmp_path / 'PanopticDeepLab_test.yaml'
    p.write_text(yaml.safe_dump(desc))
    return str(p)


@pytest.mark.parametrize('arch,kw', [
    ('PanopticDeepLab', dict(encoder='resnet18', num_classes=1)),
    ('PanopticDeepLab', dict(encoder='resnet34', num_classes=3, decoder_channels=128, aspp_channels=96,
                             low_level_channels_project=(64, 32, 16))),
    ('PanopticBiFPN', dict(encoder='regnety_200mf', num_classes=2, fpn_dim=64)),
    ('PanopticBiFPNPR', dict(encoder='resnet18', num_classes=1, fpn_dim=32)),
    ('PanopticDeepLabPR', dict(encoder='resnet18', num_classes=1, ins_decoder=True)),
])
def test_architecture_is_inferred_from_names_and_shapes(arch, kw):
    m = models.__dict__[arch](**kw)
    got_arch, got_kw = EX.infer_architecture(m.state_dict(), 'Quantizable' + arch)
    assert got_arch == arch
    for k, v in kw.items():
        assert got_kw[k] == v, (k, got_kw[k], v)
    m2, _, _ = EX.model_from_state_dict({'module.' + k: v for k, v in m.state_dict().items()})
    assert type(m2).__name__ == arch


def test_torchscript_archive_round_trip(tmp_path):
    m = models.synthesize_weights(models.PanopticDeepLab(encoder='resnet18', num_classes=1)).eval()
    x = torch.randn(1, 1, 64, 64)
    traced = torch.jit.trace(m, x, strict=False)
    path = tmp_path / 'PanopticDeepLab_test.pth'
    torch.jit.save(traced, str(path))
    model, desc = EX.load_exported(_descriptor(tmp_path, path), device='cpu', prepare=False)
    assert desc['arch'] == 'PanopticDeepLab' and desc['model_kwargs']['encoder'] == 'resnet18'
    assert desc['norms']['mean'] == 0.508979 and desc['padding_factor'] == 128
    with torch.no_grad():
        ref, out = m(x), model(x)
    for k in ref:
        assert torch.equal(ref[k], out[k])
    with pytest.raises(FileNotFoundError):
        EX.load_exported({'model': 'https://example.invalid/model.pth'}, device='cpu', prepare=False)


def test_training_checkpoint(tmp_path):
    m = models.synthesize_weights(models.PanopticBiFPN(encoder='regnety_200mf', num_classes=1, fpn_dim=32)).eval()
    p = tmp_path / 'ckpt.pth.tar'
    torch.save({'state_dict': {'module.' + k: v for k, v in m.state_dict().items()}, 'run_id': 'abc',
                'norms': {'mean': 0.5, 'std': 0.1}}, str(p))
    model, norms = EX.load_checkpoint(str(p), device='cpu', prepare=False)
    assert norms == {'mean': 0.5, 'std': 0.1} and type(model).__name__ == 'PanopticBiFPN'
    x = torch.randn(1, 1, 128, 128)
    with torch.no_grad():
        assert torch.equal(model(x)['offsets'], m(x)['offsets'])


@pytest.mark.skipif(not os.path.isdir('/root/reference/empanada'), reason='reference not present (GPU box)')
@pytest.mark.parametrize('layout', ['fused', 'plain'])
def test_archive_scripted_from_the_reference_class(tmp_path, layout):
    """build container only: the archive scripts/export_model.py writes -- the REFERENCE's
    QuantizablePanopticDeepLabPR, `fuse_model()` applied as the script does (layout 'fused': ConvReLU2d wrappers,
    BatchNorms folded into conv biases, 187 tensors instead of 423) or not ('plain'), scripted and saved in a scratch
    directory (never committed) -- loads into this package's PanopticDeepLabPR and both give the same heads through
    the exported 3-argument forward."""
    import subprocess
    code = r'''
import sys
sys.dont_write_bytecode = True
sys.path.insert(0, %r)
from oracle.gen_golden import _install_standins
_install_standins()
import warnings; warnings.filterwarnings('ignore')
import torch
from empanada.models.quantization import panoptic_deeplab as Q
from empanada_amd.models import synthesize_weights
m = synthesize_weights(Q.QuantizablePanopticDeepLabPR(encoder='resnet50', num_classes=1, quantize=False)).eval()
if sys.argv[3] == 'fused':
    m.fuse_model()                      # scripts/export_model.py:115-120: eval, fuse, then script
torch.jit.save(torch.jit.script(m), sys.argv[1])
x = torch.randn(1, 1, 96, 96, generator=torch.Generator().manual_seed(0))
with torch.no_grad():
    out = m(x, 2, False)
torch.save({k: v for k, v in out.items()}, sys.argv[2])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    arch_path, out_path = str(tmp_path / 'PanopticDeepLabPR_ref.pth'), str(tmp_path / 'out.pt')
    r = subprocess.run([sys.executable, '-c', code, arch_path, out_path, layout], capture_output=True, text=True, cwd='/tmp')
    assert r.returncode == 0, r.stderr[-3000:]
    model, desc = EX.load_exported(_descriptor(tmp_path, arch_path), device='cpu', prepare=False)
    assert desc['arch'] == 'PanopticDeepLabPR' and desc['model_kwargs']['encoder'] == 'resnet50'
    ref = torch.load(out_path, weights_only=True)
    x = torch.randn(1, 1, 96, 96, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        out = model(x, 2, False)
    for k in ref:
        assert out[k].shape == ref[k].shape
        assert (out[k] - ref[k]).abs().max() <= 1e-5 * max(1.0, float(ref[k].abs().max())), k
