"""`import empanada` resolved by this repository.

Put `<repo>/compat` (and `<repo>`) on PYTHONPATH and the reference's scripts (scripts/pdl_inference3d.py,
scripts/inference3d_multigpu.py) import unchanged: every `empanada.<module>` is the module of the same name in
``empanada_amd`` -- one module object under both names -- and the few names whose path differs are mapped below.
Nothing of the reference is in here; it is a name table.
"""
import importlib
import importlib.abc
import importlib.util
import sys

_IMPL = 'empanada_amd'
# reference module path -> module of this repository that provides its names
_RENAMED = {
    'empanada.config_loaders': 'empanada_amd.config_utils',          # load_config, read_yaml
    'empanada.data': 'empanada_amd.data',                            # VolumeDataset
    'empanada.data.volume_dataset': 'empanada_amd.data',
    'empanada.evaluation.evaluator': 'empanada_amd.evaluation',
}


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != 'empanada' and not fullname.startswith('empanada.'):
            return None
        real = _RENAMED.get(fullname, _IMPL + fullname[len('empanada'):])
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        spec = importlib.util.spec_from_loader(fullname, self, is_package=True)
        spec._emp_real = real
        return spec

    def create_module(self, spec):
        return importlib.import_module(spec._emp_real)          # the same module object under both names

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _Alias())
_impl = importlib.import_module(_IMPL)
for _name in ('models', 'inference', 'array_utils', 'consensus', 'zarr_utils', 'evaluation', 'aggregation', 'data',
              'config_utils', 'sampler'):
    globals()[_name] = importlib.import_module(f'{_IMPL}.{_name}')
__path__ = []                                                    # a package: submodules come from the finder above
