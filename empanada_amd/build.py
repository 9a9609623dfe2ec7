"""Build libemp_hip.so (gfx950) in-tree with hipcc.  `python -m empanada_amd.build`.

hipcc cross-compiles without a GPU.  The built .so is git-ignored but travels to the GPU box.
-ffp-contract=off: the only fused multiply-add on the path is the explicit one in group_pixels.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['emp_pixel.hip', 'emp_runs.hip', 'emp_ranges.hip', 'emp_tracks.hip', 'emp_dense.hip', 'emp_conv.hip', 'emp_conv1x1.hip', 'emp_gconv.hip', 'emp_stem.hip', 'emp_pointrend.hip', 'emp_chain.cpp']
LIB = os.path.join(HERE, 'libemp_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-std=c++17', '-Wno-unused-value',
         '-Wno-unused-result', '-Wno-inline-asm']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', 'hipcc')
    headers = [os.path.join(CSRC, 'emp_common.h'), os.path.join(HERE, '..', 'include', 'emp_hip.h')]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + '.o')
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
