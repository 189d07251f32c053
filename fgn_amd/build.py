"""Build libfgn_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the
library is a plain C-ABI shared object loaded with ctypes (include/fgn_hip.h)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libfgn_hip.so')

# conv_igemm: MFMA kernel, default fp contraction.  Everything else is on (or next to)
# the bit-exact selection path: no mul+add fusion, so fp32 op order is the oracle's.
SOURCES = [
    # (the atomic optimizer would turn the tile scheduler's one-lane atomic into a wave reduction that reads the returned
    # value at once: the pull could no longer fly under the K loop)
    ('conv_igemm.hip', ['-mllvm', '-amdgpu-atomic-optimizer-strategy=None']),
    ('abi.hip', []),
    ('spatial.hip', ['-ffp-contract=off']),
    ('norm.hip', []),
    ('winograd.hip', []),
    ('relation.hip', []),
    ('rpn_post.hip', ['-ffp-contract=off']),
    ('det_post.hip', ['-ffp-contract=off']),
    ('mask.hip', ['-ffp-contract=off']),
    ('train.hip', ['-ffp-contract=off']),
    ('train_bwd.hip', []),
]
COMMON = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    objs = []
    procs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + COMMON + extra + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
