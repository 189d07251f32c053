"""INTEGRATION.md's Level-2 section is executable: every ```python block under "## Level 2" is extracted and run as
written against the built libfgn_hip.so (the binding a maintainer pastes into the reference), and the RoIAlign it
binds is compared with the oracle's mmcv-flavoured RoIAlign (the call at fgn_roi_head.py:331).  The CPU half checks
the arity of every `argtypes` list the blocks declare against include/fgn_hip.h without touching a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = os.path.join(ROOT, 'INTEGRATION.md')
HEADER = os.path.join(ROOT, 'include', 'fgn_hip.h')
LIB = os.path.join(ROOT, 'fgn_amd', 'libfgn_hip.so')


def _level2_blocks():
    text = open(DOC).read()
    start = text.index('## Level 2')
    nxt = text.find('\n## ', start + 1)
    section = text[start:nxt if nxt > 0 else len(text)]
    blocks = re.findall(r'^```python\n(.*?)^```', section, flags=re.S | re.M)
    assert blocks, 'INTEGRATION.md Level 2 holds no python block'
    return blocks


def _header_arity():
    """name -> number of parameters of every `int|size_t fgn_*(...)` prototype of the header"""
    src = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|size_t)\s+(fgn_\w+)\s*\(([^)]*)\)\s*;', src):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ('', 'void') else len(args.split(','))
    return out


class _Recorder:
    """stands in for ctypes.CDLL on the CPU: records restype / argtypes assignments per symbol"""

    def __init__(self):
        object.__setattr__(self, 'syms', {})

    def __getattr__(self, name):
        return self.syms.setdefault(name, type('Sym', (), {})())


def test_doc_bindings_have_the_header_arity(monkeypatch):
    arity = _header_arity()
    assert arity['fgn_roi_align_nhwc_f32'] == 16
    import torch  # noqa: F401  (its own import binds other libraries through ctypes: before the patch)
    rec = _Recorder()
    real = ctypes.CDLL
    monkeypatch.setattr(ctypes, 'CDLL', lambda name, *a, **k: rec if 'fgn_hip' in str(name) else real(name, *a, **k))
    for block in _level2_blocks():
        exec(compile(block, DOC, 'exec'), {'__name__': 'fgn_hip_binding'})
    assert rec.syms, 'the blocks bind nothing'
    for name, sym in rec.syms.items():
        assert name in arity, f'INTEGRATION.md binds {name}, which include/fgn_hip.h does not declare'
        assert len(sym.argtypes) == arity[name], (name, len(sym.argtypes), arity[name])
        assert sym.restype is ctypes.c_int


@pytest.mark.gpu
def test_doc_blocks_run_against_the_library_and_match_the_oracle(monkeypatch):
    import torch
    from oracle import fgn_ref_cpu as O
    monkeypatch.setenv('FGN_HIP_LIB', LIB)
    ns = {'__name__': 'fgn_hip_binding'}
    for block in _level2_blocks():
        exec(compile(block, DOC, 'exec'), ns)
    g = torch.Generator().manual_seed(3)
    fmap = torch.randn(2, 64, 20, 31, generator=g)                       # NCHW for the oracle
    rois = torch.tensor([[0, 8.3, 5.1, 200.7, 150.2], [1, 0, 0, 496, 320], [0, 100, 100, 101, 100.5],
                         [1, -20, -30, 40, 50], [0, 300, 200, 600, 400], [1, 17.5, 33.25, 18.0, 300.0]],
                        dtype=torch.float32)
    ref = O.roi_align(fmap, rois.numpy(), 7, 1 / 16, 0, True)            # mmcv RoIAlign(aligned=True, sampling_ratio=0)
    got = ns['roi_align_nhwc'](fmap.permute(0, 2, 3, 1).contiguous().cuda(), rois.cuda())
    torch.cuda.synchronize()
    got = got.permute(0, 3, 1, 2).cpu()
    d = (got - ref).abs().max().item()
    assert d <= 2e-6 * ref.abs().max().item(), d                        # fp32 sums of <= 4*g*g bilinear terms
    # the product's own wrapper gives the same bytes: the doc binds the entry point the detector uses
    from fgn_amd import ops
    mine = ops.roi_align(fmap.permute(0, 2, 3, 1).contiguous().cuda(), rois.cuda(), 7, 1 / 16, 0, True)
    assert torch.equal(mine.permute(0, 3, 1, 2).cpu(), got)
