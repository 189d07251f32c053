"""Helpers shared by the CPU and GPU tests of tests/golden/fgn_glue.npz (the reference's own fgn.py glue)."""
import os

import numpy as np


def glue_inputs():
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden_fgn', os.path.join(
        os.path.dirname(os.path.abspath(__file__)), 'golden', 'make_golden_fgn.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)            # pure helpers at import time: nothing reads /root/reference
    return m.make_inputs()


def glue_expected(z, prefix='server__'):
    """The result dicts stored by make_golden_fgn.py as a list of {key: ndarray | list of RLE dicts}."""
    out = []
    for i in range(int(z[prefix + 'n_out'])):
        one = {}
        for k in z[prefix + 'out_keys']:
            k = str(k)
            if k.endswith('_rle'):
                one[k] = [{'size': z[f'{prefix}out{i}__{k}__{j}__size'].tolist(),
                           'counts': z[f'{prefix}out{i}__{k}__{j}__counts'].tobytes()}
                          for j in range(int(z[f'{prefix}out{i}__{k}__n']))]
            else:
                one[k] = z[f'{prefix}out{i}__{k}']
        out.append(one)
    return out


def assert_results_equal(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert list(g) == list(w), (list(g), list(w))            # same keys in the same order
        for k in w:
            if k.endswith('_rle'):
                assert g[k] == w[k], k
            else:
                assert isinstance(g[k], np.ndarray) and g[k].dtype == w[k].dtype and g[k].shape == w[k].shape, \
                    (k, g[k].dtype, w[k].dtype, g[k].shape, w[k].shape)
                assert np.array_equal(g[k], w[k]), k
