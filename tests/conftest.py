import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def csr_from(z, prefix):
    shape = tuple(int(s) for s in z[prefix + '_shape'])
    return sp.csr_matrix((z[prefix + '_data'], z[prefix + '_indices'], z[prefix + '_indptr']), shape=shape)


def assert_csr_equal(A, B, exact=True, rtol=0.0, atol=0.0):
    A, B = sp.csr_matrix(A).copy(), sp.csr_matrix(B).copy()
    A.sort_indices(); B.sort_indices()
    assert A.shape == B.shape
    assert np.array_equal(A.indptr, B.indptr)
    assert np.array_equal(A.indices, B.indices)
    if exact:
        assert np.array_equal(A.data, B.data)
    else:
        np.testing.assert_allclose(A.data, B.data, rtol=rtol, atol=atol)


@pytest.fixture(scope='session')
def golden():
    return load_golden
