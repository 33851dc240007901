"""Oracle (test infrastructure, see oracle/__init__.py): graph operators.

Restates, in NumPy/SciPy:
* ``laplacian``  -- lib_new/graph.py:79-98
* ``rescale_L``  -- lib_new/graph.py:146-152 (as called from
                    lib_new/models_gcn.py:590-592 with lmax=2)
* ``chebyshev``  -- lib_new/graph.py:155-172, the reference's own NumPy twin
                    of the TF recurrence in models_gcn.py:598-610
"""
import numpy as np
import scipy.sparse as sp


def laplacian(W, normalized=True):
    """L = D - W, or I - D^-1/2 W D^-1/2 (graph.py:79-98).

    Degrees are *column* sums (graph.py:83); the normalised form adds
    ``np.spacing(0)`` of W's dtype before the inverse square root
    (graph.py:90-91), so isolated vertices get a huge-but-finite scale that is
    then multiplied by their all-zero row/column.  The result dtype follows W.
    """
    W = sp.csr_matrix(W)
    deg = np.asarray(W.sum(axis=0)).ravel()
    if not normalized:
        return sp.csr_matrix(sp.diags(deg, 0) - W)
    deg = deg + np.spacing(np.array(0, W.dtype))
    scale = (1 / np.sqrt(deg)).astype(W.dtype, copy=False)
    Dm = sp.diags(scale, 0)
    eye = sp.identity(scale.size, dtype=W.dtype)
    # left-to-right product, exactly like ``I - D * W * D`` (graph.py:94)
    return sp.csr_matrix(eye - (Dm * W) * Dm)


def rescale_L(L, lmax=2):
    """L / (lmax/2) - I on a private copy (graph.py:146-152).

    The reference mutates its argument in place; its caller passes a fresh
    ``csr_matrix(L)`` copy (models_gcn.py:590-591), which is what we return.
    """
    L = sp.csr_matrix(L, copy=True)
    M = L.shape[0]
    # in-place scalar divide of a CSR matrix multiplies .data by the
    # reciprocal and keeps the dtype (fp32 stays fp32)
    L.data *= 1.0 / (lmax / 2)
    L = L - sp.identity(M, format='csr', dtype=L.dtype)
    return sp.csr_matrix(L)


def chebyshev(L, X, K):
    """Stack [T_0 X, ..., T_{K-1} X] of shape (K, M, C)  (graph.py:155-172).

    T_0 X = X, T_1 X = L X, T_k X = 2 L T_{k-1} X - T_{k-2} X, all in the
    dtype of L (the reference asserts L.dtype == X.dtype, graph.py:159).
    """
    M, C = X.shape
    if L.dtype != X.dtype:
        raise AssertionError('L and X dtypes differ')
    out = np.empty((K, M, C), L.dtype)
    out[0] = X
    if K > 1:
        out[1] = L.dot(X)
    for k in range(2, K):
        out[k] = 2 * L.dot(out[k - 1]) - out[k - 2]
    return out
