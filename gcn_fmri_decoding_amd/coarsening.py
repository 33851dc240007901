"""Multilevel graph coarsening and the binary-tree vertex ordering that turns graph
pooling into a strided max: drop-in for ``lib_new/coarsening.py`` (citations are to that
file).  The sequential inner loops (greedy matching, child lookup) run natively in
libchebgcn.so (csrc/coarsen_host.cpp); matrix assembly stays on SciPy so that weights
are summed exactly like the reference does.

Index maps are bit-exact with the reference *run with a stable edge sort* -- the
reference's ``np.argsort`` calls (:79, :114) use NumPy's unstable default, which makes
its own output depend on the NumPy build; see DESIGN.md "Index maps".
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib


# How the matching score ``vv*(1.0/weights[tid] + 1.0/weights[nid])`` (:153) is evaluated on a float32 graph.
# False (default): in float32 -- what NumPy >= 2 makes of the reference's expression (NEP 50), what this image
# runs and what the golden fixtures hold.  True: float32 values promoted to float64 -- what NumPy 1.x, the
# generation the reference was written for, did (``python float / np.float32`` was a float64); near-ties of the
# strict ``>`` can then fall differently (tests/golden/coarsen_unit_n300.npz records both outcomes).  float64
# graphs are unaffected.  ``metis`` / ``coarsen`` take ``promote=`` per call; this is their default.
NUMPY1_SCORE_PROMOTION = False


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def metis_one_level(rr, cc, vv, rid, weights, promote=None):
    """One greedy heavy-edge matching pass over edges sorted by row (:120-166).

    Returns ``cluster_id`` (int32, one entry per vertex).  Works in the precision of
    ``vv`` (float32 or float64), like NumPy evaluates the reference's expression;
    ``promote`` (default ``NUMPY1_SCORE_PROMOTION``): float32 inputs, score in float64.
    """
    if promote is None:
        promote = NUMPY1_SCORE_PROMOTION
    rr = np.ascontiguousarray(rr, np.int64)
    cc = np.ascontiguousarray(cc, np.int64)
    rid = np.ascontiguousarray(rid, np.int64)
    vv = np.ascontiguousarray(vv)
    if vv.dtype == np.float32:
        fn = _lib.lib().chebgcn_metis_one_level_f32p if promote else _lib.lib().chebgcn_metis_one_level_f32
        ft = np.float32
    else:
        fn, ft = _lib.lib().chebgcn_metis_one_level_f64, np.float64
        vv = vv.astype(np.float64)
    weights = np.ascontiguousarray(weights, ft)
    N = int(rr[-1]) + 1
    if len(rid) < N or len(weights) < N:
        raise ValueError('rid / weights shorter than the vertex count')
    out = np.zeros(N, np.int32)
    _lib.check(fn(len(rr), _ptr(rr), _ptr(cc), _ptr(vv), _ptr(rid), _ptr(weights), N, _ptr(out)),
               'metis_one_level')
    return out


def metis(W, levels, rid=None, promote=None):
    """``levels`` rounds of Graclus-weighted matching and contraction (:34-116).

    Returns (graphs, parents): ``levels + 1`` weight matrices (finest first) and, per
    round, the parent (cluster id) of every vertex.  Like the reference this reseeds the
    global NumPy RNG with 1234 for the first visiting order.
    """
    N = W.shape[0]
    if rid is None:
        np.random.seed(1234)
        rid = np.random.permutation(range(N))
    degree = W.sum(axis=0) - W.diagonal()
    graphs, parents = [W], []
    for _ in range(levels):
        weights = np.array(degree).squeeze()
        r, c, v = sp.find(W)
        order = np.argsort(r, kind='stable')
        rr, cc, vv = r[order], c[order], v[order]
        cid = metis_one_level(rr, cc, vv, rid, weights, promote)
        parents.append(cid)
        n_new = int(cid.max()) + 1
        W = sp.csr_matrix((vv, (cid[rr], cid[cc])), shape=(n_new, n_new))
        W.eliminate_zeros()
        graphs.append(W)
        degree = W.sum(axis=0)
        rid = np.argsort(np.array(W.sum(axis=0)).squeeze(), kind='stable')
    return graphs, parents


def compute_perm(parents):
    """Vertex orderings, finest level first, such that consecutive pairs share a parent
    (:168-215).  Singletons get a fake sibling, childless (fake) parents two fake
    children; fake ids start at the real vertex count of their level."""
    if len(parents) == 0:
        return []
    fn = _lib.lib().chebgcn_compute_perm_level
    order = np.arange(int(np.max(parents[-1])) + 1, dtype=np.int64)
    orders = [order]
    for parent in parents[::-1]:
        parent = np.ascontiguousarray(parent, np.int32)
        out = np.empty(2 * len(order), np.int64)
        _lib.check(fn(_ptr(parent), len(parent), _ptr(order), len(order), _ptr(out)), 'compute_perm')
        order = out
        orders.append(order)
    for i, layer in enumerate(orders):
        if not np.array_equal(np.sort(layer), np.arange(len(orders[0]) * 2 ** i)):
            raise AssertionError('ordering of level %d is not a permutation' % i)
    return [o.tolist() for o in orders[::-1]]


def perm_data(x, indices):
    """Host version of the vertex reordering for x[S, M] (:220-241); float64 like the
    reference.  The training path uses the GPU gather ``ops.perm_data`` instead."""
    if indices is None:
        return x
    S, M = x.shape
    idx = np.asarray(indices)
    if len(idx) < M:
        raise AssertionError('permutation shorter than data')
    out = np.zeros((S, len(idx)))
    keep = idx < M
    out[:, keep] = x[:, idx[keep]]
    return out


def perm_data_3d(x, indices):
    """Host version for x[S, M, F] (:244-265); see ``perm_data``."""
    if indices is None:
        return x
    S, M, F = x.shape
    idx = np.asarray(indices)
    if len(idx) < M:
        raise AssertionError('permutation shorter than data')
    out = np.zeros((S, len(idx), F))
    keep = idx < M
    out[:, keep, :] = x[:, idx[keep], :]
    return out


def perm_adjacency(A, indices):
    """Append isolated fake vertices and relabel rows/columns (:267-294); COO out."""
    if indices is None:
        return A
    M, Mnew = A.shape[0], len(indices)
    if Mnew < M:
        raise AssertionError('permutation shorter than graph')
    A = A.tocoo()
    new_pos = np.argsort(indices, kind='stable')
    return sp.coo_matrix((A.data, (new_pos[A.row], new_pos[A.col])), shape=(Mnew, Mnew), dtype=A.dtype)


def coarsen(A, levels, self_connections=False, verbose=True, promote=None):
    """Coarsen ``A`` ``levels`` times and order every level as a binary tree (:5-31).

    Returns (graphs, perm): CSR adjacencies, the first ``levels`` padded with fake
    vertices and permuted, and the permutation to apply to the input data (None for
    levels == 0)."""
    graphs, parents = metis(A, levels, promote=promote)
    perms = compute_perm(parents)
    for i, G in enumerate(graphs):
        M = G.shape[0]
        if not self_connections:
            G = G.tocoo()
            G.setdiag(0)
        if i < levels:
            G = perm_adjacency(G, perms[i])
        G = G.tocsr()
        G.eliminate_zeros()
        graphs[i] = G
        if verbose:
            print('Layer {0}: M_{0} = |V| = {1} nodes ({2} added),|E| = {3} edges'.format(
                i, G.shape[0], G.shape[0] - M, G.nnz // 2))
    return graphs, (perms[0] if levels > 0 else None)
