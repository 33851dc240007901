"""Oracle (test infrastructure, see oracle/__init__.py): multilevel coarsening.

Restates lib_new/coarsening.py of the reference:
* ``coarsen``          -- :5-31
* ``metis``            -- :34-116   (Graclus-weighted greedy matching driver)
* ``metis_one_level``  -- :120-166  (one greedy pass, incl. its row-length quirk)
* ``compute_perm``     -- :168-215  (binary-tree ordering with fake nodes)
* ``perm_data``        -- :220-241,  ``perm_data_3d`` -- :244-265
* ``perm_adjacency``   -- :267-294

Determinism contract (DESIGN.md "Index maps"): the reference sorts the edge
triplets with NumPy's default *unstable* argsort (coarsening.py:79, :114), so
its output depends on the NumPy build.  The oracle -- and the golden fixtures
generated from the reference by ``oracle/gen_golden.py`` -- use a *stable*
sort, i.e. row-major / ascending-column edge order and ascending-index ties.
"""
import numpy as np
import scipy.sparse as sp


def metis_one_level(rr, cc, vv, rid, weights, promote=False):
    """One greedy heavy-edge matching pass (coarsening.py:120-166).

    ``promote``: evaluate the score in float64 on float32 inputs -- what NumPy 1.x
    (value-based casting: ``1.0 / np.float32`` is a float64) made of :153; the default is
    what NumPy >= 2 does (NEP 50: the expression stays float32).

    ``rr`` is assumed sorted.  Row extents are derived exactly like the
    reference's counting loop (:134-139): the boundary test runs *after* the
    increment, so the first run also claims the first entry of the second run
    and the last run is one entry short; runs are numbered by order of
    appearance (``count``), not by row id, which only matters for graphs with
    empty rows.
    """
    if promote:
        vv, weights = np.asarray(vv, np.float64), np.asarray(weights, np.float64)
    nnz = rr.shape[0]
    N = int(rr[nnz - 1]) + 1
    # positions ii where rr[ii] exceeds everything seen before it
    run_max = np.maximum.accumulate(rr)
    starts = np.flatnonzero(rr[1:] > run_max[:-1]) + 1
    rowstart = np.zeros(N, np.int64)
    rowlength = np.zeros(N, np.int64)
    nruns = len(starts) + 1
    rowstart[1:nruns] = starts
    bounds = np.concatenate(([-1], starts, [nnz - 1]))
    # run c receives one increment for every ii in (bounds[c], bounds[c+1]]
    rowlength[:nruns] = np.diff(bounds)

    marked = np.zeros(N, bool)
    cluster_id = np.zeros(N, np.int32)
    nclusters = 0
    for tid in rid[:N]:
        if marked[tid]:
            continue
        marked[tid] = True
        best, wmax = -1, 0.0
        lo = rowstart[tid]
        for e in range(lo, lo + rowlength[tid]):
            nid = cc[e]
            if marked[nid]:
                tval = 0.0
            else:
                # evaluated in the dtype of vv/weights, as NumPy does for
                # ``vv[e] * (1.0/weights[tid] + 1.0/weights[nid])`` (:153)
                tval = vv[e] * (1.0 / weights[tid] + 1.0 / weights[nid])
            if tval > wmax:          # strict: first maximum wins (:154)
                wmax, best = tval, nid
        cluster_id[tid] = nclusters
        if best > -1:
            cluster_id[best] = nclusters
            marked[best] = True
        nclusters += 1
    return cluster_id


def metis(W, levels, rid=None, promote=False):
    """``levels`` rounds of matching + graph contraction (coarsening.py:34-116)."""
    N = W.shape[0]
    if rid is None:
        # reseeds the *global* NumPy RNG, like the reference (:55-57)
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
        Nnew = int(cid.max()) + 1
        # duplicate (row, col) pairs are summed by the CSR constructor (:99)
        W = sp.csr_matrix((vv, (cid[rr], cid[cc])), shape=(Nnew, Nnew))
        W.eliminate_zeros()
        graphs.append(W)
        degree = W.sum(axis=0)                     # self loops kept (:105)
        ss = np.array(W.sum(axis=0)).squeeze()
        rid = np.argsort(ss, kind='stable')        # ascending weighted degree
    return graphs, parents


def compute_perm(parents):
    """Binary-tree node ordering per level, finest first (coarsening.py:168-215).

    Walks from the coarsest level down.  For each node of the current ordering
    list the children are the finer-level vertices whose parent it is (in
    ascending index); one child -> append a fresh fake sibling, no child (the
    node itself was fake) -> two fresh fake children.  Fake ids are handed out
    consecutively starting at the real vertex count of that level.
    """
    orders = []
    if len(parents) > 0:
        orders.append(list(range(int(max(parents[-1])) + 1)))
    for parent in parents[::-1]:
        parent = np.asarray(parent)
        by_parent = np.argsort(parent, kind='stable')
        first = np.searchsorted(parent[by_parent], np.arange(parent.max() + 2))
        next_fake = len(parent)
        layer = []
        for node in orders[-1]:
            if node + 1 < len(first):
                kids = [int(k) for k in by_parent[first[node]:first[node + 1]]]
            else:
                kids = []
            if len(kids) > 2:
                raise AssertionError('more than two children')
            while len(kids) < 2:
                kids.append(next_fake)
                next_fake += 1
            layer.extend(kids)
        orders.append(layer)
    for i, layer in enumerate(orders):
        if sorted(layer) != list(range(len(orders[0]) * 2 ** i)):
            raise AssertionError('ordering is not a permutation')
    return orders[::-1]


def perm_data(x, indices):
    """Reorder/pad the vertex axis of x[S, M] (coarsening.py:220-241); float64 out."""
    if indices is None:
        return x
    S, M = x.shape
    idx = np.asarray(indices)
    if len(idx) < M:
        raise AssertionError('permutation shorter than data')
    out = np.zeros((S, len(idx)))
    real = idx < M
    out[:, real] = x[:, idx[real]]
    return out


def perm_data_3d(x, indices):
    """Reorder/pad the vertex axis of x[S, M, F] (coarsening.py:244-265); float64 out."""
    if indices is None:
        return x
    S, M, F = x.shape
    idx = np.asarray(indices)
    if len(idx) < M:
        raise AssertionError('permutation shorter than data')
    out = np.zeros((S, len(idx), F))
    real = idx < M
    out[:, real, :] = x[:, idx[real], :]
    return out


def perm_adjacency(A, indices):
    """Pad with isolated vertices and relabel (coarsening.py:267-294); COO out."""
    if indices is None:
        return A
    M = A.shape[0]
    Mnew = len(indices)
    if Mnew < M:
        raise AssertionError('permutation shorter than graph')
    A = A.tocoo()
    rank = np.argsort(indices, kind='stable')     # old id -> new position
    return sp.coo_matrix((A.data, (rank[A.row], rank[A.col])),
                         shape=(Mnew, Mnew), dtype=A.dtype)


def coarsen(A, levels, self_connections=False):
    """Multilevel coarsening + tree ordering (coarsening.py:5-31).

    Returns (graphs, perm): ``levels+1`` CSR adjacency matrices, the first
    ``levels`` of them padded/reordered so that pooling of size 2 merges
    siblings, and the finest-level permutation (None when levels == 0).
    """
    graphs, parents = metis(A, levels)
    perms = compute_perm(parents)
    for i, G in enumerate(graphs):
        if not self_connections:
            G = G.tocoo()
            G.setdiag(0)
        if i < levels:
            G = perm_adjacency(G, perms[i])
        G = G.tocsr()
        G.eliminate_zeros()
        graphs[i] = G
    return graphs, (perms[0] if levels > 0 else None)
