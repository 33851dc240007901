"""Graph construction on the host: drop-in for the parts of ``lib_new/graph.py`` that
feed the Chebyshev hot path (reference citations are to that file).

These run once per model build on NumPy/SciPy, like the reference; the results (CSR
Laplacians) are uploaded to the GPU by ``ops.Graph``.  Out of scope here and absent on
purpose: ``fourier`` / ``plot_spectrum`` (spectral models) and the dense NumPy
recurrence ``chebyshev`` (the GPU kernel replaces it).
"""
import numpy as np
import scipy.sparse as sp


def distance_sklearn_metrics(z, k=4, metric='euclidean'):
    """Exact k nearest neighbours from the full distance matrix (graph.py:9-17).

    Returns (d, idx): the k smallest non-self distances per row, ascending, and the
    matching column indices.
    """
    import sklearn.metrics
    d = sklearn.metrics.pairwise.pairwise_distances(z, metric=metric, n_jobs=-2)
    idx = np.argsort(d)[:, 1:k + 1]
    d.sort()
    return d[:, 1:k + 1], idx


def adjacency(dist, idx):
    """Gaussian-weighted, symmetrised kNN adjacency (graph.py:19-45).

    w_ij = exp(-d_ij^2 / sigma^2) with sigma = mean distance to the k-th neighbour;
    an undirected edge keeps the larger of the two directed weights.
    """
    M, k = dist.shape
    if idx.shape != (M, k):
        raise ValueError('dist and idx shapes differ')
    if dist.min() < 0:
        raise ValueError('negative distance')
    sigma2 = np.mean(dist[:, -1]) ** 2
    w = np.exp(-dist ** 2 / sigma2)
    rows = np.arange(0, M).repeat(k)
    W = sp.coo_matrix((w.reshape(M * k), (rows, idx.reshape(M * k))), shape=(M, M))
    W.setdiag(0)
    flip = W.T > W
    W = W - W.multiply(flip) + W.T.multiply(flip)
    return sp.csr_matrix(W)


def replace_random_edges(A, noise_level):
    """Swap a fraction of the edges for uniformly random unit edges (graph.py:48-76).

    Consumes the global NumPy RNG in the reference's order (permutation, two randint
    draws, one uniform draw) so that a seeded call reproduces its graph.
    """
    M = A.shape[0]
    n = int(noise_level * A.nnz // 2)
    victims = np.random.permutation(A.nnz // 2)[:n]
    new_r = np.random.randint(0, M, n)
    new_c = np.random.randint(0, M, n)
    np.random.uniform(0, 1, n)            # drawn and unused by the reference as well
    upper = sp.triu(A, format='coo')
    if upper.nnz < n:
        raise ValueError('not enough edges to replace')
    A = A.tolil()
    for e, r, c in zip(victims, new_r, new_c):
        i, j = upper.row[e], upper.col[e]
        A[i, j] = 0
        A[j, i] = 0
        A[r, c] = 1
        A[c, r] = 1
    A.setdiag(0)
    A = A.tocsr()
    A.eliminate_zeros()
    return A


def laplacian(W, normalized=True):
    """Combinatorial (D - W) or normalised (I - D^-1/2 W D^-1/2) Laplacian, CSR, in the
    dtype of W (graph.py:79-98).  Degrees are column sums plus ``spacing(0)``."""
    W = sp.csr_matrix(W)
    d = np.asarray(W.sum(axis=0)).ravel()
    if not normalized:
        return sp.csr_matrix(sp.diags(d, 0) - W)
    d = d + np.spacing(np.array(0, W.dtype))
    s = sp.diags((1 / np.sqrt(d)).astype(W.dtype, copy=False), 0)
    eye = sp.identity(d.size, dtype=W.dtype)
    return sp.csr_matrix(eye - (s * W) * s)


def lmax(L, normalized=True):
    """Upper bound of the spectrum (graph.py:101-107)."""
    if normalized:
        return 2
    import scipy.sparse.linalg
    return scipy.sparse.linalg.eigsh(L, k=1, which='LM', return_eigenvectors=False)[0]


def rescale_L(L, lmax=2):
    """Map the spectrum to [-1, 1]: L / (lmax/2) - I (graph.py:146-152).

    Unlike the reference this does not mutate its argument (its only caller passes a
    private copy, models_gcn.py:590-591).  Entries that cancel exactly are dropped.
    """
    L = sp.csr_matrix(L, copy=True)
    L.data *= 1.0 / (lmax / 2)
    return sp.csr_matrix(L - sp.identity(L.shape[0], format='csr', dtype=L.dtype))


def rescaled_laplacian_csr(L):
    """What ``chebyshev5`` hands to its sparse matmul (models_gcn.py:590-596): the
    rescaled Laplacian as float32 CSR with row-major, ascending-column entries
    (``tf.sparse_reorder``).  Returns (indptr int32, indices int32, data float32)."""
    Lr = rescale_L(sp.csr_matrix(L), lmax=2).astype(np.float32)
    Lr.sort_indices()
    return (Lr.indptr.astype(np.int32), Lr.indices.astype(np.int32), Lr.data.astype(np.float32))


def length_order(L):
    """Vertex order in which the library's recurrence kernels are fastest: rows of the rescaled Laplacian sorted by
    DESCENDING length (number of neighbours), ties in the caller's order (stable), so isolated vertices -- the fake
    vertices the coarsening adds -- come last.  ``order[i]`` = the caller's index of internal vertex ``i``.

    With ``Lp = permute(L, order)`` a quad of four consecutive vertices is four rows of (nearly) equal length: a
    16-byte piece of an activation plane is then exactly the rows one thread of the recurrence kernel owns, and planes
    go from HBM to registers and back without a pass through LDS (csrc/recurrence_ord.hip).  The network is invariant
    under a relabelling of the vertices as long as everything per-vertex follows it (input columns, per-vertex biases,
    the rows of the first FC layer; pooling needs the tree order of ``coarsening.compute_perm`` and is not relabelled):
    ``models_gcn.cgcnn`` does that behind the reference's variable layout."""
    indptr, _, _ = rescaled_laplacian_csr(L)
    lengths = np.diff(indptr)
    return np.argsort(-lengths.astype(np.int64), kind='stable').astype(np.int64)


def bank_order(L, sweeps=4, stats=None):
    """``length_order(L)`` refined inside its classes of equal row length so that the LDS reads of the ordered recurrence
    kernels' gather spread over the banks (``chebgcn_bank_order``, csrc/graph.hip: pairwise label swaps inside the gather's
    lane sets, host only, deterministic).  Still sorted by descending row length: everything said of ``length_order`` holds.
    ``stats``: a list that receives [fullest-bank sum before, after, swaps]."""
    import ctypes as C
    from . import _lib
    order = length_order(L)
    indptr, indices, _ = rescaled_laplacian_csr(permute(L, order))
    M = int(L.shape[0])
    perm = np.empty(M, np.int32)
    st = np.zeros(3, np.int64)
    _lib.check(_lib.lib().chebgcn_bank_order(M, indptr.ctypes.data_as(C.c_void_p), indices.ctypes.data_as(C.c_void_p), int(sweeps),
                                             perm.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p)), 'bank_order')
    if stats is not None:
        stats[:] = [int(v) for v in st]
    return order[perm.astype(np.int64)]


def permute(L, order):
    """``P L P^T``: row / column ``i`` of the result is row / column ``order[i]`` of ``L`` (CSR, sorted indices)."""
    L = sp.csr_matrix(L)
    Lp = sp.csr_matrix(L[order][:, order])
    Lp.sort_indices()
    return Lp


def synthetic_graph(n_nodes=10000, k=8, levels=1, noise_level=0.01, seed=0, dtype=np.float32):
    """The seeded synthetic "brain" graph of the benchmark: kNN
    graph on uniform points in the unit cube, 1 % random edges, ``levels`` rounds of
    coarsening, one normalised Laplacian per level.  Returns (laplacians, perm, graphs).
    """
    from . import coarsening
    z = np.random.RandomState(seed).rand(n_nodes, 3).astype(np.float32)
    d, idx = distance_sklearn_metrics(z, k=k, metric='euclidean')
    A = adjacency(d, idx).astype(dtype)
    np.random.seed(seed)
    A = replace_random_edges(A, noise_level)
    graphs, perm = coarsening.coarsen(A, levels=levels, self_connections=False, verbose=False)
    return [laplacian(G, normalized=True) for G in graphs], perm, graphs
