"""Product host code (gcn_fmri_decoding_amd.graph / .coarsening, native loops in
libchebgcn.so) against golden vectors produced by the reference.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import assert_csr_equal, csr_from, load_golden
from gcn_fmri_decoding_amd import coarsening as C
from gcn_fmri_decoding_amd import graph as G


def test_compute_perm_known_answer():
    # lib_new/coarsening.py:217-218
    got = C.compute_perm([np.array([4, 1, 1, 2, 2, 3, 0, 0, 3]), np.array([2, 1, 0, 1, 0])])
    assert got == [[3, 4, 0, 9, 1, 2, 5, 8, 6, 7, 10, 11], [2, 4, 1, 3, 0, 5], [0, 1, 2]]
    assert C.compute_perm([]) == []


@pytest.mark.parametrize('name', ['graph_n64', 'graph_n212', 'graph_n100_f64'])
def test_graph_construction(name):
    z = load_golden(name)
    d, idx = G.distance_sklearn_metrics(z['z'], k=int(z['k']))
    assert np.array_equal(idx, z['idx'])
    np.testing.assert_allclose(d, z['dist'], rtol=1e-6)
    A = G.adjacency(z['dist'], z['idx']).astype(z['A_data'].dtype)
    assert_csr_equal(A, csr_from(z, 'A'))
    np.random.seed(0)
    An = G.replace_random_edges(csr_from(z, 'A'), float(z['noise']))
    assert_csr_equal(An, csr_from(z, 'An'))
    Ln = G.laplacian(An, normalized=True)
    assert Ln.dtype == An.dtype
    assert_csr_equal(Ln, csr_from(z, 'Ln'))
    assert_csr_equal(G.laplacian(An, normalized=False), csr_from(z, 'Lu'))
    assert_csr_equal(G.rescale_L(Ln, lmax=2), csr_from(z, 'Lr'))
    assert_csr_equal(Ln, csr_from(z, 'Ln'))          # argument untouched
    ptr, ind, dat = G.rescaled_laplacian_csr(Ln)
    Lr = csr_from(z, 'Lr')
    assert np.array_equal(ptr, Lr.indptr) and np.array_equal(ind, Lr.indices)
    assert dat.dtype == np.float32 and np.array_equal(dat, Lr.data.astype(np.float32))


@pytest.mark.parametrize('name', ['coarsen_n64', 'coarsen_n212', 'coarsen_n100_f64', 'coarsen_n512'])
def test_coarsening_bit_exact(name):
    z = load_golden(name)
    A = csr_from(z, 'A')
    levels = int(z['levels'])
    cid = C.metis_one_level(z['one_rr'], z['one_cc'], z['one_vv'], z['one_rid'], z['one_w'])
    assert cid.dtype == np.int32 and np.array_equal(cid, z['one_cid'])
    graphs, parents = C.metis(A, levels)
    for i in range(levels):
        assert np.array_equal(parents[i], z['parents%d' % i])
    for i in range(levels + 1):
        assert_csr_equal(graphs[i], csr_from(z, 'metis%d' % i))
    perms = C.compute_perm(parents)
    for i in range(levels + 1):
        assert perms[i] == z['perms%d' % i].tolist()
    cgraphs, perm = C.coarsen(A, levels, verbose=False)
    assert perm == z['perm'].tolist()
    for i in range(levels + 1):
        assert_csr_equal(cgraphs[i], csr_from(z, 'graph%d' % i))
    assert np.array_equal(C.perm_data(z['pd_x2'], perm), z['pd_y2'])
    assert np.array_equal(C.perm_data_3d(z['pd_x3'], perm), z['pd_y3'])


def test_coarsen_edge_cases():
    z = load_golden('coarsen_n64')
    graphs, perm = C.coarsen(csr_from(z, 'A'), 0, verbose=False)
    assert perm is None and len(graphs) == 1
    with pytest.raises(Exception):
        C.metis_one_level(np.array([0, 1]), np.array([1, 5]), np.ones(2, np.float32), np.arange(2), np.ones(2, np.float32))


def test_bench_graph_matches_reference_recipe():
    """N=10000 synthetic benchmark graph: sizes, nnz and a checksum of the
    permutation / rescaled Laplacian as produced by the reference's own functions."""
    import hashlib
    z = load_golden('bench_graph_n10000')
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    Ls, perm, graphs = G.synthetic_graph(10000, k=8, levels=1)
    assert [g.shape[0] for g in graphs] == z['l1_sizes'].tolist() == [10466, 5233]
    assert [g.nnz for g in graphs] == z['l1_nnz'].tolist()
    assert np.array_equal(np.array(perm, np.int32), z['l1_perm'])
    ptr, ind, dat = G.rescaled_laplacian_csr(Ls[0])
    assert len(dat) == int(z['l1_Lr_nnz']) == 93880
    assert sha(ind.astype(np.int64)) == str(z['l1_Lr_indices_sha256'])
    assert sha(dat) == str(z['l1_Lr_data_sha256'])


def test_bench_graph_six_levels_matches_reference_recipe():
    """The same graph coarsened six times (the pooling ChebNet of the legacy monolith,
    HCP_task_fmri_gcn_test8.py:1633-1636, 2071): level sizes, nnz and checksums of the permutation
    and of the rescaled level-0 Laplacian as the reference's own functions produce them."""
    import hashlib
    z = load_golden('bench_graph_n10000')
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    Ls, perm, graphs = G.synthetic_graph(10000, k=8, levels=6)
    assert [g.shape[0] for g in graphs] == z['l6_sizes'].tolist() == [12672, 6336, 3168, 1584, 792, 396, 198]
    assert [g.nnz for g in graphs] == z['l6_nnz'].tolist()
    assert sha(np.array(perm, np.int64)) == str(z['l6_perm_sha256'])
    ptr, ind, dat = G.rescaled_laplacian_csr(Ls[0])
    assert len(dat) == int(z['l6_Lr_nnz'])
    assert sha(ind.astype(np.int64)) == str(z['l6_Lr_indices_sha256'])
    assert sha(dat) == str(z['l6_Lr_data_sha256'])


@pytest.mark.parametrize('tag, promote', [('f32', False), ('f32p', True)])
def test_coarsening_near_ties_both_numpy_generations(tag, promote):
    """Integer-weight graph full of ties and near-ties of the matching score (coarsening.py:153): the fixture holds the
    REFERENCE's output as NumPy >= 2 evaluates the score (float32, ``f32_*``) and as NumPy 1.x did (float32 values
    promoted to float64, ``f32p_*``).  23 first-level parents differ between the two; each mode reproduces its own."""
    z = load_golden('coarsen_ties_n300')
    A = csr_from(z, 'A')
    levels = int(z['levels'])
    assert int(z['first_level_parents_differing']) == int((z['f32_parents0'] != z['f32p_parents0']).sum()) > 0
    assert C.NUMPY1_SCORE_PROMOTION is False          # the default: what this image's NumPy makes of the reference
    graphs, parents = C.metis(A, levels, promote=promote)
    if not promote:
        assert all(np.array_equal(a, b) for a, b in zip(C.metis(A, levels)[1], parents))
    for i in range(levels):
        assert np.array_equal(parents[i], z['%s_parents%d' % (tag, i)])
    for i in range(levels + 1):
        assert_csr_equal(graphs[i], csr_from(z, '%s_metis%d' % (tag, i)))
    assert C.compute_perm(parents)[0] == z['%s_perm' % tag].tolist()
