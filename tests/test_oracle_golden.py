"""Oracle (oracle/*.py) vs golden vectors produced by the reference itself
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import assert_csr_equal, csr_from, load_golden
from oracle import coarsening_ref as C
from oracle import graph_ref as G
from oracle import layers_ref as R


def test_compute_perm_known_answer():
    # the reference's only KAT: lib_new/coarsening.py:217-218
    z = load_golden('kat_compute_perm')
    got = C.compute_perm([z['parents0'], z['parents1']])
    assert got == [[3, 4, 0, 9, 1, 2, 5, 8, 6, 7, 10, 11], [2, 4, 1, 3, 0, 5], [0, 1, 2]]
    assert got == [z['perm0'].tolist(), z['perm1'].tolist(), z['perm2'].tolist()]
    assert C.compute_perm([]) == []


@pytest.mark.parametrize('name', ['graph_n64', 'graph_n212', 'graph_n100_f64'])
def test_laplacian_rescale_chebyshev(name):
    z = load_golden(name)
    An = csr_from(z, 'An')
    Ln = G.laplacian(An, normalized=True)
    assert Ln.dtype == An.dtype
    assert_csr_equal(Ln, csr_from(z, 'Ln'))
    assert_csr_equal(G.laplacian(An, normalized=False), csr_from(z, 'Lu'))
    Lr = G.rescale_L(Ln, lmax=2)
    assert Lr.dtype == An.dtype
    assert_csr_equal(Lr, csr_from(z, 'Lr'))
    # the argument is not mutated (the reference's caller passes a copy)
    assert_csr_equal(Ln, csr_from(z, 'Ln'))
    T = G.chebyshev(Lr, z['X'], 7)
    assert T.dtype == z['T'].dtype
    assert np.array_equal(T, z['T'])


@pytest.mark.parametrize('name', ['coarsen_n64', 'coarsen_n212', 'coarsen_n100_f64', 'coarsen_n512'])
def test_coarsening_bit_exact(name):
    z = load_golden(name)
    A = csr_from(z, 'A')
    levels = int(z['levels'])
    cid = C.metis_one_level(z['one_rr'], z['one_cc'], z['one_vv'], z['one_rid'], z['one_w'])
    assert np.array_equal(cid, z['one_cid'])
    graphs, parents = C.metis(A, levels)
    for i in range(levels):
        assert np.array_equal(parents[i], z['parents%d' % i])
    for i in range(levels + 1):
        assert_csr_equal(graphs[i], csr_from(z, 'metis%d' % i))
    perms = C.compute_perm(parents)
    for i in range(levels + 1):
        assert perms[i] == z['perms%d' % i].tolist()
    cgraphs, perm = C.coarsen(A, levels)
    assert perm == z['perm'].tolist()
    for i in range(levels + 1):
        assert_csr_equal(cgraphs[i], csr_from(z, 'graph%d' % i))
    y2 = C.perm_data(z['pd_x2'], perm)
    y3 = C.perm_data_3d(z['pd_x3'], perm)
    assert y2.dtype == np.float64 and np.array_equal(y2, z['pd_y2'])
    assert y3.dtype == np.float64 and np.array_equal(y3, z['pd_y3'])


def test_coarsen_levels0():
    z = load_golden('coarsen_n64')
    graphs, perm = C.coarsen(csr_from(z, 'A'), 0)
    assert perm is None and len(graphs) == 1


def test_layers_vs_reference_source():
    """chebyshev5 / b1relu / b2relu / mpool1 / apool1 against the reference's
    own methods executed under the TF stand-in (gen_golden.py)."""
    z = load_golden('layers_n212')
    Ls = [csr_from(z, 'L%d' % i) for i in range(4)]
    for tag in 'abcde':
        x, W = z['cheb_%s_x' % tag], z['cheb_%s_W' % tag]
        K, lvl = int(z['cheb_%s_K' % tag]), int(z['cheb_%s_lvl' % tag])
        y = R.chebyshev5_fwd(x, Ls[lvl], W, K)
        ref = z['cheb_%s_y' % tag]
        assert y.dtype == np.float32
        # same operation order -> agreement to matmul blocking noise
        np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
        y1 = R.brelu_fwd(ref, z['cheb_%s_b1' % tag])
        y2 = R.brelu_fwd(ref, z['cheb_%s_b2' % tag])
        assert np.array_equal(y1, z['cheb_%s_y1' % tag])
        assert np.array_equal(y2, z['cheb_%s_y2' % tag])
        for p in (1, 2, 4):
            mp, _ = R.mpool1_fwd(y2, p)
            assert np.array_equal(mp, z['cheb_%s_mp%d' % (tag, p)])
            np.testing.assert_allclose(R.apool1_fwd(y2, p), z['cheb_%s_ap%d' % (tag, p)], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('name', ['inference_pool_n212', 'inference_flat_n212', 'inference_config1_n512', 'inference_pool6_n512'])
def test_inference_vs_reference_source(name):
    z = load_golden(name)
    Ls = [csr_from(z, 'L%d' % i) for i in range(int(z['nlevels']))]
    net = R.Net(Ls, z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist(),
                channel=int(z['channel']), brelu=str(z['brelu']))
    params = {k[len('param:'):]: z[k] for k in z.files if k.startswith('param:')}
    assert {k: v.shape for k, v in params.items()} == net.param_shapes()
    logits, _ = net.forward(params, z['x'])
    ref = z['logits']
    np.testing.assert_allclose(logits, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize('tag, promote', [('f32', False), ('f32p', True)])
def test_coarsening_near_ties_both_numpy_generations(tag, promote):
    """Integer-weight graph full of ties and near-ties of the matching score (coarsening.py:153): the fixture holds the
    REFERENCE's output as NumPy >= 2 evaluates the score (float32, ``f32_*``) and as NumPy 1.x did (float32 values
    promoted to float64, ``f32p_*``).  23 first-level parents differ between the two; each mode reproduces its own."""
    z = load_golden('coarsen_ties_n300')
    A = csr_from(z, 'A')
    levels = int(z['levels'])
    assert int(z['first_level_parents_differing']) == int((z['f32_parents0'] != z['f32p_parents0']).sum()) > 0
    graphs, parents = C.metis(A, levels, promote=promote)
    for i in range(levels):
        assert np.array_equal(parents[i], z['%s_parents%d' % (tag, i)])
    for i in range(levels + 1):
        assert_csr_equal(graphs[i], csr_from(z, '%s_metis%d' % (tag, i)))
    assert C.compute_perm(parents)[0] == z['%s_perm' % tag].tolist()
