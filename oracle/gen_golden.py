#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Test infrastructure (see oracle/__init__.py).  Runs only where
``/root/reference`` exists; the fixtures it writes are plain data (inputs and
the reference's outputs) and are what travels to the GPU box.

    python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]

Three sources:
1. ``lib_new/graph.py`` and ``lib_new/coarsening.py`` imported directly.
   ``coarsening`` is run with ``np.argsort`` forced stable (the shim wraps the
   module's ``np`` name; the reference file is untouched) -- see the
   determinism contract in oracle/coarsening_ref.py.  For every case we also
   record whether the un-shimmed run on this NumPy build agrees.
2. ``lib_new/models_gcn.py`` layer methods (chebyshev5, b1relu, b2relu,
   mpool1, apool1, fc, _inference) executed verbatim under ``_TFStub``, a
   NumPy stand-in for the TensorFlow symbols those methods touch.
   TensorFlow is not installed here; the stand-in encodes TF-1 documented
   semantics.
3. The reference's only known-answer vector (coarsening.py:217-218).
"""
import argparse
import contextlib
import hashlib
import io
import os
import sys
import types
import warnings

import numpy as np
import scipy.sparse as sp

warnings.filterwarnings('ignore')


# ----------------------------------------------------------------------------
# NumPy stand-in for the TF symbols used by models_gcn.py:587-682
# ----------------------------------------------------------------------------

class _T(np.ndarray):
    """ndarray that answers ``get_shape()`` like a static-shape tf.Tensor."""

    def get_shape(self):
        return tuple(self.shape)


def _t(a):
    return np.asarray(a).view(_T)


class _Sparse:
    def __init__(self, indices, values, dense_shape):
        idx = np.asarray(indices)
        self.mat = sp.csr_matrix((np.asarray(values), (idx[:, 0], idx[:, 1])),
                                 shape=tuple(dense_shape))
        self.mat.sort_indices()      # tf.sparse_reorder: row-major order


def _pool(x, ksize, strides, padding, reducer):
    assert padding == 'SAME' and list(ksize) == list(strides)
    p = ksize[1]
    N, M, F, one = x.shape
    assert M % p == 0, 'stub handles M % p == 0 only'
    return _t(reducer(np.asarray(x).reshape(N, M // p, p, F, one), axis=2))


def _make_tf_stub(scope_log):
    tf = types.ModuleType('tensorflow')
    tf.float32 = np.float32
    tf.transpose = lambda x, perm=None: _t(np.transpose(x, perm))
    tf.reshape = lambda x, shape: _t(np.reshape(np.ascontiguousarray(x), shape))
    tf.expand_dims = lambda x, axis: _t(np.expand_dims(x, axis))
    tf.concat = lambda xs, axis: _t(np.concatenate(xs, axis=axis))
    tf.squeeze = lambda x, axes: _t(np.squeeze(x, axis=tuple(axes)))
    tf.SparseTensor = _Sparse
    tf.sparse_reorder = lambda s: s
    tf.sparse_tensor_dense_matmul = lambda s, x: _t(s.mat.dot(np.asarray(x)))
    tf.matmul = lambda a, b: _t(np.matmul(np.asarray(a), np.asarray(b)))
    tf.reduce_mean = lambda x, axis: _t(np.mean(np.asarray(x), axis=axis, dtype=x.dtype))

    @contextlib.contextmanager
    def scope(name):
        scope_log.append(name)
        yield
        scope_log.pop()
    tf.variable_scope = scope
    tf.name_scope = scope
    nn = types.SimpleNamespace()
    nn.relu = lambda x: _t(np.maximum(x, 0))
    nn.max_pool = lambda x, ksize, strides, padding: _pool(x, ksize, strides, padding, np.max)
    nn.avg_pool = lambda x, ksize, strides, padding: _pool(x, ksize, strides, padding, np.mean)
    # second argument is keep_prob; the fixtures use keep_prob == 1
    nn.dropout = lambda x, keep: x if keep == 1 else (_ for _ in ()).throw(NotImplementedError())
    tf.nn = nn
    tf.train = types.SimpleNamespace(Saver=object)
    return tf


class _StableNumpy:
    """Proxy for the ``np`` name inside lib_new.coarsening: stable argsort."""

    def __init__(self):
        self.calls = 0

    def __getattr__(self, name):
        return getattr(np, name)

    def argsort(self, a, *args, **kw):
        kw['kind'] = 'stable'
        return np.argsort(a, *args, **kw)


def _csr_fields(prefix, A):
    A = sp.csr_matrix(A)
    A.sort_indices()
    return {prefix + '_indptr': A.indptr.astype(np.int64),
            prefix + '_indices': A.indices.astype(np.int64),
            prefix + '_data': A.data, prefix + '_shape': np.array(A.shape, np.int64)}


def _quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    ap.add_argument('--big', type=int, default=1, help='also run the N=10000 bench graph')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)

    scope_log = []
    sys.modules['tensorflow'] = _make_tf_stub(scope_log)
    sys.path.insert(0, args.ref)
    import lib_new.graph as rgraph
    import lib_new.coarsening as rcoarse
    import lib_new.models_gcn as rmodels

    def save(name, **arrays):
        path = os.path.join(args.out, name + '.npz')
        np.savez_compressed(path, **arrays)
        print('wrote %-28s %7.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))

    # ------------------------------------------------------------------ KAT
    kat_parents = [np.array([4, 1, 1, 2, 2, 3, 0, 0, 3]), np.array([2, 1, 0, 1, 0])]
    kat = rcoarse.compute_perm(kat_parents)
    assert kat == [[3, 4, 0, 9, 1, 2, 5, 8, 6, 7, 10, 11], [2, 4, 1, 3, 0, 5], [0, 1, 2]]
    save('kat_compute_perm', parents0=kat_parents[0], parents1=kat_parents[1],
         perm0=np.array(kat[0]), perm1=np.array(kat[1]), perm2=np.array(kat[2]))

    # ------------------------------------------------- graph construction
    def knn_graph(N, k, seed, dtype):
        z = np.random.RandomState(seed).rand(N, 3).astype(np.float32)
        d, idx = rgraph.distance_sklearn_metrics(z, k=k, metric='euclidean')
        A = rgraph.adjacency(d, idx).astype(dtype)
        return z, d, idx, A

    for name, N, k, dtype, noise in [('graph_n64', 64, 4, np.float32, 0.05),
                                     ('graph_n212', 212, 8, np.float32, 0.01),
                                     ('graph_n100_f64', 100, 6, np.float64, 0.02)]:
        z, d, idx, A = knn_graph(N, k, 0, dtype)
        np.random.seed(0)
        An = rgraph.replace_random_edges(A, noise)
        Ln = rgraph.laplacian(An, normalized=True)
        Lu = rgraph.laplacian(An, normalized=False)
        Lr = rgraph.rescale_L(sp.csr_matrix(Ln), lmax=2)
        X = np.random.RandomState(1).randn(N, 6).astype(dtype)
        T = rgraph.chebyshev(Lr, X, 7)
        fields = dict(z=z, k=np.int64(k), dist=d, idx=idx, noise=np.float64(noise), X=X, T=T)
        for p, M in [('A', A), ('An', An), ('Ln', Ln), ('Lu', Lu), ('Lr', Lr)]:
            fields.update(_csr_fields(p, M))
        save(name, **fields)

    # ----------------------------------------------------------- coarsening
    def run_coarsen(A, levels, stable):
        saved = rcoarse.np
        rcoarse.np = _StableNumpy() if stable else np
        try:
            graphs, parents = _quiet(rcoarse.metis, A, levels)
            np.random.seed(4321)            # metis reseeds; make later draws defined
            cgraphs, perm = _quiet(rcoarse.coarsen, A, levels, False)
            perms = rcoarse.compute_perm(parents)
        finally:
            rcoarse.np = saved
        return graphs, parents, cgraphs, perm, perms

    coarsen_cases = [('coarsen_n64', 64, 4, np.float32, 3),
                     ('coarsen_n212', 212, 8, np.float32, 2),
                     ('coarsen_n100_f64', 100, 6, np.float64, 2),
                     ('coarsen_n512', 512, 8, np.float32, 4)]
    for name, N, k, dtype, levels in coarsen_cases:
        _, _, _, A = knn_graph(N, k, 0, dtype)
        np.random.seed(0)
        A = rgraph.replace_random_edges(A, 0.01)
        graphs, parents, cgraphs, perm, perms = run_coarsen(A, levels, stable=True)
        _, parents_u, _, perm_u, _ = run_coarsen(A, levels, stable=False)
        same = all(np.array_equal(a, b) for a, b in zip(parents, parents_u))
        fields = dict(levels=np.int64(levels), perm=np.array(perm, np.int64),
                      unstable_run_agrees=np.bool_(same))
        fields.update(_csr_fields('A', A))
        for i, par in enumerate(parents):
            fields['parents%d' % i] = par
        for i, pl in enumerate(perms):
            fields['perms%d' % i] = np.array(pl, np.int64)
        for i, G in enumerate(graphs):
            fields.update(_csr_fields('metis%d' % i, G))
        for i, G in enumerate(cgraphs):
            fields.update(_csr_fields('graph%d' % i, G))
        # one-level pass on explicit inputs (first level)
        r, c, v = sp.find(A)
        o = np.argsort(r, kind='stable')
        np.random.seed(1234)
        rid = np.random.permutation(range(A.shape[0]))
        w = np.array(A.sum(axis=0) - A.diagonal()).squeeze()
        fields.update(one_rr=r[o], one_cc=c[o], one_vv=v[o], one_rid=rid, one_w=w,
                      one_cid=rcoarse.metis_one_level(r[o], c[o], v[o], rid, w))
        # perm_data / perm_data_3d
        x2 = np.random.RandomState(2).randn(3, A.shape[0]).astype(np.float32)
        x3 = np.random.RandomState(3).randn(3, A.shape[0], 5).astype(np.float32)
        fields.update(pd_x2=x2, pd_y2=rcoarse.perm_data(x2, perm),
                      pd_x3=x3, pd_y3=rcoarse.perm_data_3d(x3, perm))
        save(name, **fields)
        print('   levels=%d sizes=%s unstable_run_agrees=%s' %
              (levels, [g.shape[0] for g in cgraphs], same))

    # ------------------------------------------------ layers under TF stub
    class Harness(rmodels.cgcnn):
        """cgcnn without __init__: variables come from a preset list."""

        def __init__(self, variables, **attrs):      # noqa: super not called on purpose
            self._vars = list(variables)
            self.__dict__.update(attrs)

        def _weight_variable(self, shape, regularization=True):
            v = self._vars.pop(0)
            assert list(v.shape) == [int(s) for s in shape], (v.shape, shape)
            return _t(v)

        _bias_variable = _weight_variable

    rs = np.random.RandomState(7)
    _, _, _, A = knn_graph(212, 8, 0, np.float32)
    np.random.seed(0)
    A = rgraph.replace_random_edges(A, 0.01)
    graphs, perm = _quiet(_with_stable, rcoarse, lambda: rcoarse.coarsen(A, 3, False))
    Ls = [rgraph.laplacian(G, normalized=True) for G in graphs]
    M0 = Ls[0].shape[0]

    layer_fields = {}
    for i, G in enumerate(Ls):
        layer_fields.update(_csr_fields('L%d' % i, G))
    # single-layer cases: (tag, level, N, Fin, Fout, K)
    for tag, lvl, N, Fin, Fout, K in [('a', 0, 3, 1, 4, 1), ('b', 0, 2, 3, 5, 2),
                                      ('c', 0, 4, 5, 8, 5), ('d', 1, 2, 4, 3, 9),
                                      ('e', 0, 1, 15, 32, 5)]:
        M = Ls[lvl].shape[0]
        x = rs.randn(N, M, Fin).astype(np.float32)
        W = (rs.randn(Fin * K, Fout) * 0.3).astype(np.float32)
        h = Harness([W])
        y = h.chebyshev5(_t(x), Ls[lvl], Fout, K)
        b1 = (rs.randn(1, 1, Fout) * 0.5).astype(np.float32)
        b2 = (rs.randn(1, M, Fout) * 0.5).astype(np.float32)
        y1 = Harness([b1]).b1relu(y)
        y2 = Harness([b2]).b2relu(y)
        layer_fields.update({'cheb_%s_x' % tag: x, 'cheb_%s_W' % tag: W,
                             'cheb_%s_K' % tag: np.int64(K), 'cheb_%s_lvl' % tag: np.int64(lvl),
                             'cheb_%s_y' % tag: np.asarray(y), 'cheb_%s_b1' % tag: b1,
                             'cheb_%s_b2' % tag: b2, 'cheb_%s_y1' % tag: np.asarray(y1),
                             'cheb_%s_y2' % tag: np.asarray(y2)})
        for p in (1, 2, 4):
            layer_fields['cheb_%s_mp%d' % (tag, p)] = np.asarray(Harness([]).mpool1(y2, p))
            layer_fields['cheb_%s_ap%d' % (tag, p)] = np.asarray(Harness([]).apool1(y2, p))
    save('layers_n212', **layer_fields)

    # whole _inference: pooling net and the training.py-style net (model.py:271-280)
    def run_inference(name, Ls_all, F, K, p, Mfc, channel, brelu, N):
        Lk, j = [], 0
        for pp in p:
            Lk.append(Ls_all[j])
            j += int(np.log2(pp)) if pp > 1 else 0
        variables, names = [], []
        Fin, Mcur = channel, Lk[0].shape[0]
        for i, (Fo, Kk, pp) in enumerate(zip(F, K, p)):
            Mi = Lk[i].shape[0]
            variables.append((rs.randn(Fin * Kk, Fo) * np.sqrt(2.0 / (Fin * Kk))).astype(np.float32))
            names.append('conv%d/weights' % (i + 1))
            bshape = (1, 1, Fo) if brelu == 'b1relu' else (1, Mi, Fo)
            variables.append((0.2 + 0.1 * rs.randn(*bshape)).astype(np.float32))
            names.append('conv%d/bias' % (i + 1))
            Fin, Mcur = Fo, Mi // pp
        Min = Mcur
        for i, Mo in enumerate(Mfc):
            scope = 'logits' if i == len(Mfc) - 1 else 'fc%d' % (i + 1)
            variables.append((rs.randn(Min, Mo) * np.sqrt(2.0 / Min)).astype(np.float32))
            names.append(scope + '/weights')
            variables.append((0.2 + 0.1 * rs.randn(Mo)).astype(np.float32))
            names.append(scope + '/bias')
            Min = Mo
        x = rs.randn(N, Lk[0].shape[0], channel).astype(np.float32)
        h = Harness([v.copy() for v in variables], L=Lk, F=F, K=K, p=p, M=Mfc)
        h.filter, h.brelu, h.pool = h.chebyshev5, getattr(h, brelu), h.mpool1
        logits = h._inference(_t(x), 1)
        assert not h._vars
        fields = dict(x=x, logits=np.asarray(logits), F=np.array(F), K=np.array(K),
                      p=np.array(p), M=np.array(Mfc), channel=np.int64(channel),
                      brelu=np.array(brelu), nlevels=np.int64(len(Ls_all)))
        for i, G in enumerate(Ls_all):
            fields.update(_csr_fields('L%d' % i, G))
        for n, v in zip(names, variables):
            fields['param:' + n] = v
        save(name, **fields)

    run_inference('inference_pool_n212', Ls, F=[4, 6, 8], K=[3, 2, 4], p=[2, 4, 1],
                  Mfc=[16, 5], channel=3, brelu='b1relu', N=3)
    run_inference('inference_flat_n212', Ls[:1], F=[8, 8, 8], K=[5, 5, 5], p=[1, 1, 1],
                  Mfc=[32, 16, 22], channel=15, brelu='b2relu', N=2)
    # config 1 of BASELINE.json: K=1 single layer, N=512 graph, block_dura=1, batch 4
    _, _, _, A512 = knn_graph(512, 8, 0, np.float32)
    np.random.seed(0)
    A512 = rgraph.replace_random_edges(A512, 0.01)
    g512, _ = _quiet(_with_stable, rcoarse, lambda: rcoarse.coarsen(A512, 1, False))
    L512 = [rgraph.laplacian(G, normalized=True) for G in g512]
    run_inference('inference_config1_n512', L512[:1], F=[32], K=[1], p=[1],
                  Mfc=[512, 256, 22], channel=1, brelu='b2relu', N=4)

    # the pooling ChebNet of the legacy monolith (HCP_task_fmri_gcn_test8.py:1633-1636, 2071):
    # six coarsening levels, p = [1,4,1,4,1,4], K = [20,10,10,10,5,5], F = [32,32,64,64,128,128], b2relu
    g512_6, _ = _quiet(_with_stable, rcoarse, lambda: rcoarse.coarsen(A512, 6, False))
    L512_6 = [rgraph.laplacian(G, normalized=True) for G in g512_6]
    run_inference('inference_pool6_n512', L512_6, F=[32, 32, 64, 64, 128, 128], K=[20, 10, 10, 10, 5, 5],
                  p=[1, 4, 1, 4, 1, 4], Mfc=[64, 22], channel=3, brelu='b2relu', N=2)

    # -------------------------------------------- the N=10000 bench graph
    if args.big:
        z, d, idx, A = knn_graph(10000, 8, 0, np.float32)
        np.random.seed(0)
        A = rgraph.replace_random_edges(A, 0.01)
        out = {}
        for levels in (1, 6):
            graphs, perm = _quiet(_with_stable, rcoarse, lambda: rcoarse.coarsen(A, levels, False))
            Lr = rgraph.rescale_L(sp.csr_matrix(rgraph.laplacian(graphs[0], normalized=True)), 2)
            Lr.sort_indices()
            out['l%d_sizes' % levels] = np.array([g.shape[0] for g in graphs], np.int64)
            out['l%d_nnz' % levels] = np.array([g.nnz for g in graphs], np.int64)
            out['l%d_perm_sha256' % levels] = np.array(_sha(np.array(perm, np.int64)))
            out['l%d_Lr_nnz' % levels] = np.int64(Lr.nnz)
            out['l%d_Lr_indices_sha256' % levels] = np.array(_sha(Lr.indices.astype(np.int64)))
            out['l%d_Lr_data_sha256' % levels] = np.array(_sha(Lr.data.astype(np.float32)))
            if levels == 1:
                out['l1_perm'] = np.array(perm, np.int32)
        out['A_nnz'] = np.int64(A.nnz)
        out['A_indices_sha256'] = np.array(_sha(sp.csr_matrix(A).indices.astype(np.int64)))
        out['A_data_sha256'] = np.array(_sha(sp.csr_matrix(A).data.astype(np.float32)))
        save('bench_graph_n10000', **out)
        print('   ', {k: (v.tolist() if v.ndim else v.item()) for k, v in out.items()
                      if k.endswith('sizes') or k.endswith('nnz')})

    near_tie_fixture(rgraph, rcoarse, save)


def near_tie_fixture(rgraph, rcoarse, save):
    """A graph whose matching scores (coarsening.py:153) hold exact ties and near-ties: small-integer edge weights
    (1..7) on a kNN pattern, so the scores v*(1/d_i + 1/d_j) are ratios of small integers and the precision the
    expression is evaluated in decides some matches of the strict ``>``.  (Unit weights alone do not: for a fixed
    vertex the score is monotone in 1/d_j in either precision -- six unit-weight graphs tried, no difference.)
    Two runs of the REFERENCE's ``metis`` (stable argsort shim as everywhere): as this NumPy (>= 2) evaluates it --
    float32 throughout -- and with its ``metis_one_level`` handed the same float32 numbers as float64 arrays, which
    is exactly NumPy 1.x's value-based promotion of ``1.0 / np.float32`` (the reference file is untouched; ``metis``
    finds the wrapper through its module's global name)."""
    N, k, levels, seed, hi = 300, 10, 3, 5, 7
    z = np.random.RandomState(seed).rand(N, 3).astype(np.float32)
    d, idx = rgraph.distance_sklearn_metrics(z, k=k, metric='euclidean')
    U = sp.triu(rgraph.adjacency(d, idx), 1).tocoo()
    w = np.random.RandomState(seed + 100).randint(1, hi + 1, U.nnz).astype(np.float32)
    U = sp.csr_matrix((w, (U.row, U.col)), shape=U.shape)
    A = (U + U.T).tocsr().astype(np.float32)
    out = dict(levels=np.int64(levels))
    out.update(_csr_fields('A', A))
    orig = rcoarse.metis_one_level
    res = {}
    for tag in ('f32', 'f32p'):
        if tag == 'f32p':
            rcoarse.metis_one_level = lambda rr, cc, vv, rid, w: orig(rr, cc, np.asarray(vv, np.float64), rid,
                                                                     np.asarray(w, np.float64))
        try:
            graphs, parents = _quiet(_with_stable, rcoarse, lambda: rcoarse.metis(A, levels))
            perms = rcoarse.compute_perm(parents)
        finally:
            rcoarse.metis_one_level = orig
        res[tag] = parents
        for i, par in enumerate(parents):
            out['%s_parents%d' % (tag, i)] = np.asarray(par)
        out['%s_perm' % tag] = np.array(perms[0], np.int64)
        for i, G in enumerate(graphs):
            out.update(_csr_fields('%s_metis%d' % (tag, i), G))
    ndiff = int((res['f32'][0] != res['f32p'][0]).sum())
    assert ndiff > 0, 'choose another graph: the two evaluations agree at the first level'
    out['first_level_parents_differing'] = np.int64(ndiff)
    save('coarsen_ties_n300', **out)
    print('    integer weights 1..%d: float32 and promoted-float64 scores give different first-level parents for %d of %d '
          'vertices' % (hi, ndiff, N))


def _with_stable(rcoarse, fn):
    saved = rcoarse.np
    rcoarse.np = _StableNumpy()
    try:
        return fn()
    finally:
        rcoarse.np = saved


if __name__ == '__main__':
    main()
