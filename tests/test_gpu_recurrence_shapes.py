"""Every kernel shape of the Chebyshev recurrence BY NAME, over the graph sizes that reach it.  Needs an MI355X: ``-m gpu``.

Part 1: the ordered kernels (csrc/recurrence_ord_kernel.h).  Part 2 (``test_caller_order_recurrence_shape``): the kernels a
graph in the caller's vertex order runs -- networks with pooling, atlas sizes, pinned plane counts, sizes the ordered kernels
do not serve (csrc/recurrence.hip ``dispatch_onchip``: 256 / 512 / 768 threads, 1..40 rows per thread, two or four planes;
csrc/recurrence4.hip ``dispatch_onchip4``: four planes beyond 2048 rows).

A network without pooling relabels its vertices by descending row length (``graph.length_order``), and the library then runs
``cheb_ord_kernel<ENT,NQ,NG,512,*>`` (four planes per workgroup, 16-byte LDS entries: 2049 ... 10238 active vertices; round 6:
``cheb_ord_kernel<ENT,NQ,NG,256,false>`` for planes of more than 1024 vertices with at most 2048 active ones, forward only) or
``cheb_ord2_kernel<...>`` (two planes, 8-byte entries: from 10753 vertices up to 20476 active ones; in the window between, the
768-thread two-plane kernel of recurrence.hip on the caller's order is faster and is what runs).  NG = quad levels with rows = ceil(active / 4 / 512),
NQ = quad levels in all = ceil(Mp / 4 / 512).  Reference semantics: ``lib_new/models_gcn.py:598-610`` (the recurrence),
``lib_new/graph.py:155-172`` (``graph.chebyshev``, the oracle's twin), TF autodiff of it for the adjoint.

Every case: the kernel template asserted through ``chebgcn_last_dispatch()``; EVERY plane of every order of a launch with
more plane groups than workgroups and a partial last group against the same recurrence in float64 (``torch.sparse`` on the
device, itself tied to the CPU oracle ``oracle/graph_ref`` on three planes at 1e-12), 1e-5 of the plane's own maximum
(adjoint 2e-5); in place (T_0 already in slab 0) == with the copy of x, bit for bit; pads poisoned with NaN never leak;
K = 2 (one step: the first is the last) and K = 3 beside K = 5.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import record_measured

pytestmark = pytest.mark.gpu
REL, GREL = 1e-5, 2e-5

# (points of the synthetic kNN graph, coarsening levels) -> (kernel family, ENT, NQ, NG)
SHAPES = [
    # round 6: planes of 1025 ... 2048 vertices on the 256-thread shapes (forward; the Clenshaw adjoint of such a graph stays on the
    # on-chip kernel of recurrence.hip, which runs in any vertex order and is the faster one there)
    ((1000, 1), ('cheb_ord_kernel', 1040, 2, 1, 256, 'cheb_onchip_kernel<4,4,2,256,true>')),
    ((2000, 1), ('cheb_ord_kernel', 2064, 3, 2, 256, 'cheb_onchip_kernel<4,8,3,256,true>')),
    ((800, 1), ('cheb_onchip_kernel', 4, 4, 1, 256)),       # at most 1024 vertices per plane: the caller's order
    ((2600, 1), ('cheb_ord_kernel', 4112, 2, 2)),
    ((6000, 1), ('cheb_ord_kernel', 6160, 4, 3)),
    ((8000, 1), ('cheb_ord_kernel', 8208, 5, 4)),
    ((9000, 1), ('cheb_ord_kernel', 10240, 5, 5)),
    ((10000, 0), ('cheb_ord_kernel', 10240, 5, 5)),         # the benchmark's points without a coarsening level
    ((10000, 1), ('cheb_ord_kernel', 10240, 6, 5)),         # the benchmark graph
    # six coarsening levels (the pooling network of HCP_task_fmri_gcn_test8.py:1632-1635): 2672 fake vertices behind the 10000 real
    # ones = more than one quad level of isolated vertices: the kernel's NQ stops at NG + 1, cheb_ord_tail_kernel streams the rest
    ((10000, 6), ('cheb_ord_kernel', 10240, 6, 5)),
    ((5000, 6), ('cheb_ord_kernel', 6160, 4, 3)),
    # 2560 quads of rows are two entries too many for 16-byte entries; up to 10752 vertices the two-plane kernel of recurrence.hip
    # on the CALLER's order is the faster one (recurrence_ord.hip ordered_shape): no ordered image, that kernel by name
    ((10239, 0), ('cheb_onchip_kernel', 2, 14, 4)),
    ((10242, 1), ('cheb_onchip_kernel', 2, 14, 4)),         # a 10242-vertex cortical mesh's size (M = 10742)
    ((11000, 1), ('cheb_ord2_kernel', 12304, 6, 6)),
    ((13000, 1), ('cheb_ord2_kernel', 14352, 7, 7)),
    ((19000, 1), ('cheb_ord2_kernel', 20480, 10, 10)),
]


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


def _sparse64(Lcsr, dev):
    return torch.sparse_csr_tensor(torch.as_tensor(Lcsr.indptr.astype(np.int64)), torch.as_tensor(Lcsr.indices.astype(np.int64)),
                                   torch.as_tensor(Lcsr.data.astype(np.float64)), size=Lcsr.shape).to(dev)


@pytest.mark.parametrize('K', [5, 2, 3])
@pytest.mark.parametrize('graph_key,shape', SHAPES, ids=['n%d_l%d' % k for k, _ in SHAPES])
def test_ordered_recurrence_every_plane(dev, graph_key, shape, K):
    import bench
    from gcn_fmri_decoding_amd import _lib, graph, ops
    from oracle import graph_ref as GR
    lib = _lib.lib()
    family, ENT, NQ, NG = shape[:4]
    nt_small = shape[4] if len(shape) > 4 else None
    PL = 2 if (family == 'cheb_ord2_kernel' or (family == 'cheb_onchip_kernel' and nt_small is None)) else 4
    ordered = family != 'cheb_onchip_kernel'
    if K != 5 and graph_key not in ((1000, 1), (2600, 1), (10000, 1), (11000, 1)):
        pytest.skip('K = 2, 3 on one graph per kernel family and the smallest shape')
    Ls, _ = bench.load_graph(graph_key[0], graph_key[1], 0, 1, None)
    L0 = Ls[0]
    M = L0.shape[0]
    order = graph.length_order(L0)
    g = ops.Graph(L0, dev, order=order)
    if ordered:
        assert g.ordered, 'no ordered kernel shape for M = %d' % M
        assert g.query(16) == PL
        L = graph.permute(L0, order)
    else:
        assert not g.ordered                      # cgcnn then keeps the caller's order (models_gcn.cgcnn.__init__)
        g, L = ops.Graph(L0, dev), L0
        assert g.query(6) == PL
    Mp = g.Mp
    nt = (nt_small or 512) if ordered else (nt_small or 768)
    name_f = '%s<%d,%d,%d,%d,false>' % (family, ENT, NQ, NG, nt)
    name_a = '%s<%d,%d,%d,%d,true>' % (family, ENT, NQ, NG, nt)
    if ordered and Mp > 4 * nt * NQ:              # vertices behind the kernel's quad levels: the streamed tail
        assert g.query(17) == 4 * nt * NQ
        name_f, name_a = name_f + ' + cheb_ord_tail_kernel<false>', name_a + ' + cheb_ord_tail_kernel<true>'
    elif ordered:
        assert g.query(17) == 0
    if len(shape) > 5:
        name_a = shape[5]
    # more plane groups than the launch has workgroups (256 CUs x at most 2 workgroups), partial last group
    B, Fin = (15 if nt_small and ordered else 7), 301 if PL == 4 else 151     # (the 256-thread shapes: four workgroups per CU)
    nplanes = B * Fin
    assert nplanes % PL != 0 and (nplanes + PL - 1) // PL > 2 * 256
    gen = torch.Generator(device=dev)
    gen.manual_seed(graph_key[0] + K)
    x = torch.randn((B, Fin, Mp), generator=gen, device=dev)
    x[:, :, M:] = float('nan')
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    stack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(x), P(stack), B, Fin, K, st), 'fwd copy')
    assert _lib.last_dispatch() == name_f
    stack2 = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    stack2[0].copy_(x)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(stack2), P(stack2), B, Fin, K, st), 'fwd in place')
    assert _lib.last_dispatch() == name_f
    assert torch.equal(stack[:, :, :, :M], stack2[:, :, :, :M]), 'in place and copied T_0 differ'
    assert torch.isfinite(stack[:, :, :, :M]).all()
    del stack2

    Lr = GR.rescale_L(L, 2)
    L64 = Lr.astype(np.float64).tocsr()
    Ld = _sparse64(L64, dev)
    X = x[:, :, :M].double().reshape(nplanes, M).t().contiguous()
    T64 = [X, torch.sparse.mm(Ld, X)]
    for k in range(2, K):
        T64.append(2 * torch.sparse.mm(Ld, T64[-1]) - T64[-2])
    for (b, f) in [(0, 0), (B // 2, 5), (B - 1, Fin - 1)]:       # the float64 device recurrence against the CPU oracle
        t0, t1 = x[b, f, :M].cpu().numpy().astype(np.float64), None
        t1 = L64 @ t0
        for k in range(2, K):
            t0, t1 = t1, 2 * (L64 @ t1) - t0
        got = T64[K - 1][:, b * Fin + f].cpu().numpy()
        assert np.abs(got - t1).max() <= 1e-12 * np.abs(t1).max()
    worst_f = 0.0
    for k in range(K):
        got = stack[k, :, :, :M].reshape(nplanes, M).double()
        ref = T64[k].t()
        per_plane = (got - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)
        worst_f = max(worst_f, float(per_plane.max()))
        assert float(per_plane.max()) <= REL, '%s order %d: plane %d is %.3e from float64' % (
            name_f, k, int(per_plane.argmax()), float(per_plane.max()))
    del X, T64, stack

    G = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    G[:, :, :, M:] = float('nan')
    dx = torch.full((B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_recurrence_bwd(g.handle, P(G), P(dx), B, Fin, K, st), 'bwd')
    assert _lib.last_dispatch() == name_a
    assert torch.isfinite(dx[:, :, :M]).all()
    LTd = _sparse64(L64.T.tocsr(), dev)
    Gk = lambda k: G[k, :, :, :M].double().reshape(nplanes, M).t().contiguous()
    c2, c1 = torch.zeros_like(Gk(0)), Gk(K - 1)
    for j in range(K - 2, 0, -1):
        c2, c1 = c1, Gk(j) + 2 * torch.sparse.mm(LTd, c1) - c2
    dref = (Gk(0) + torch.sparse.mm(LTd, c1) - c2).t()
    gotx = dx[:, :, :M].reshape(nplanes, M).double()
    per_plane = (gotx - dref).abs().amax(dim=1) / dref.abs().amax(dim=1)
    record_measured('ordered_recurrence_every_plane[n%d_l%d,K%d]' % (graph_key[0], graph_key[1], K), kernel=name_f,
                    fwd_worst_plane=worst_f, adjoint_worst_plane=float(per_plane.max()), planes=nplanes)
    assert float(per_plane.max()) <= GREL, '%s: plane %d is %.3e from float64' % (name_a, int(per_plane.argmax()), float(per_plane.max()))


# Every ordered kernel shape the library instantiates: (planes, NG, NQ) -> a random graph with exactly that many active and
# isolated vertices (5 random neighbours per vertex, symmetrised: rows of 5 ... ~16 entries, so the record classes of 8 / 10 / 12
# and more entries all occur; Gaussian weights).  (Two planes, NG = 5, NQ = 5 cannot occur: two planes are used from 10753 vertices.)
ORD_TABLE = [(4, ng, nq) for ng in (2, 3, 4, 5) for nq in (ng, ng + 1)] + [(2, 5, 6)] + [(2, ng, nq) for ng in range(6, 11) for nq in (ng, ng + 1)]
# nq = ng + 2 / ng + 3: more isolated vertices than one quad level -- the NQ = NG + 1 kernel + cheb_ord_tail_kernel
ORD_TABLE += [(4, 2, 4), (4, 5, 7), (2, 6, 8), (4, 3, 6)]
ORD_TABLE = [t + (512,) for t in ORD_TABLE]
# round 6: the 256-thread shapes (planes of more than 1024 vertices, at most 2048 active: NG = 1, 2; NQ = 1 cannot occur -- such a
# plane has at most 1024 vertices), with and without a streamed tail
ORD_TABLE += [(4, 1, 2, 256), (4, 2, 2, 256), (4, 2, 3, 256), (4, 1, 3, 256), (4, 2, 5, 256)]


def _random_graph(n_active, n_iso, seed):
    import scipy.sparse as sp
    from gcn_fmri_decoding_amd import graph
    rs = np.random.RandomState(seed)
    rows = np.repeat(np.arange(n_active), 5)
    cols = rs.randint(0, n_active, rows.size)
    keep = rows != cols
    W = sp.coo_matrix((np.exp(-rs.rand(int(keep.sum())) * 2).astype(np.float32), (rows[keep], cols[keep])), shape=(n_active, n_active)).tocsr()
    W = W.maximum(W.T)
    W = sp.block_diag([W, sp.csr_matrix((n_iso, n_iso), dtype=np.float32)], format='csr')      # isolated vertices: empty rows and columns
    return graph.laplacian(W.astype(np.float32), normalized=True)


@pytest.mark.parametrize('pl,ng,nq,nt', ORD_TABLE, ids=['p%d_ng%d_nq%d' % t[:3] + ('' if t[3] == 512 else '_nt%d' % t[3]) for t in ORD_TABLE])
def test_ordered_shape_table(dev, pl, ng, nq, nt):
    """One launch per instantiated shape, named; every plane of the forward and of the adjoint against float64 on the device
    (K = 4, partial last plane group, more plane groups than workgroups, isolated vertices carrying data)."""
    from gcn_fmri_decoding_amd import _lib, graph, ops
    from oracle import graph_ref as GR
    lib = _lib.lib()
    lvl = 4 * nt                                   # vertices per quad level
    n_active = lvl * (ng - 1) + lvl // 2 - 24 if not (pl == 2 and ng == 5) else 10240
    M = n_active + 4 if nq == ng else (lvl * (nq - 1) + 40 if not (pl == 2 and ng == 5) else 10800)
    L0 = _random_graph(n_active, M - n_active, 100 * pl + 10 * ng + nq)
    order = graph.length_order(L0)
    g = ops.Graph(L0, dev, order=order)
    assert g.ordered and g.query(16) == pl, (g.ordered, g.query(16))
    L = graph.permute(L0, order)
    Mp = g.Mp
    ent = min(lvl * ng + 16, 160 * 1024 // (4 * pl))
    stem = '%s<%d,%d,%d,%d,' % ('cheb_ord_kernel' if pl == 4 else 'cheb_ord2_kernel', ent, min(nq, ng + 1), ng, nt)
    tail_f, tail_a = (' + cheb_ord_tail_kernel<false>', ' + cheb_ord_tail_kernel<true>') if nq > ng + 1 else ('', '')
    assert g.query(17) == (lvl * (ng + 1) if nq > ng + 1 else 0)
    B, Fin, K = (15 if nt == 256 else 7), 301 if pl == 4 else 151, 4
    nplanes = B * Fin
    gen = torch.Generator(device=dev)
    gen.manual_seed(M)
    x = torch.randn((B, Fin, Mp), generator=gen, device=dev)
    x[:, :, M:] = float('nan')
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    stack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(x), P(stack), B, Fin, K, st), 'fwd')
    assert _lib.last_dispatch() == stem + 'false>' + tail_f, _lib.last_dispatch()
    L64 = GR.rescale_L(L, 2).astype(np.float64).tocsr()
    Ld = _sparse64(L64, dev)
    X = x[:, :, :M].double().reshape(nplanes, M).t().contiguous()
    T64 = [X, torch.sparse.mm(Ld, X)]
    for k in range(2, K):
        T64.append(2 * torch.sparse.mm(Ld, T64[-1]) - T64[-2])
    worst_f = 0.0
    for k in range(K):
        got = stack[k, :, :, :M].reshape(nplanes, M).double()
        per_plane = (got - T64[k].t()).abs().amax(dim=1) / T64[k].t().abs().amax(dim=1)
        worst_f = max(worst_f, float(per_plane.max()))
    assert worst_f <= REL, '%s: %.3e' % (stem, worst_f)
    del X, T64
    G = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    G[:, :, :, M:] = float('nan')
    dx = torch.full((B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_recurrence_bwd(g.handle, P(G), P(dx), B, Fin, K, st), 'bwd')
    if nt == 256:                                  # (the Clenshaw adjoint of these graphs: recurrence.hip's kernel, in this vertex order)
        # (beyond 2048 rows in all -- isolated vertices count there --: the four-plane kernel of recurrence4.hip)
        assert _lib.last_dispatch().startswith(('cheb_onchip_kernel<4,', 'cheb4_kernel<')) and ',true' in _lib.last_dispatch(), _lib.last_dispatch()
    else:
        assert _lib.last_dispatch() == stem + 'true>' + tail_a, _lib.last_dispatch()
    LTd = _sparse64(L64.T.tocsr(), dev)
    Gk = lambda k: G[k, :, :, :M].double().reshape(nplanes, M).t().contiguous()
    c2, c1 = torch.zeros_like(Gk(0)), Gk(K - 1)
    for j in range(K - 2, 0, -1):
        c2, c1 = c1, Gk(j) + 2 * torch.sparse.mm(LTd, c1) - c2
    dref = (Gk(0) + torch.sparse.mm(LTd, c1) - c2).t()
    per_plane = (dx[:, :, :M].reshape(nplanes, M).double() - dref).abs().amax(dim=1) / dref.abs().amax(dim=1)
    record_measured('ordered_shape_table[p%d_ng%d_nq%d%s]' % (pl, ng, nq, '' if nt == 512 else '_nt%d' % nt), fwd_worst_plane=worst_f, adjoint_worst_plane=float(per_plane.max()))
    assert float(per_plane.max()) <= GREL, '%s: %.3e' % (stem, float(per_plane.max()))


def test_ordered_recurrence_not_for_atlas_sizes(dev):
    """Up to 1024 vertices per plane the generic on-chip kernel and the fused atlas layer work in the caller's order: a graph in
    length order gets no ordered image there (and cgcnn keeps the reference's tree order, models_gcn.py)."""
    import bench
    from gcn_fmri_decoding_amd import graph, ops
    Ls, _ = bench.load_graph(900, 1, 0, 1, None)
    assert Ls[0].shape[0] <= 1024
    g = ops.Graph(Ls[0], dev, order=graph.length_order(Ls[0]))
    assert not g.ordered and g.query(16) == 0


# (points, coarsening levels, planes asked for) -> template of chebgcn_recurrence_fwd / _bwd on a launch of 6 planes.
# tools/shape_names.py prints this table for a list of sizes; it is every reachable arm of dispatch_onchip / dispatch_onchip4.
ONCHIP = 'cheb_onchip_kernel<%d,%d,%d,%d,%s>'
CALLER_ORDER = [
    ((40, 1, 0), (4, 1, 1, 256)), ((40, 1, 2), (2, 1, 1, 256)), ((360, 1, 0), (4, 2, 1, 256)), ((360, 1, 2), (2, 2, 1, 256)),
    ((500, 1, 2), (2, 4, 1, 256)), ((900, 1, 0), (4, 4, 1, 256)), ((1000, 1, 0), (4, 4, 2, 256)), ((1000, 1, 2), (2, 8, 2, 256)),
    ((1500, 1, 0), (4, 8, 2, 256)), ((2000, 1, 0), (4, 8, 3, 256)), ((2000, 1, 2), (2, 8, 3, 512)), ((4000, 1, 2), (2, 8, 3, 768)),
    ((6000, 1, 2), (2, 11, 4, 768)), ((10000, 1, 2), (2, 14, 4, 768)), ((10500, 1, 2), (2, 24, 7, 512)),
    ((13000, 1, 2), (2, 32, 9, 512)), ((16000, 1, 2), (2, 40, 11, 512)),
    # automatic plane choice beyond 10752 vertices, small launch (7 planes: fewer plane groups than CUs): two planes in both directions
    ((10000, 6, 0), (2, 32, 9, 512)),
    # recurrence4.hip: <entries, rows per thread, pieces per thread, 512, adjoint, isolated vertices in registers>
    ((2600, 1, 4), ('cheb4_kernel<5120,10,3,512,false,true>', 'cheb4_kernel<5120,10,3,512,true,false>')),
    ((5000, 3, 4), ('cheb4_kernel<5120,10,3,512,false,false>', 'cheb4_kernel<5120,10,3,512,true,false>')),
    ((5000, 6, 4), ('cheb4_kernel<5120,10,4,512,false,false>', 'cheb4_kernel<5120,10,4,512,true,false>')),
    ((6000, 1, 4), ('cheb4_kernel<10240,20,6,512,false,true>', 'cheb4_kernel<10240,20,6,512,true,false>')),
    ((9000, 3, 4), ('cheb4_kernel<10240,20,6,512,false,false>', 'cheb4_kernel<10240,20,6,512,true,false>')),
    ((10000, 6, 4), ('cheb4_kernel<10240,20,7,512,false,false>', 'cheb4_kernel<10240,20,7,512,true,false>')),
]


@pytest.mark.parametrize('key,shape', CALLER_ORDER, ids=['n%d_l%d_p%d' % k for k, _ in CALLER_ORDER])
def test_caller_order_recurrence_shape(dev, key, shape):
    """Every plane of a launch of 7 planes (partial last group for either plane count), K = 5, against the CPU oracle
    (``oracle/graph_ref.chebyshev`` = ``lib_new/graph.py:155-172``) and the float64 Clenshaw adjoint; x and G non-zero at the
    coarsening's fake vertices, pads NaN."""
    import bench
    from gcn_fmri_decoding_amd import _lib, ops
    from oracle import graph_ref as GR
    lib = _lib.lib()
    nodes, levels, planes = key
    if len(shape) == 4:
        name_f, name_a = ONCHIP % (shape + ('false',)), ONCHIP % (shape + ('true',))
    else:
        name_f, name_a = shape
    Ls, _ = bench.load_graph(nodes, levels, 0, 1, None)
    L = Ls[0]
    M = L.shape[0]
    g = ops.Graph(L, dev, planes=planes)
    assert not g.ordered
    B, Fin, K, Mp = 1, 7, 5, g.Mp
    gen = torch.Generator(device=dev)
    gen.manual_seed(nodes + planes)
    x = torch.randn((B, Fin, Mp), generator=gen, device=dev)
    x[:, :, M:] = float('nan')
    G = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    G[:, :, :, M:] = float('nan')
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    stack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(x), P(stack), B, Fin, K, st), 'fwd')
    assert _lib.last_dispatch() == name_f
    dx = torch.full((B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_recurrence_bwd(g.handle, P(G), P(dx), B, Fin, K, st), 'bwd')
    assert _lib.last_dispatch() == name_a
    Lr = GR.rescale_L(L, 2)
    ref = GR.chebyshev(Lr, x[0, :, :M].cpu().numpy().T.copy(), K)                 # [K, M, Fin]
    got = stack[:, 0, :, :M].permute(0, 2, 1).cpu().numpy()
    e_f = max(np.abs(got[:, :, f] - ref[:, :, f]).max() / np.abs(ref[:, :, f]).max() for f in range(Fin))
    LT = Lr.T.tocsr().astype(np.float64)
    Gb = G[:, 0, :, :M].cpu().numpy().transpose(0, 2, 1).astype(np.float64)
    c1, c2 = Gb[K - 1], np.zeros_like(Gb[0])
    for j in range(K - 2, 0, -1):
        c1, c2 = Gb[j] + 2 * (LT @ c1) - c2, c1
    dref = Gb[0] + LT @ c1 - c2
    gx = dx[0, :, :M].cpu().numpy().T
    e_a = max(np.abs(gx[:, f] - dref[:, f]).max() / np.abs(dref[:, f]).max() for f in range(Fin))
    record_measured('caller_order_recurrence_shape[n%d_l%d_p%d]' % key, kernel=name_f, fwd_worst_plane=e_f, adjoint_worst_plane=e_a)
    assert e_f <= REL, '%s: %.3e' % (name_f, e_f)
    assert e_a <= GREL, '%s: %.3e' % (name_a, e_a)
