"""Parity of the kernel instantiations the BENCHMARK runs -- the shapes of BASELINE.json
configs[1] and configs[3] on the full-size graph (M = 10466) -- against the CPU oracle,
through the C ABI.  Needs an MI355X: ``-m gpu``.

The layer tests of test_gpu_parity.py run on the N = 212 fixture graphs, where the
dispatchers pick other template instantiations (Fin*K <= 75 there).  The single-layer cases
here run batches of 1..3 windows through autograd: on the 256 CUs of an MI355X those are SMALL
launches (contract.hip ``small_launch``: ceil(M/512) * B < 2 * CUs), i.e. the split kernels.
Each case asserts the templates it reaches through ``chebgcn_last_dispatch()``
(``_lib.dispatch_log``); the big-launch arms (ring / LDS / config 4 / config 5 at batch >= 25)
have their value tests in test_gpu_dispatch.py.

=====================================  =========================================================
case                                   templates (asserted)
=====================================  =========================================================
layer 32 -> 32, K=5, b2relu, B=3       contract_fwd_splitk_kernel, contract_bwd_x_kernel<true,true,true>,
                                       contract_bwd_w_kernel<5,true> + reduce_partials_small,
                                       bias_grad_relu_kernel<VERTEX,4>, cheb_onchip_kernel<2,14,4,768,*>
layer 15 -> 32 (first layer), B=2      contract_bwd_w_kernel<3,true>, no dx
layer 64 -> 64, K=25 (config 4), B=1   contract_fwd_kernel<2>, contract_bwd_x_kernel<false,true,true>,
                                       contract_bwd_w_kernel<5,true> with gy = 10
RT sweep on the N=212 graph            contract_bwd_w_kernel<RT,true> for RT = 1..5 and gy = 2 / 10
PARTS sweep                            brelu_pool_bwd_kernel<BIAS, 1 / 4 / 8> for all three bias kinds
6-layer configs[1] net, B = 2          everything above in sequence: logits, loss, 3 Adam steps
contraction launches at batch 64/256   ring forward, two-phase bwd_x, bwd_w on 768 workgroups (float64 products)
=====================================  =========================================================

Tolerance: max|hip - ref| <= 1e-5 * max|ref| forward (north star: 1e-5 relative fp32);
2e-5 for gradients (two chained fp32 reductions of up to B*M = 2e4 terms), stated per assert.
The oracle's backward has no runnable reference counterpart (TF autodiff): it is checked
against torch.autograd in float64 by tests/test_oracle_layers.py.
"""
import numpy as np
import pytest
import torch

from conftest import assert_adam_params_close, csr_from, load_golden
from oracle import layers_ref as R

pytestmark = pytest.mark.gpu
REL = 1e-5
GREL = 2e-5


def close(got, ref, rel=REL, what=''):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max() / scale
    assert err <= rel, '%s: rel err %.3e > %.1e' % (what, err, rel)
    return err


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops():
    from gcn_fmri_decoding_amd import ops as o
    return o


@pytest.fixture(scope='module')
def bench_graph():
    """The N=10000 benchmark graph (levels = 1): built by the product's host code, which
    tests/test_host_golden.py pins bit-exactly to the reference's output."""
    import bench
    Ls, perm = bench.load_graph(10000, 1, 0, 1, None)
    assert Ls[0].shape[0] == 10466
    return Ls[0]


def to_storage(ops, x_bmf, dev):
    B, M, F = x_bmf.shape
    st = torch.full((B, F, ops.plane_stride(M)), float('nan'), device=dev)
    st[:, :, :M] = torch.as_tensor(np.ascontiguousarray(x_bmf.transpose(0, 2, 1))).to(dev)
    return st


def from_storage(st, M):
    return st[:, :, :M].permute(0, 2, 1).cpu().numpy()


def run_layer(ops, dev, L, B, Fin, Fout, K, p, pool_kind, bias, seed, need_dx=True, wscale=0.3, expect=None, ordered=False):
    """One conv layer forward + backward, HIP against oracle.  Returns the error dict.  ``expect``: entry point ->
    kernel templates (prefix) the dispatchers must have chosen for it (chebgcn_last_dispatch)."""
    from gcn_fmri_decoding_amd import _lib
    _lib.dispatch_log = log = []
    try:
        errs = _run_layer(ops, dev, L, B, Fin, Fout, K, p, pool_kind, bias, seed, need_dx, wscale, ordered)
        torch.cuda.synchronize()
    finally:
        _lib.dispatch_log = None
    for what, name in (expect or {}).items():
        seen = [n for w, n in log if w == what]
        assert seen and all(n.startswith(name) for n in seen), (what, seen, name)
    return errs


def _run_layer(ops, dev, L, B, Fin, Fout, K, p, pool_kind, bias, seed, need_dx=True, wscale=0.3, ordered=False):
    M = L.shape[0]
    if ordered:
        # the same layer on the relabelled graph (graph.length_order): the oracle runs on the relabelled Laplacian too
        from gcn_fmri_decoding_amd import graph as G
        order = G.length_order(L)
        g = ops.Graph(L, dev, order=order)
        assert g.ordered
        L = G.permute(L, order)
    else:
        g = ops.graph_for(L, dev)
    rs = np.random.RandomState(seed)
    x = rs.randn(B, M, Fin).astype(np.float32)
    W = (rs.randn(Fin * K, Fout) * wscale / np.sqrt(Fin * K / 15.0)).astype(np.float32)
    if bias == 1:
        b = (rs.randn(1, 1, Fout) * 0.5).astype(np.float32)
    elif bias == 2:
        b = (rs.randn(1, M, Fout) * 0.5).astype(np.float32)
    else:
        b = np.zeros((1, 1, Fout), np.float32)
    relu = bias != 0
    y, T = R.chebyshev5_fwd(x, L, W, K, return_stack=True)
    a = R.brelu_fwd(y, b) if relu else y + b
    if pool_kind == 0:
        o, arg = R.mpool1_fwd(a, p)
    else:
        o, arg = R.apool1_fwd(a, p), None
    do = rs.randn(*o.shape).astype(np.float32)
    da = R.mpool1_bwd(do, arg, p, M) if pool_kind == 0 else (np.repeat(do, p, axis=1) / p if p > 1 else do)
    if relu:
        dy, db = R.brelu_bwd(da, a, b.shape)
    else:
        dy, db = da, None
    dx_ref, dW_ref = R.chebyshev5_bwd(dy.astype(np.float32), L, W, K, T, need_dx=need_dx)

    xs = to_storage(ops, x, dev).requires_grad_(need_dx)
    Wd = torch.as_tensor(W).to(dev).requires_grad_(True)
    if bias == 1:
        bd = torch.as_tensor(b.reshape(-1)).to(dev).requires_grad_(True)
        kind = ops.BIAS_FILTER
    elif bias == 2:
        bd = torch.zeros((Fout, g.Mp), device=dev)
        bd[:, :M] = torch.as_tensor(b[0].T.copy()).to(dev)
        bd.requires_grad_(True)
        kind = ops.BIAS_VERTEX
    else:
        bd, kind = None, ops.BIAS_NONE
    out = ops.cheb_conv(xs, Wd, bd, g, K, pool=p, pool_kind=pool_kind, relu=relu, bias_kind=kind)
    errs = {'out': close(from_storage(out.detach(), M // p), o, what='out')}
    gout = torch.full(out.shape, float('nan'), device=dev)
    gout[:, :, :M // p] = torch.as_tensor(np.ascontiguousarray(do.transpose(0, 2, 1))).to(dev)
    out.backward(gout)
    if need_dx:
        errs['dx'] = close(from_storage(xs.grad, M), dx_ref, what='dx', rel=GREL)
    else:
        assert xs.grad is None
    errs['dW'] = close(Wd.grad.cpu().numpy(), dW_ref, what='dW', rel=GREL)
    if bias == 1:
        errs['db'] = close(bd.grad.cpu().numpy(), db.reshape(-1), what='db1', rel=GREL)
    elif bias == 2:
        errs['db'] = close(bd.grad[:, :M].t().cpu().numpy(), db[0], what='db2', rel=GREL)
        assert float(bd.grad[:, M:].abs().sum()) == 0.0
    return errs


# ---------------------------------------------------------------------------------------
# single layers at M = 10466
# ---------------------------------------------------------------------------------------

@pytest.fixture(params=['forward', 'clenshaw'])
def dx_form(request, ops, monkeypatch):
    """How a layer with Fout <= Fin forms its gradient wrt the input (ops.dx_by_forward): 'forward' (default) = the forward
    recurrence on the planes of dy with the transposed operator + the forward contraction kernel on the re-indexed weights;
    'clenshaw' = chebgcn_contract_bwd_x + chebgcn_recurrence_bwd.  The same sum, associated the other way round: both are held
    to the oracle."""
    monkeypatch.setattr(ops, 'dx_by_forward', request.param == 'forward')
    return request.param


def test_config2_layer_32_32_k5(ops, dev, bench_graph, dx_form):
    """Layers 2-6 of configs[1]: Fin = Fout = 32, K = 5, b2relu, no pooling."""
    expect = {'contract_fwd': 'contract_fwd_splitk_kernel', 'brelu_pool_bwd': 'bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX,4>',
              'recurrence_fwd': 'cheb_onchip_kernel<2,14,4,768,false>'}
    if dx_form == 'forward':
        # (the graph in length order, as cgcnn builds it: the forward form applies to ordered graphs only)
        # dy is materialised (slab 0 of the stack the recurrence fills): plain weight gradient, forward kernels for dx
        expect.update({'recurrence_fwd': 'cheb_ord_kernel<10240,6,5,512,false>', 'contract_bwd_w': 'contract_bwd_w_kernel<5,false>',
                       'recurrence_fwd_t': 'cheb_ord_kernel<10240,6,5,512,false>', 'contract_bwd_x': 'contract_fwd_splitk_kernel'})
    else:
        expect.update({'contract_bwd_x_relu': 'contract_bwd_x_kernel<true,true,true>', 'contract_bwd_w': 'contract_bwd_w_kernel<5,true>',
                       'recurrence_bwd': 'cheb_onchip_kernel<2,14,4,768,true>'})
    errs = run_layer(ops, dev, bench_graph, B=3, Fin=32, Fout=32, K=5, p=1, pool_kind=0, bias=2, seed=1, expect=expect,
                     ordered=dx_form == 'forward')
    print('config2 layer errors', errs)


def test_config2_first_layer_15_32_k5(ops, dev, bench_graph):
    """Layer 1 of configs[1]: block_dura = 15 input features, no gradient wrt the input."""
    run_layer(ops, dev, bench_graph, B=2, Fin=15, Fout=32, K=5, p=1, pool_kind=0, bias=2, seed=2, need_dx=False,
              expect={'contract_fwd': 'contract_fwd_splitk_kernel', 'contract_bwd_w': 'contract_bwd_w_kernel<3,true>'})


def test_config4_layer_64_64_k25(ops, dev, bench_graph, dx_form):
    """configs[3]: K = 25, Fin = Fout = 64 -- Fin*K = 1600 (50 row tiles -> gy = 10 in bwd_w),
    Fout = 64 (two filter tiles in contract_fwd, dy streamed instead of held in bwd_x); one window = a small launch
    (the big-launch arms of this shape: test_gpu_dispatch.py ``config4_b25``).  (``ops.cheb_conv``'s own default precision is
    'f32': these are the fp32 kernels; cgcnn's 'auto' would compute this shape in split bf16.)"""
    expect = {'contract_fwd': 'contract_fwd_kernel<2>'}
    if dx_form == 'forward':
        expect.update({'contract_bwd_w': 'contract_bwd_w_kernel<5,false>', 'contract_bwd_x': 'contract_fwd_kernel<2>'})
    else:
        expect.update({'contract_bwd_x_relu': 'contract_bwd_x_kernel<false,true,true>', 'contract_bwd_w': 'contract_bwd_w_kernel<5,true>'})
    run_layer(ops, dev, bench_graph, B=1, Fin=64, Fout=64, K=25, p=1, pool_kind=0, bias=2, seed=3, expect=expect,
              ordered=dx_form == 'forward')


def test_pooled_layer_full_size(ops, dev, bench_graph):
    """p = 2 on the full-size graph with per-filter biases (brelu_pool_bwd<FILTER>, argmax path)."""
    run_layer(ops, dev, bench_graph, B=2, Fin=4, Fout=32, K=3, p=2, pool_kind=0, bias=1, seed=4)


# ---------------------------------------------------------------------------------------
# dispatcher sweeps on the small fixture graph (cheap), value comparisons
# ---------------------------------------------------------------------------------------

def levels(name='layers_n212'):
    z = load_golden(name)
    return [csr_from(z, 'L%d' % i) for i in range(4)]


@pytest.mark.parametrize('Fin,K,rt,gy', [(4, 5, 1, 1), (8, 5, 2, 1), (13, 7, 3, 1), (32, 4, 4, 1), (32, 5, 5, 1),
                                          (11, 17, 5, 2), (64, 25, 5, 10)])
def test_bwd_w_row_tile_sweep(ops, dev, Fin, K, rt, gy, monkeypatch):
    """contract_bwd_w_kernel<RT> for RT = 1..5 and several row-tile groups (gy > 1), the ReluGrad folded in (the Clenshaw form
    of the input gradient: with ``dx_by_forward`` the last shape, Fin = 64 > Fout = 40, would materialise dy)."""
    monkeypatch.setattr(ops, 'dx_by_forward', False)
    ntiles = (Fin * K + 31) // 32
    assert min(ntiles, 5) == rt and (ntiles + rt - 1) // rt == gy      # the dispatcher's arithmetic (contract.hip bw_rt)
    run_layer(ops, dev, levels()[0], B=2, Fin=Fin, Fout=40, K=K, p=1, pool_kind=0, bias=2, seed=Fin + K,
              expect={'contract_bwd_w': 'contract_bwd_w_kernel<%d,true>' % rt})


@pytest.mark.parametrize('lvl,F,parts', [(0, 256, 4), (0, 24, 8), (2, 33, 8)])
@pytest.mark.parametrize('bias', [0, 1, 2])
def test_brelu_pool_bwd_parts_small_graph(ops, dev, lvl, F, parts, bias):
    """brelu_pool_bwd_kernel<BIAS, PARTS> with PARTS = 4 / 8 (small graphs), all bias kinds, with
    pooling.  (From 2048 vertices a pooled layer takes the 16-byte-store kernel pool_scatter_bwd_kernel: the M = 10466 test
    below and tests/test_gpu_round6.py.)"""
    L = levels()[lvl]
    M = L.shape[0]
    got = 1 if ((M + 255) // 256) * F >= 1024 else 4 if ((M + 63) // 64) * F >= 1024 else 8
    assert got == parts                                                # the dispatcher's arithmetic (pointwise.hip)
    kind = ['CHEBGCN_BIAS_NONE', 'CHEBGCN_BIAS_FILTER', 'CHEBGCN_BIAS_VERTEX'][bias]
    run_layer(ops, dev, L, B=5, Fin=2, Fout=F, K=2, p=2, pool_kind=0, bias=bias, seed=F + bias,
              expect={'brelu_pool_bwd': 'brelu_pool_bwd_kernel<%s,%d>' % (kind, parts)})


@pytest.mark.parametrize('bias,p,pool_kind', [(0, 1, 0), (1, 1, 0), (2, 2, 0), (2, 2, 1)])     # M = 10466 = 2 * 5233
def test_brelu_pool_bwd_parts1_full_size(ops, dev, bench_graph, bias, p, pool_kind):
    """PARTS = 1 (41 * 32 >= 1024 blocks) for the bias kinds / pooling forms not covered above."""
    kind = ['CHEBGCN_BIAS_NONE', 'CHEBGCN_BIAS_FILTER', 'CHEBGCN_BIAS_VERTEX'][bias]
    # (pool 1 with ReLU takes the bit-mask kernel instead; without ReLU -- bias 0 -- and with pooling: PARTS = 1)
    name = ('bias_grad_relu_kernel<%s,4>' % kind if (p == 1 and bias != 0) else 'brelu_pool_bwd_kernel<%s,1>' % kind if p == 1
            else 'pool_scatter_bwd_kernel<%s> + pool_bias_reduce_kernel<%s>' % (kind, kind))
    run_layer(ops, dev, bench_graph, B=2, Fin=3, Fout=32, K=2, p=p, pool_kind=pool_kind, bias=bias, seed=10 * bias + p,
              expect={'brelu_pool_bwd': name})


# ---------------------------------------------------------------------------------------
# the whole configs[1] network at full size
# ---------------------------------------------------------------------------------------

def test_config2_network_b2_vs_oracle(ops, dev, bench_graph):
    """BASELINE configs[1] (6 x [K=5, F=32, p=1, b2relu], FC 512-256-22, block_dura 15) with a
    batch of 2 windows: logits, loss and three TF-form Adam steps against the oracle."""
    from gcn_fmri_decoding_amd import models_gcn
    L = bench_graph
    M = L.shape[0]
    F, K, p, Mfc, channel, B = [32] * 6, [5] * 6, [1] * 6, [512, 256, 22], 15, 2
    reg = 5e-4
    net = models_gcn.cgcnn({'device': dev}, [L] * 6, F, K, p, Mfc, filter='chebyshev5', brelu='b2relu', pool='mpool1',
                           initial='he', channel=channel, regularization=reg, dropout=1, batch_size=B, verbose=False)
    onet = R.Net([L], F, K, p, Mfc, channel=channel, brelu='b2relu', regularization=reg)
    rs = np.random.RandomState(0)
    params = {}
    for k, s in onet.param_shapes().items():
        params[k] = ((0.2 + 0.05 * rs.randn(*s)) if k.endswith('bias') else rs.randn(*s) * np.sqrt(2.0 / s[0])).astype(np.float32)
        net.set_variable(k, params[k])
    x = rs.randn(B, M, channel).astype(np.float32)
    labels = np.array([3, 17])
    xs = to_storage(ops, x, dev)
    ld = torch.as_tensor(labels).to(dev)
    with torch.no_grad():
        logits = net._inference_storage(xs, 1).cpu().numpy()
    ologits, _ = onet.forward(params, x)
    close(logits, ologits, rel=GREL, what='logits')
    state, ill = {}, {}
    for step in range(3):
        ologits, cache = onet.forward(params, x)
        loss, dlogits = onet.loss(params, ologits, labels)
        grads = onet.backward(params, cache, dlogits)
        _, loss_avg = net.train_step(xs, ld)
        if step == 0:
            assert abs(float(loss_avg) - 0.1 * loss) <= GREL * abs(0.1 * loss)      # (first read of the zero-initialised 0.9-EMA)
            for k in params:
                gk = net.gradient(k)
                ref = grads[k] - (reg * params[k] if onet.regularized(k) else 0)
                close(gk.cpu().numpy(), ref, rel=5e-5, what='grad ' + k)
        R.adam_tf_step(params, grads, state)
        for k in params:
            # Adam moments are linear / quadratic in the gradient: tight bounds
            def ref_shape(flat):
                return net._ref_view(flat, k).cpu().numpy()
            # step 0: the same variables on both sides; later steps start from variables that already differ in
            # their ill-conditioned elements (below), so the gradients agree less tightly
            if step == 0:
                close(ref_shape(net._adam_m), state['m/' + k], rel=5e-5, what='step 0 m ' + k)
                close(ref_shape(net._adam_v), state['v/' + k], rel=1e-4, what='step 0 v ' + k)
            else:
                # a ReLU that flips on one side only (its pre-activation within round-off of 0) changes single
                # gradient elements outright: all but 0.1 % of the elements within 5e-3 / 1e-2 of the scale
                for flat, key, rel in ((net._adam_m, 'm/', 5e-3), (net._adam_v, 'v/', 1e-2)):
                    ref = state[key + k].astype(np.float64)
                    d = np.abs(ref_shape(flat).astype(np.float64) - ref)
                    assert np.quantile(d, 0.999) <= rel * np.abs(ref).max(), 'step %d %s%s' % (step, key, k)
            assert_adam_params_close(net.get_var(k), params[k], state['v/' + k], step, ill, k, rel=GREL if step == 0 else 1e-4, quantile=1.0 if step == 0 else 0.999)

# ---------------------------------------------------------------------------------------
# the pooling ChebNet of the legacy monolith at full size
# ---------------------------------------------------------------------------------------

@pytest.mark.parametrize('contraction', ['f32', 'auto'])
def test_pooling_chebnet_full_size_vs_oracle(ops, dev, contraction):
    """``contraction``: 'f32' = exact products in every layer; 'auto' (cgcnn's default) = split bf16 in the layers of 64 and
    128 filters (ops.resolve_precision): each contraction is within 1e-5 of float64 (test_split_bf16_arm_vs_float64), the
    logits of the six-layer network within 2e-5; gradients that pass through max-pooling picks and ReLU decisions which flip on
    a 5e-6 difference are held to a quantile bound, the measured values are recorded.

    HCP_task_fmri_gcn_test8.py:1633-1636, 2071: six coarsening levels of the N = 10000 graph
    (M = 12672 / 6336 / ... / 198), p = [1,4,1,4,1,4], K = [20,10,10,10,5,5], F = [32,32,64,64,128,128],
    b2relu, block_dura 15, batch 2: logits, loss and the gradients of one training step against
    the oracle (layers at M = 12672, 3168 and 792 vertices with Fout up to 128 and max pooling 4)."""
    import bench
    from gcn_fmri_decoding_amd import models_gcn
    Ls, perm = bench.load_graph(10000, 6, 0, 1, None)
    assert [L.shape[0] for L in Ls] == [12672, 6336, 3168, 1584, 792, 396, 198] and len(perm) == 12672
    F, K, p, Mfc, channel, B = [32, 32, 64, 64, 128, 128], [20, 10, 10, 10, 5, 5], [1, 4, 1, 4, 1, 4], [512, 256, 22], 15, 2
    reg = 5e-4
    net = models_gcn.cgcnn({'device': dev}, Ls, F, K, p, Mfc, filter='chebyshev5', brelu='b2relu', pool='mpool1',
                           initial='he', channel=channel, regularization=reg, dropout=1, batch_size=B, verbose=False)
    net.contraction = contraction
    assert [ops.resolve_precision(contraction, fi, k, fo) for fi, k, fo in zip([channel] + F[:-1], K, F)] == (
        ['f32'] * 6 if contraction == 'f32' else ['f32', 'f32', 'bf16x3', 'bf16x3', 'bf16x3', 'bf16x3'])
    assert [g.M for g in net.graphs] == [12672, 12672, 3168, 3168, 792, 792]
    onet = R.Net(Ls, F, K, p, Mfc, channel=channel, brelu='b2relu', regularization=reg)
    rs = np.random.RandomState(5)
    params = {}
    for k, s in onet.param_shapes().items():
        params[k] = ((0.2 + 0.05 * rs.randn(*s)) if k.endswith('bias') else rs.randn(*s) * np.sqrt(2.0 / s[0])).astype(np.float32)
        net.set_variable(k, params[k])
    # input through the product's staging path: perm_data_3d on the device
    raw = rs.randn(B, 10000, channel).astype(np.float32)
    from oracle import coarsening_ref as CR
    x = CR.perm_data_3d(raw, perm.tolist()).astype(np.float32)
    xs = ops.perm_data(torch.as_tensor(raw).to(dev), torch.as_tensor(perm.astype(np.int32)).to(dev))
    assert np.array_equal(from_storage(xs, 12672), x)
    labels = np.array([20, 4])
    with torch.no_grad():
        logits = net._inference_storage(xs, 1).cpu().numpy()
    ologits, cache = onet.forward(params, x)
    close(logits, ologits, rel=GREL, what='logits')
    loss, dlogits = onet.loss(params, ologits, labels)
    grads = onet.backward(params, cache, dlogits)
    from gcn_fmri_decoding_amd import _lib
    _lib.dispatch_log = log = []
    try:
        _, loss_avg = net.train_step(xs, torch.as_tensor(labels).to(dev))
        torch.cuda.synchronize()
    finally:
        _lib.dispatch_log = None
    assert abs(float(loss_avg) - 0.1 * loss) <= GREL * abs(0.1 * loss)      # (first read of the zero-initialised 0.9-EMA)
    # Which kernels ran (round 6): levels 0 (10000 active of 12672 vertices) and 2 (2721 of 3168) are relabelled by descending
    # row length and run the ORDERED recurrence kernels -- level 0 with the streamed tail for its 2672 fake vertices --, their
    # pooled layers pool through index maps; level 4 (792 vertices) keeps the tree order and the caller-order kernels
    assert net.vertex_order == 'length' and [o is not None for o in net._orders] == [True, True, True, True, False, False]
    seen = {}
    for what, name in log:
        seen.setdefault(what, set()).add(name)
    assert seen['recurrence_fwd'] == {'cheb_ord_kernel<10240,6,5,512,false> + cheb_ord_tail_kernel<false>',
                                      'cheb_ord_kernel<4112,2,2,512,false>', 'cheb_onchip_kernel<4,4,1,256,false>'}, seen['recurrence_fwd']
    # input gradients: layers 2, 4 (Fout <= Fin, ordered level) by the forward recurrence on dy; layer 3 (32 -> 64) Clenshaw on the
    # ordered kernel; layers 5, 6 on the caller-order kernel
    assert seen['recurrence_fwd_t'] == {'cheb_ord_kernel<10240,6,5,512,false> + cheb_ord_tail_kernel<false>',
                                        'cheb_ord_kernel<4112,2,2,512,false>'}, seen['recurrence_fwd_t']
    assert seen['recurrence_bwd'] == {'cheb_ord_kernel<4112,2,2,512,true>', 'cheb_onchip_kernel<4,4,1,256,true>'}, seen['recurrence_bwd']
    assert seen['pool_gather_fwd'] == {'pool_gather_fwd_kernel<map>'} and seen['pool_scatter_bwd'] == {
        'pool_scatter_bwd_kernel<CHEBGCN_BIAS_VERTEX><map> + pool_bias_reduce_kernel<CHEBGCN_BIAS_VERTEX>'}, (seen['pool_gather_fwd'], seen['pool_scatter_bwd'])
    # (level 4 -> 6, tree order on both sides, 792 vertices: below 2048 vertices the scalar kernel with its batch split is the faster one)
    assert seen['brelu_pool_bwd'] == {'brelu_pool_bwd_kernel<CHEBGCN_BIAS_VERTEX,4>', 'bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX,4>'}, seen['brelu_pool_bwd']
    from conftest import record_measured
    measured = {}
    for k in params:
        gk = net.gradient(k).cpu().numpy().astype(np.float64)
        ref = grads[k] - (reg * params[k] if onet.regularized(k) else 0)
        d = np.abs(gk - ref) / max(np.abs(ref).max(), 1e-30)
        measured[k] = [float(np.quantile(d, 0.99)), float(d.max())]
    record_measured('pooling_chebnet_full_size[%s]' % contraction, **measured)
    for k, (q99, worst) in measured.items():
        if contraction == 'f32':
            assert worst <= 5e-5, 'grad %s: rel err %.3e' % (k, worst)
        else:
            # Measured (profiles/r05_parity_measured.jsonl): the gradients of the split-bf16 layers 4-6 and of the head are 1e-6 ... 7e-6
            # from the oracle in EVERY element -- the arithmetic is fp32-grade.  Layer 3 and everything below it differ by what a
            # handful of flipped ReLU decisions do: a pre-activation within 5e-6 of zero takes the other side (expected: ~4 of the
            # 405 504 of layer 3 with two windows), the bias gradient at that vertex then differs by a whole dy (one element of
            # conv3/bias 5.5e-2 off, its 99 % quantile 1.3e-6), and the weight gradients below it by that one term of their sums
            # (conv1/weights 7e-4 at the 99 % quantile, 1.2e-3 max).  The same happens between fp32 and float64, fifty times
            # more rarely.  The bound is therefore on the quantile; cgcnn.contraction = 'f32' is the exact path
            assert q99 <= 2e-3, 'grad %s: 99 %% quantile %.3e, max %.3e' % (k, q99, worst)


# ---------------------------------------------------------------------------------------
# the north-star launch itself: K = 5, Fin = 32, batch 256 (8192 planes: the four-plane kernels)
# ---------------------------------------------------------------------------------------

# north star; BASELINE configs[3] (bench.py `config4`); the two recurrence launches of the bench step (batch 64: 2048 and
# 960 planes, which pick_ell sends to the two-plane kernel cheb_onchip_kernel<2,14,4,768,*>)
@pytest.mark.parametrize('ordered', [False, True])
@pytest.mark.parametrize('B,Fin,K', [(256, 32, 5), (64, 64, 25), (64, 32, 5), (64, 15, 5)])
def test_northstar_launch_properties(ops, dev, bench_graph, B, Fin, K, ordered):
    """The launches bench.py's ``northstar`` object times (BASELINE.json's north-star shape: K=5
    recurrence, Fin=32, batch 256, M=10466 -- cheb4_kernel<10240,20,6,512,false/true>), checked at
    full size through what does not need a 1.7 GB oracle run:
    * planes drawn from all over the batch agree with the CPU oracle for every order k (1e-5);
    * EVERY plane of every order, and every plane of the adjoint, against the same recurrence in float64 on the GPU
      (torch.sparse, tied to the oracle on three planes at 1e-12): 1e-5 / 2e-5 of the plane's own maximum;
    * in place (T_0 already in slab 0, as the model runs it) and with the copy of x: the same bits;
    * the adjoint identity  sum_k <T_k(L~) x, G_k> = <x, recurrence_bwd(G)>  over ALL 8192 planes
      (float64 sums of fp32 results, 1e-5 of the magnitude sum);
    * linearity of the forward launch (1e-5)."""
    import ctypes
    from gcn_fmri_decoding_amd import _lib
    from oracle import graph_ref as GR
    lib = _lib.lib()
    L = bench_graph
    M = L.shape[0]
    if ordered:
        # the graph as cgcnn builds it for a network without pooling: vertices relabelled by descending row length
        # (cheb_ord_kernel<10240,6,5,512,*>, csrc/recurrence_ord.hip; what the bench step and `northstar` time since round 4)
        from gcn_fmri_decoding_amd import graph as G
        order = G.length_order(L)
        g = ops.Graph(L, dev, order=order)
        assert g.ordered
        L = G.permute(L, order)
    else:
        g = ops.graph_for(L, dev)
        assert not g.ordered
    assert g.query(6) == 4                                              # the automatic graph carries four planes ...
    Mp = g.Mp
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    x = torch.randn((B, Fin, Mp), generator=gen, device=dev)
    x[:, :, M:] = float('nan')                                         # pads must never leak
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    stack = torch.empty((K, B, Fin, Mp), device=dev)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(x), P(stack), B, Fin, K, st), 'fwd copy')
    stack2 = torch.empty((K, B, Fin, Mp), device=dev)
    stack2[0].copy_(x)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(stack2), P(stack2), B, Fin, K, st), 'fwd in place')
    assert torch.equal(stack[:, :, :, :M], stack2[:, :, :, :M]), 'in place and copied T_0 differ'
    assert torch.isfinite(stack[:, :, :, :M]).all()

    # oracle on a handful of planes (first, last, around group boundaries)
    Lr = GR.rescale_L(L, 2)
    for (b, f) in [(0, 0), (0, 3), (0, 4), (17, Fin - 1), (B // 2, 5), (B - 1, Fin - 4), (B - 1, Fin - 1)]:
        xv = x[b, f, :M].cpu().numpy().astype(np.float32)
        T = [xv, (Lr @ xv).astype(np.float32)]
        for k in range(2, K):
            T.append((2 * (Lr @ T[-1]) - T[-2]).astype(np.float32))
        for k in range(K):
            close(stack[k, b, f, :M].cpu().numpy(), T[k], what='plane (%d,%d) order %d' % (b, f, k))

    # EVERY plane and order against the same recurrence in float64 on the GPU (torch.sparse CSR x dense: an implementation that
    # shares nothing with the library), after that float64 recurrence has itself been tied to the CPU oracle on the planes
    # above; the bound holds PER PLANE (max over the plane's vertices), not for the tensor as a whole
    L64 = Lr.astype(np.float64).tocsr()
    Ld = torch.sparse_csr_tensor(torch.as_tensor(L64.indptr.astype(np.int64)), torch.as_tensor(L64.indices.astype(np.int64)),
                                 torch.as_tensor(L64.data), size=L64.shape).to(dev)
    X = x[:, :, :M].double().reshape(B * Fin, M).t().contiguous()                       # [M, planes]
    T64 = [X, torch.sparse.mm(Ld, X)]
    for k in range(2, K):
        T64.append(2 * torch.sparse.mm(Ld, T64[-1]) - T64[-2])
    for (b, f) in [(0, 0), (B // 2, 5), (B - 1, Fin - 1)]:
        xv = x[b, f, :M].cpu().numpy().astype(np.float64)
        t0, t1 = xv, L64 @ xv
        for k in range(2, K):
            t0, t1 = t1, 2 * (L64 @ t1) - t0
        ref = t1 if K > 1 else t0
        got = T64[K - 1][:, b * Fin + f].cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), 'float64 torch recurrence differs from the oracle'
    worst = 0.0
    for k in range(K):
        got = stack[k, :, :, :M].reshape(B * Fin, M).double()
        ref = T64[k].t()
        per_plane = (got - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)
        worst = max(worst, float(per_plane.max()))
        assert float(per_plane.max()) <= REL, 'order %d: plane %d is %.3e from float64' % (k, int(per_plane.argmax()), float(per_plane.max()))
    del X

    # adjoint over the whole launch: every plane of dx against the float64 Clenshaw recurrence with L~^T
    G = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    G[:, :, :, M:] = float('nan')
    dx = torch.empty((B, Fin, Mp), device=dev)
    _lib.check(lib.chebgcn_recurrence_bwd(g.handle, P(G), P(dx), B, Fin, K, st), 'bwd')
    assert torch.isfinite(dx[:, :, :M]).all()
    LT64 = L64.T.tocsr()
    LTd = torch.sparse_csr_tensor(torch.as_tensor(LT64.indptr.astype(np.int64)), torch.as_tensor(LT64.indices.astype(np.int64)),
                                  torch.as_tensor(LT64.data), size=LT64.shape).to(dev)
    Gk = lambda k: G[k, :, :, :M].double().reshape(B * Fin, M).t().contiguous()
    if K == 1:
        dref = Gk(0)
    else:
        c2, c1 = torch.zeros_like(Gk(0)), Gk(K - 1)                                     # c_{j+2}, c_{j+1}
        for j in range(K - 2, 0, -1):
            c2, c1 = c1, Gk(j) + 2 * torch.sparse.mm(LTd, c1) - c2
        dref = Gk(0) + torch.sparse.mm(LTd, c1) - c2
    gotx = dx[:, :, :M].reshape(B * Fin, M).double()
    per_plane = (gotx - dref.t()).abs().amax(dim=1) / dref.t().abs().amax(dim=1)
    from conftest import record_measured
    record_measured('northstar_launch_every_plane[%d,%d,%d,%s]' % (B, Fin, K, 'ordered' if ordered else 'reference'),
                    fwd_worst_plane=worst, adjoint_worst_plane=float(per_plane.max()), planes=B * Fin)
    assert float(per_plane.max()) <= 2e-5, 'adjoint: plane %d is %.3e from float64' % (int(per_plane.argmax()), float(per_plane.max()))
    del dref, c1, c2
    lhs = float((stack[:, :, :, :M].double() * G[:, :, :, :M].double()).sum())
    rhs = float((x[:, :, :M].double() * dx[:, :, :M].double()).sum())
    mag = float((stack[:, :, :, :M].double() * G[:, :, :, :M].double()).abs().sum())
    assert abs(lhs - rhs) <= 1e-5 * mag, (lhs, rhs, mag)

    # linearity: T(a x + y) = a T(x) + T(y)
    y = torch.randn((B, Fin, Mp), generator=gen, device=dev)
    sy = torch.empty_like(stack)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(y), P(sy), B, Fin, K, st), 'fwd y')
    z = (0.5 * x + y)
    sz = torch.empty_like(stack)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(z), P(sz), B, Fin, K, st), 'fwd z')
    ref = 0.5 * stack[:, :, :, :M] + sy[:, :, :, :M]
    err = float((sz[:, :, :, :M] - ref).abs().max() / ref.abs().max())
    assert err <= REL, 'linearity: %.3e' % err


def test_contraction_bench_launch_b64(ops, dev):
    """The contraction launches of the bench step at their real size (batch 64, M = 10466, Fin = Fout = 32,
    K = 5: contract_fwd_ring_kernel with the ReLU bit mask, contract_bwd_w_kernel<5,true> on 768 workgroups,
    contract_bwd_x_lds_kernel<true>, bias_grad_relu_kernel) against float64 matrix
    products of the same operands computed on the device (1e-5 / 2e-5 of max, pads poisoned)."""
    import ctypes
    from gcn_fmri_decoding_amd import _lib
    lib = _lib.lib()
    B, M, Fin, K, Fout = 64, 10466, 32, 5, 32
    Mp = ops.plane_stride(M)
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    stack = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    W = torch.randn((Fin * K, Fout), generator=gen, device=dev) * 0.1
    bias = torch.zeros((Fout, Mp), device=dev)
    bias[:, :M] = torch.randn((Fout, M), generator=gen, device=dev) * 0.3
    stack[..., M:] = float('nan')
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    out = torch.full((B, Fout, Mp), float('nan'), device=dev)
    mask = torch.zeros((B, Fout, Mp // 4), dtype=torch.uint8, device=dev)
    _lib.check(lib.chebgcn_contract_fwd(P(stack), P(W), P(bias), ops.BIAS_VERTEX, P(out), P(mask), B, M, Fin, K, Fout, 1, 0, 1, st), 'fwd')
    S = stack[..., :M].permute(2, 0, 1, 3).reshape(Fin * K, B, M).double()          # rows fin*K + k
    pre = torch.einsum('rbm,ro->bom', S, W.double()) + bias[:, :M].double()
    ref = pre.clamp(min=0)
    err = float((out[..., :M].double() - ref).abs().max() / pre.abs().max())
    assert err <= REL, 'contract_fwd: %.3e' % err
    bits = torch.stack([(mask >> r) & 1 for r in range(4)], -1).reshape(B, Fout, Mp)[..., :M].bool()
    assert torch.equal(bits, out[..., :M] > 0), 'ReLU bit mask disagrees with the output'

    gout = torch.randn((B, Fout, Mp), generator=gen, device=dev)
    gout[..., M:] = float('nan')
    dy = (gout[..., :M] * bits).double()                                                # ReluGrad
    n = lib.chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    dW = torch.full((Fin * K, Fout), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_w_relu(P(stack), P(gout), P(mask), P(dW), P(ws), n, B, M, Fin, K, Fout, st), 'bwd_w')
    dW_ref = torch.einsum('rbm,bom->ro', S, dy)
    err = float((dW.double() - dW_ref).abs().max() / dW_ref.abs().max())
    assert err <= GREL, 'contract_bwd_w_relu: %.3e' % err
    del S

    gstack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_x_relu(P(gout), P(mask), P(W), P(gstack), B, M, Fin, K, Fout, st), 'bwd_x')
    gs_ref = torch.einsum('ro,bom->rbm', W.double(), dy).reshape(Fin, K, B, M).permute(1, 2, 0, 3)
    err = float((gstack[..., :M].double() - gs_ref).abs().max() / gs_ref.abs().max())
    assert err <= GREL, 'contract_bwd_x_relu: %.3e' % err

    dbias = torch.full((Fout, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_brelu_pool_bwd(P(gout), None, P(mask), None, P(dbias), ops.BIAS_VERTEX, B, M, Fout, 1, 0, 1, None, 0, st), 'bias')
    db_ref = dy.sum(0)
    err = float((dbias[:, :M].double() - db_ref).abs().max() / db_ref.abs().max())
    assert err <= GREL, 'bias gradient: %.3e' % err
    assert float(dbias[:, M:].abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('B,M,Fin,K,Fout,bias_kind,relu', [
    (256, 1050, 32, 5, 32, 'vertex', 1),      # Mp = 1056 = 8 * 128 + 32: one of four accumulators of the last tile inside the plane
    (200, 1200, 15, 5, 32, 'filter', 1),      # Fin*K = 75: padded ring rounds forward, partial row tiles backward; 64 of 128 inside
    (130, 2000, 32, 10, 24, 'none', 0),       # K = 10 (320 rows of W in LDS), Fout < 32, no ReLU (plain gradients); 96 of 128 inside
])
def test_contraction_big_launch_edges(ops, dev, B, M, Fin, K, Fout, bias_kind, relu):
    """Big-launch contraction kernels (the ring forward, the two-phase bwd_x, bwd_w) where a window's last 128-vertex tile
    sticks out of the plane, Fin*K is not a multiple of the ring round / row tile, Fout < 32, with and without the ReLU
    mask -- against float64 products of the same operands (1e-5 / 2e-5 of max, pads poisoned with NaN)."""
    import ctypes
    from gcn_fmri_decoding_amd import _lib
    lib = _lib.lib()
    Mp = ops.plane_stride(M)
    assert Mp % 128 != 0
    gen = torch.Generator(device=dev)
    gen.manual_seed(B + M)
    stack = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    stack[..., M:] = float('nan')
    W = torch.randn((Fin * K, Fout), generator=gen, device=dev) * 0.1
    kind = {'vertex': ops.BIAS_VERTEX, 'filter': ops.BIAS_FILTER, 'none': ops.BIAS_NONE}[bias_kind]
    bias, bias_ref = None, 0.0
    if bias_kind == 'vertex':
        bias = torch.zeros((Fout, Mp), device=dev)
        bias[:, :M] = torch.randn((Fout, M), generator=gen, device=dev) * 0.3
        bias_ref = bias[:, :M].double()
    elif bias_kind == 'filter':
        bias = torch.randn((Fout,), generator=gen, device=dev) * 0.3
        bias_ref = bias.double()[:, None]
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    out = torch.full((B, Fout, Mp), float('nan'), device=dev)
    mask = torch.zeros((B, Fout, Mp // 4), dtype=torch.uint8, device=dev) if relu else None
    _lib.check(lib.chebgcn_contract_fwd(P(stack), P(W), P(bias), kind, P(out), P(mask), B, M, Fin, K, Fout, 1, 0, relu, st), 'fwd')
    S = stack[..., :M].permute(2, 0, 1, 3).reshape(Fin * K, B, M).double()          # rows fin*K + k
    pre = torch.einsum('rbm,ro->bom', S, W.double()) + bias_ref
    ref = pre.clamp(min=0) if relu else pre
    err = float((out[..., :M].double() - ref).abs().max() / pre.abs().max())
    assert err <= REL, 'contract_fwd: %.3e' % err

    gout = torch.randn((B, Fout, Mp), generator=gen, device=dev)
    gout[..., M:] = float('nan') if relu else 0.0        # (without a mask the pad columns of dy are the caller's zeros)
    if relu:
        bits = torch.stack([(mask >> r) & 1 for r in range(4)], -1).reshape(B, Fout, Mp)[..., :M].bool()
        assert torch.equal(bits, out[..., :M] > 0), 'ReLU bit mask disagrees with the output'
        dy = (gout[..., :M] * bits).double()
    else:
        dy = gout[..., :M].double()
    n = lib.chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    dW = torch.full((Fin * K, Fout), float('nan'), device=dev)
    if relu:
        _lib.check(lib.chebgcn_contract_bwd_w_relu(P(stack), P(gout), P(mask), P(dW), P(ws), n, B, M, Fin, K, Fout, st), 'bwd_w')
    else:
        _lib.check(lib.chebgcn_contract_bwd_w(P(stack), P(gout), P(dW), P(ws), n, B, M, Fin, K, Fout, st), 'bwd_w')
    dW_ref = torch.einsum('rbm,bom->ro', S, dy)
    err = float((dW.double() - dW_ref).abs().max() / dW_ref.abs().max())
    assert err <= GREL, 'contract_bwd_w: %.3e' % err
    del S

    gstack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    if relu:
        _lib.check(lib.chebgcn_contract_bwd_x_relu(P(gout), P(mask), P(W), P(gstack), B, M, Fin, K, Fout, st), 'bwd_x')
    else:
        _lib.check(lib.chebgcn_contract_bwd_x(P(gout), P(W), P(gstack), B, M, Fin, K, Fout, st), 'bwd_x')
    gs_ref = torch.einsum('ro,bom->rbm', W.double(), dy).reshape(Fin, K, B, M).permute(1, 2, 0, 3)
    err = float((gstack[..., :M].double() - gs_ref).abs().max() / gs_ref.abs().max())
    assert err <= GREL, 'contract_bwd_x: %.3e' % err
    # nothing was written beyond the planes: the next plane's first columns would be hit first
    assert bool(torch.isfinite(gstack[..., :M]).all())


@pytest.mark.gpu
@pytest.mark.parametrize('B,M,Fin,K,Fout,bias_kind', [(64, 10466, 32, 5, 32, 'vertex'), (256, 1050, 15, 5, 24, 'filter')])
def test_last_layer_with_feature_mean(ops, dev, B, M, Fin, K, Fout, bias_kind):
    """chebgcn_contract_fwd_mean and the three ``_mean`` gradients (the last conv layer fused with tf.reduce_mean(x, -1),
    models_gcn.py:673) against float64: mean of relu(y + bias) over the filters; dW, dstack, dbias for dy[b][o][m] =
    gmean[b][m] gated by the ReLU mask."""
    import ctypes
    from gcn_fmri_decoding_amd import _lib
    lib = _lib.lib()
    assert lib.chebgcn_contract_fwd_mean_supported(B, M, Fin, K, Fout) == 1
    assert lib.chebgcn_contract_fwd_mean_supported(2, 380, Fin, K, Fout) == 0         # a small launch: not served
    Mp = ops.plane_stride(M)
    gen = torch.Generator(device=dev)
    gen.manual_seed(B + M)
    stack = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    stack[..., M:] = float('nan')
    W = torch.randn((Fin * K, Fout), generator=gen, device=dev) * 0.1
    if bias_kind == 'vertex':
        kind = ops.BIAS_VERTEX
        bias = torch.zeros((Fout, Mp), device=dev)
        bias[:, :M] = torch.randn((Fout, M), generator=gen, device=dev) * 0.3
        bias_ref = bias[:, :M].double()
    else:
        kind = ops.BIAS_FILTER
        bias = torch.randn((Fout,), generator=gen, device=dev) * 0.3
        bias_ref = bias.double()[:, None]
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    mean = torch.full((B, Mp), float('nan'), device=dev)
    mask = torch.zeros((B, Fout, Mp // 4), dtype=torch.uint8, device=dev)
    _lib.check(lib.chebgcn_contract_fwd_mean(P(stack), P(W), P(bias), kind, P(mean), P(mask), B, M, Fin, K, Fout, st), 'fwd_mean')
    S = stack[..., :M].permute(2, 0, 1, 3).reshape(Fin * K, B, M).double()
    pre = torch.einsum('rbm,ro->bom', S, W.double()) + bias_ref
    ref = pre.clamp(min=0).mean(1)
    assert float((mean[:, :M].double() - ref).abs().max() / pre.abs().max()) <= REL
    bits = torch.stack([(mask >> r) & 1 for r in range(4)], -1).reshape(B, Fout, Mp)[..., :M].bool()
    assert torch.equal(bits, pre > 0) or float(((bits != (pre > 0)) & (pre.abs() > 1e-5 * pre.abs().max())).sum()) == 0

    gm = torch.zeros((B, Mp), device=dev)
    gm[:, :M] = torch.randn((B, M), generator=gen, device=dev)
    dy = (gm[:, None, :M] * bits).double()
    n = lib.chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    dW = torch.full((Fin * K, Fout), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_w_relu_mean(P(stack), P(gm), P(mask), P(dW), P(ws), n, B, M, Fin, K, Fout, st), 'bwd_w_mean')
    dW_ref = torch.einsum('rbm,bom->ro', S, dy)
    assert float((dW.double() - dW_ref).abs().max() / dW_ref.abs().max()) <= GREL
    del S
    gstack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_x_relu_mean(P(gm), P(mask), P(W), P(gstack), B, M, Fin, K, Fout, st), 'bwd_x_mean')
    gs_ref = torch.einsum('ro,bom->rbm', W.double(), dy).reshape(Fin, K, B, M).permute(1, 2, 0, 3)
    assert float((gstack[..., :M].double() - gs_ref).abs().max() / gs_ref.abs().max()) <= GREL
    nb = lib.chebgcn_brelu_pool_bwd_workspace(B, M, Fout, 1, kind)
    bws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    dbias = torch.full(tuple(bias.shape), float('nan'), device=dev)
    _lib.check(lib.chebgcn_bias_grad_relu_mean(P(gm), P(mask), P(dbias), kind, B, M, Fout, P(bws), nb, st), 'bias_mean')
    db_ref = dy.sum(0) if bias_kind == 'vertex' else dy.sum((0, 2))
    got = dbias[:, :M].double() if bias_kind == 'vertex' else dbias.double()
    assert float((got - db_ref).abs().max() / db_ref.abs().max()) <= GREL
