"""GPU tests added in round 3 (all through the C ABI): the fixed-order per-filter bias gradient of ``b1relu``, the
training step captured as one HIP graph against the eager step, and the full network at the shape the reference's
own ``training.py`` builds (atlas-sized graph, K = 10 x 6, batch 128) against the oracle.
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import layers_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops():
    from gcn_fmri_decoding_amd import ops as _ops
    return _ops


@pytest.mark.parametrize('M,F,B,pool,relu_mask', [(10466, 32, 64, 1, True), (232, 8, 5, 1, True), (232, 8, 5, 1, False),
                                                  (792, 16, 6, 4, False), (10466, 32, 3, 2, False)])
def test_b1relu_bias_gradient_is_a_fixed_order_sum(ops, dev, M, F, B, pool, relu_mask):
    """``chebgcn_brelu_pool_bwd`` with ``CHEBGCN_BIAS_FILTER`` (b1relu, models_gcn.py:619-623): per-workgroup partials
    plus one wave per filter instead of a float atomicAdd across workgroups.  Twenty launches give BIT-identical sums
    (the atomics differed from run to run), equal to a float64 sum of the same gated gradient to fp32 round-off; a
    missing workspace is refused."""
    from gcn_fmri_decoding_amd import _lib
    lib, P, st = _lib.lib(), ops._p, ops._stream()
    Mp, Mo = _lib.plane_stride(M), M // pool
    Mpo = _lib.plane_stride(Mo)
    gen = torch.Generator(device=dev)
    gen.manual_seed(M + F)
    gout = torch.randn((B, F, Mpo), generator=gen, device=dev)
    gout[:, :, Mo:] = 1e30                                  # plane pads must not leak into the sums
    nws = lib.chebgcn_brelu_pool_bwd_workspace(B, M, F, pool, ops.BIAS_FILTER)
    # (a per-vertex bias needs none at pool 1; a pooled layer's 16-byte-store kernel keeps per-batch-part partials since round 6)
    assert nws > 0 and (lib.chebgcn_brelu_pool_bwd_workspace(B, M, F, pool, ops.BIAS_VERTEX) == 0) == (pool == 1 or Mp < 2048)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    if relu_mask:                                           # pool == 1 layer: the ReLU bit mask contract_fwd leaves
        bits = torch.randint(0, 16, (B, F, Mp // 4), generator=gen, device=dev, dtype=torch.uint8)
        keep = torch.stack([(bits >> i) & 1 for i in range(4)], -1).reshape(B, F, Mp)[:, :, :M].bool()
        gated = torch.where(keep, gout[:, :, :M], torch.zeros((), device=dev))
        args = lambda db: (P(gout), None, P(bits), None, P(db), ops.BIAS_FILTER, B, M, F, 1, 0, 1, P(ws), nws, st)
    else:                                                   # forward output + argmax of a max-pooling layer
        out = torch.randn((B, F, Mpo), generator=gen, device=dev)
        arg = torch.randint(0, pool, (B, F, Mpo), generator=gen, device=dev, dtype=torch.uint8)
        gated = torch.where(out[:, :, :Mo] > 0, gout[:, :, :Mo], torch.zeros((), device=dev))
        dy = torch.empty((B, F, Mp), device=dev)
        args = lambda db: (P(gout), P(out), P(arg) if pool > 1 else None, P(dy), P(db), ops.BIAS_FILTER, B, M, F, pool, 0, 1,
                           P(ws), nws, st)
    ref = gated.double().sum((0, 2))
    results = []
    for _ in range(20):
        db = torch.full((F,), float('nan'), device=dev)
        _lib.check(lib.chebgcn_brelu_pool_bwd(*args(db)), 'brelu_pool_bwd')
        results.append(db.clone())
    for r in results[1:]:
        assert torch.equal(r, results[0])
    err = float((results[0].double() - ref).abs().max() / gated.double().abs().sum((0, 2)).max())
    assert err <= 1e-6, err
    db = torch.zeros((F,), device=dev)
    a = list(args(db))
    a[-3], a[-2] = None, 0
    assert lib.chebgcn_brelu_pool_bwd(*a) != 0 and b'workspace' in lib.chebgcn_last_error()


def _build(z, dev, **kw):
    from conftest import csr_from
    from gcn_fmri_decoding_amd import models_gcn
    Ls = [csr_from(z, 'L%d' % i) for i in range(int(z['nlevels']))]
    net = models_gcn.cgcnn({'device': dev}, Ls, z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist(),
                           channel=int(z['channel']), brelu=str(z['brelu']), batch_size=int(z['x'].shape[0]), verbose=False, **kw)
    for k in z.files:
        if k.startswith('param:'):
            net.set_variable(k[len('param:'):], z[k])
    return net


@pytest.mark.parametrize('name', ['inference_flat_n212', 'inference_pool_n212', 'inference_pool6_n512'])
def test_captured_step_equals_eager_step(ops, dev, name):
    """``cgcnn.enable_step_graph``: the training step captured as ONE HIP graph (forward, loss, backward with the second
    stream of contract_bwd_w, Adam reading lr_t from device memory, EMA bookkeeping) replays the same kernels on the
    same operands: after six steps (two eager warm-up steps, capture at the third, three replays) on changing batches
    the variables, the Adam moments and the reported loss_average series are BIT-identical to six eager steps."""
    z = load_golden(name)
    B, M0, C = z['x'].shape
    rs = np.random.RandomState(3)
    batches = [rs.randn(B, M0, C).astype(np.float32) for _ in range(6)]
    labels = [rs.randint(0, int(z['M'][-1]), B) for _ in range(6)]
    runs = []
    for graphed in (False, True):
        net = _build(z, dev, regularization=5e-4, dropout=1)
        net.enable_step_graph(graphed)
        series = []
        for xb, lb in zip(batches, labels):
            x = ops.plane_storage(torch.as_tensor(xb).to(dev)).contiguous()
            lr, la = net.train_step(x, torch.as_tensor(lb).to(dev))
            series.append(la)
        torch.cuda.synchronize()
        assert net.global_step == 6 and (net._sg is not None) == graphed
        runs.append((net._flat.clone(), net._adam_m.clone(), net._adam_v.clone(), [float(v) for v in series]))
    for a, b in zip(runs[0][:3], runs[1][:3]):
        assert torch.equal(a, b)
    assert runs[0][3] == runs[1][3]


def test_captured_step_with_dropout_trains(ops, dev):
    """With dropout (keep 0.5, as the reference trains) the captured step draws a fresh mask per replay (graph-safe
    Philox offsets): losses stay finite and differ between replays of the same batch."""
    z = load_golden('inference_flat_n212')
    net = _build(z, dev, regularization=5e-4, dropout=0.5)
    net.enable_step_graph(True)
    x = ops.plane_storage(torch.as_tensor(z['x']).to(dev)).contiguous()
    lb = torch.as_tensor(np.arange(z['x'].shape[0]) % int(z['M'][-1])).to(dev)
    before = net._flat.clone()
    torch.manual_seed(0)
    for _ in range(6):
        net.train_step(x, lb)
    # two replays of the forward alone on the same variables differ only through the dropout masks
    vals = [float(net.train_step(x, lb)[1]) for _ in range(3)]
    assert all(np.isfinite(v) for v in vals) and not torch.equal(before, net._flat)


@pytest.mark.parametrize('n_nodes', [360, 400, 1000])
def test_reference_training_shape_vs_oracle(ops, dev, n_nodes):
    """The network the reference's ``training.py`` actually builds (model.py:271-280, training.py:34,
    configure_fmri.py:28, 41): atlas-sized graph (360 = MMP atlas; 1000), kNN-8, one coarsening level, ChebNet
    K = 10 x 6, F = 32, p = 1, b2relu, FC 512-256-22, batch 128, block_dura 15 -- logits, loss, every gradient and
    one TF-form Adam step against the oracle, on the kernels this shape selects (fused atlas layer / generic four-plane recurrence /
    the 256-thread ordered recurrence on the relabelled graph; batch split of the bias gradient), eagerly and through the
    captured HIP graph."""
    from gcn_fmri_decoding_amd import graph, models_gcn
    from conftest import assert_adam_params_close
    Ls, perm, _ = graph.synthetic_graph(n_nodes, k=8, levels=1)
    L = Ls[0]
    M, B, C = L.shape[0], 128, 15
    F, K, p, Mfc = [32] * 6, [10] * 6, [1] * 6, [512, 256, 22]
    reg = 5e-4
    torch.manual_seed(0)
    net = models_gcn.cgcnn({'device': dev}, [L] * 6, F, K, p, Mfc, filter='chebyshev5', brelu='b2relu', pool='mpool1', initial='he',
                           channel=C, regularization=reg, dropout=1, batch_size=B, learning_rate=0.001, decay_rate=0.9,
                           momentum=0.9, verbose=False)
    params = {k: net.get_var(k).copy() for k in net.variables()}
    onet = R.Net([L] * 6, F, K, p, Mfc, channel=C, brelu='b2relu', regularization=reg)
    rs = np.random.RandomState(1)
    x = np.zeros((B, M, C), np.float32)
    keep = np.asarray(perm) < n_nodes
    x[:, keep, :] = rs.randn(B, int(keep.sum()), C).astype(np.float32)      # fake vertices carry zeros (perm_data_3d)
    labels = rs.randint(0, 21, B)
    logits, cache = onet.forward(params, x)
    loss, dlogits = onet.loss(params, logits, labels)
    grads = onet.backward(params, cache, dlogits)
    # the same network in float64: six K = 10 layers deep, two fp32 evaluations of a gradient differ by ~1e-4 of its
    # scale (summation order; a ReLU within round-off of zero falls on either side and moves a per-vertex bias
    # gradient by a whole term) -- so gradients are compared with the float64 result.  A weight gradient is a plain fp32
    # sum over batch x vertices = 128 x M ~ 1e5 products of either sign: sqrt(n) * 2^-24 * (sum |terms| / |sum|) ~ 1e-4
    # of the gradient's scale is what fp32 accumulation gives (measured: 5e-5 at the 99.9 % quantile for N = 1000; NumPy's
    # blocked BLAS sums reach 7e-6 there; the last layer's weights, whose terms cancel most, 2e-4 on the GPU and 1.6e-4 in a
    # NumPy fp32 run with other variables); bound 5e-4, or three times the fp32 oracle's own error where that is larger
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    onet64 = R.Net([L.astype(np.float64)] * 6, F, K, p, Mfc, channel=C, brelu='b2relu', regularization=reg)
    logits64, cache64 = onet64.forward(p64, x.astype(np.float64))
    loss64, dlogits64 = onet64.loss(p64, logits64, labels)
    grads64 = onet64.backward(p64, cache64, dlogits64)
    assert logits64.dtype == np.float64
    xs = ops.plane_storage(torch.as_tensor(x).to(dev)).contiguous()
    with torch.no_grad():
        got = net._inference_storage(xs, 1).cpu().numpy()
    assert np.abs(got - logits).max() <= 2e-5 * np.abs(logits).max()
    assert np.abs(got - logits64).max() <= 1e-5 * np.abs(logits64).max()       # the north star's 1e-5, against float64
    ld = torch.as_tensor(labels).to(dev)
    _, loss_avg = net.train_step(xs, ld)
    assert abs(float(loss_avg) - 0.1 * loss) <= 2e-5 * abs(0.1 * loss)      # (first read of the zero-initialised 0.9-EMA)
    measured = {}
    for k in params:
        spec = next(s for s in net._spec_list if s.name == k)
        g = net.gradient(k)
        l2 = reg * p64[k] if onet.regularized(k) else 0
        ref64 = grads64[k] - l2
        scale = max(np.abs(ref64).max(), 1e-30)
        e_gpu = np.abs(g.cpu().numpy().astype(np.float64) - ref64) / scale
        e_o32 = np.abs(grads[k].astype(np.float64) - l2 - ref64) / scale
        # per-vertex bias gradients are sums over the 128 windows only: there a ReLU that falls on the other side of zero
        # (its pre-activation within round-off of 0) moves single elements by a whole term -- 0.1 % of the elements of the
        # deeper layers in either fp32 evaluation -- so they are compared at the 99 % quantile, weights at 99.9 %
        qq = 0.99 if spec.group == 'convb' else 0.999
        q_gpu, q_o32 = np.quantile(e_gpu, qq), np.quantile(e_o32, qq)
        measured[k] = [float(q_gpu), float(e_gpu.max()), float(q_o32), float(e_o32.max()), spec.group]
    from conftest import record_measured
    record_measured('reference_training_shape_vs_oracle[%d]' % n_nodes, what='[quantile, max] of the GPU, of the fp32 oracle; of scale',
                    **measured)
    for k, (q_gpu, m_gpu, q_o32, m_o32, group) in measured.items():
        # (round 6, tools/probes/n1000_grad_noise.py: the same variables in the caller's and in the length order, three seeds each --
        # an instance without a ReLU flip comes out at 1e-7 for EVERY conv gradient in either order, one with flips at 5e-5 ...
        # 4.5e-4 at these quantiles in either order (a flip in layer l moves dy of every vertex within K - 1 = 9 hops in the layers
        # below: most of a 1000-vertex graph).  The instance of this test has its flip; the bound is the weights' for both groups)
        assert q_gpu <= max(5e-4, 3 * q_o32), \
            'grad %s: quantile %.3e (fp32 oracle %.3e)' % (k, q_gpu, q_o32)
        if group != 'convb':          # a weight gradient sums over every vertex and window: single flips average out
            assert m_gpu <= max(2e-3, 3 * m_o32), 'grad %s: max %.3e (fp32 oracle %.3e)' % (k, m_gpu, m_o32)
    state, ill = {}, {}
    R.adam_tf_step(params, grads, state)
    for k in params:
        # (lr = 2e-3 in the bound: where a ReLU flipped, the two gradients of a bias element can have opposite signs and the
        # first Adam update, -lr * sign(g), differs by two learning rates)
        assert_adam_params_close(net.get_var(k), params[k], state['v/' + k], 0, ill, k, rel=2e-5, lr=2e-3, quantile=0.999)
    # which kernels a step of this shape runs: up to 384 vertices every conv layer and its input gradient are ONE on-chip
    # launch each (csrc/fused_small.hip; 12 waves at M = 376, a window split between two workgroups at batch 128)
    ops.timers = ops.KernelTimers(by_dispatch=True)
    net.train_step(xs, ld)
    names = sorted(ops.timers.summary())
    ops.timers = None
    if ops.plane_stride(M) <= 384:
        nw = 12
        for want in ('fused_layer_fwd | fused_layer_kernel<%d,8,false> + fused_combine_kernel' % nw,
                     'fused_layer_bwd_x | fused_layer_kernel<%d,8,true>' % nw):
            assert want in names, (want, names)
        assert not any(n.startswith('recurrence') for n in names), names
    elif ops.plane_stride(M) <= 1024:
        assert not any('fused_layer' in n for n in names) and any(n.startswith('recurrence_fwd | cheb_onchip_kernel<4,') for n in names), names
    else:
        # planes of more than 1024 vertices (N = 1000: M = 1044): cgcnn relabels the level by row length and the forward recurrence --
        # on the activations and, for the input gradient, on dy -- is the 256-thread ordered kernel (round 6)
        assert net.vertex_order == 'length' and net.graphs[0].ordered
        for want in ('recurrence_fwd | cheb_ord_kernel<1040,2,1,256,false>', 'recurrence_fwd_t | cheb_ord_kernel<1040,2,1,256,false>'):
            assert want in names, (want, names)
        assert not any('fused_layer' in n or n.startswith('recurrence_bwd') for n in names), names
    # the same model through the captured graph: two more eager steps, capture, replay -- bit-identical to a twin that
    # runs all of them eagerly
    twin = models_gcn.cgcnn({'device': dev}, [L] * 6, F, K, p, Mfc, filter='chebyshev5', brelu='b2relu', pool='mpool1', initial='he',
                            channel=C, regularization=reg, dropout=1, batch_size=B, learning_rate=0.001, decay_rate=0.9,
                            momentum=0.9, verbose=False)
    twin.load_state_dict(net.state_dict())
    twin._loss_ema = net._loss_ema.clone()
    net.enable_step_graph(True)
    for _ in range(4):
        net.train_step(xs, ld)
        twin.train_step(xs, ld)
    torch.cuda.synchronize()
    assert net._sg is not None and torch.equal(net._flat, twin._flat)


@pytest.mark.gpu
def test_bench_prints_one_json_line_on_stdout():
    """``python bench.py`` (kernel legs on: the dp_overhead leg brings up an RCCL communicator, which prints a version
    banner to file descriptor 1): stdout must carry the result line and nothing else."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '3', '--warmup', '3', '--repeats', '1',
                        '--instrumented-steps', '2', '--cpu-windows', '0'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 1 and line['n_ranks_seen'] == 1 and line['steps'] == 3
    for key in ('roofline', 'step_roofline', 'kernels', 'northstar', 'config4', 'config5', 'dp_overhead', 'refshape', 'fit'):        # (cpu_baseline: --cpu-windows 0 here)
        assert key in line, key
    assert line['roofline']['bound'] == 'hbm' and 0.2 < line['roofline']['frac'] < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize('B,I,O,ld', [(128, 512, 256, 512), (128, 360, 512, 384), (64, 256, 22, 256), (50, 1000, 512, 1024),
                                      (3, 8, 1, 8), (128, 4096, 64, 4096), (33, 36, 33, 40), (8, 30, 4, 32), (5, 1, 3, 4),
                                      (64, 10466, 512, 10496), (16, 2051, 40, 2052)])
@pytest.mark.parametrize('relu', [False, True])
def test_fc_forward_small_products(B, I, O, ld, relu):
    """Head FC layer (reference models_gcn.py:650-656) by the library kernel against float64; x as a view of a wider
    buffer, ragged batch / output sizes, inner sizes that are not a multiple of the wave split."""
    from gcn_fmri_decoding_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(B * 7 + I + O)
    buf = torch.randn(B, ld, generator=g).to(dev)
    buf[:, I:] = float('nan')                      # what lies beyond the logical row never reaches the result
    x = buf[:, :I]
    W = (torch.randn(I, O, generator=g) * 0.1).to(dev)
    b = torch.randn(O, generator=g).to(dev)
    y = ops.fc_forward(x, W, b, relu)
    assert y is not None and y.shape == (B, O)
    ref = x.double() @ W.double() + b.double()
    if relu:
        ref = ref.clamp_min(0)
    scale = (x.double().abs() @ W.double().abs()).max().item()
    assert (y.double() - ref).abs().max().item() <= 2e-6 * scale          # fp32 accumulation over I terms
    y2 = ops.fc_forward(x, W, b, relu)
    assert torch.equal(y, y2)                                             # fixed summation order
    assert ops.fc_forward(x, W, None, relu) is not None


@pytest.mark.gpu
def test_fc_forward_declines_what_it_does_not_cover():
    from gcn_fmri_decoding_amd import ops
    dev = torch.device('cuda:0')
    assert ops.fc_forward(torch.zeros(8, 30, device=dev), torch.zeros(30, 4, device=dev), None, True) is None       # row stride
    assert ops.fc_forward(torch.zeros(8, 33, device=dev)[:, 1:], torch.zeros(32, 4, device=dev), None, True) is None  # alignment
    assert ops.fc_forward(torch.zeros(2 ** 11, 8, device=dev), torch.zeros(8, 2 ** 10, device=dev), None, True) is None


@pytest.mark.gpu
@pytest.mark.parametrize('B,I,O,ld', [(128, 512, 256, 512), (128, 360, 512, 384), (64, 256, 22, 256), (50, 1000, 512, 1024),
                                      (3, 8, 1, 8), (33, 36, 33, 40)])
@pytest.mark.parametrize('relu', [False, True])
def test_fc_backward_small_products(B, I, O, ld, relu):
    """Gradients of the head FC layer (TF autodiff of reference models_gcn.py:650-656, ReluGrad included) by the library
    kernels against float64."""
    from gcn_fmri_decoding_amd import ops
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(B * 11 + I + O)
    buf = torch.randn(B, ld, generator=gen).to(dev)
    x = buf[:, :I]
    W = (torch.randn(I, O, generator=gen) * 0.1).to(dev)
    b = torch.randn(O, generator=gen).to(dev)
    g = torch.randn(B, O, generator=gen).to(dev)
    y = ops.fc_forward(x, W, b, relu) if relu else None
    dW = torch.full((I, O), float('nan'), device=dev)
    db = torch.full((O,), float('nan'), device=dev)
    (dx,) = ops.fc_backward(x, W, g, y, dW, db, True)
    assert dx.shape == (B, I)
    gm = g.double() * (y > 0).double() if relu else g.double()
    for got, ref, scale in ((dW, x.double().t() @ gm, (x.double().abs().t() @ gm.abs()).max().item()),
                            (db, gm.sum(0), gm.abs().sum(0).max().item()),
                            (dx, gm @ W.double().t(), (gm.abs() @ W.double().abs().t()).max().item())):
        assert (got.double() - ref).abs().max().item() <= 2e-6 * max(scale, 1e-30)
    dW2, db2 = torch.empty_like(dW), torch.empty_like(db)
    (dx2,) = ops.fc_backward(x, W, g, y, dW2, db2, True)
    assert torch.equal(dW, dW2) and torch.equal(db, db2) and torch.equal(dx, dx2)      # fixed summation order
    assert ops.fc_backward(x, W, g, y, dW2, db2, False) == (None,) and torch.equal(dW, dW2)
    big = torch.zeros(4, 10466, device=dev)
    assert ops.fc_backward(big, torch.zeros(10466, 8, device=dev), torch.zeros(4, 8, device=dev), None,
                           torch.zeros(10466, 8, device=dev), torch.zeros(8, device=dev), True) is None
