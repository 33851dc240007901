#!/usr/bin/env python3
"""Throughput of the Chebyshev graph-CNN training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward + loss + backward + (RCCL gradient all-reduce) + TF-form Adam on one batch
of synthetic fMRI windows, inputs resident in HBM.  Workload = BASELINE.json configs[1]:
ChebNet K=5, 6 conv layers (F=32, no pooling, per-vertex biases), FC 512-256-22, the seeded
synthetic N=10000 kNN graph (M=10466 after one coarsening level), block_dura=15, batch 64
per GPU (weak scaling; configs[2] is the same at 8 GPUs).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = 'fMRI-windows/sec (fwd+bwd) 6-layer ChebNet K=5 N≈10k; HBM GB/s vs roofline'
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TFLOPS = 157.3


def load_graph(n_nodes, levels, rank, world, barrier):
    """Seeded synthetic brain graph; built once per node, cached in /tmp."""
    import scipy.sparse as sp
    path = '/tmp/chebgcn_bench_graph_n%d_l%d.npz' % (n_nodes, levels)
    if rank == 0 and not os.path.exists(path):
        from gcn_fmri_decoding_amd import graph
        Ls, perm, _ = graph.synthetic_graph(n_nodes, k=8, levels=levels)
        if perm is None:                    # levels = 0: no coarsening, no fake vertices
            perm = np.arange(Ls[0].shape[0])
        fields = {'perm': np.asarray(perm, np.int32), 'nl': np.int64(len(Ls))}
        for i, L in enumerate(Ls):
            L = sp.csr_matrix(L)
            fields.update({'p%d' % i: L.indptr, 'i%d' % i: L.indices, 'd%d' % i: L.data, 's%d' % i: np.array(L.shape)})
        np.savez(path + '.tmp.npz', **fields)
        os.replace(path + '.tmp.npz', path)
    if world > 1:
        barrier()
    z = np.load(path)
    Ls = [sp.csr_matrix((z['d%d' % i], z['i%d' % i], z['p%d' % i]), shape=tuple(z['s%d' % i])) for i in range(int(z['nl']))]
    return Ls, z['perm']


def cpu_baseline(Ls, cfg, n_windows, seed=0, repeats=5):
    """CPU baselines on the GPU box's host, same workload, ``n_windows`` windows per step
    (BASELINE.md section 2):

    * B2 (the headline ``value``): oracle/torch_cpu_ref.py -- torch CPU, ``torch.sparse_csr``
      SpMM + matmul + autograd + TF-form Adam on every host core torch uses; one warm-up step,
      then the median of ``repeats`` steps;
    * ``scipy``: oracle/layers_ref.py -- NumPy/SciPy with the hand-written backward (SciPy's
      SpMM runs on one thread, BLAS on many), one warm-up and one timed step.
    Reported baselines, not the optimisation target."""
    import torch
    from oracle import layers_ref as R
    from oracle import torch_cpu_ref as TR
    try:
        import threadpoolctl
        blas_threads = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    net = R.Net(Ls, cfg['F'], cfg['K'], cfg['p'], cfg['M'], channel=cfg['channel'], brelu='b2relu', regularization=5e-4)
    rs = np.random.RandomState(seed)
    params = {}
    for k, s in net.param_shapes().items():
        params[k] = (np.full(s, 0.2, np.float32) if k.endswith('bias')
                     else (rs.randn(*s) * np.sqrt(2.0 / s[0])).astype(np.float32))
    x = rs.randn(n_windows, Ls[0].shape[0], cfg['channel']).astype(np.float32)
    labels = rs.randint(0, 21, n_windows)

    tnet = TR.TorchNet([Ls[0]] * len(cfg['F']), cfg['F'], cfg['K'], cfg['p'], cfg['M'], channel=cfg['channel'], brelu='b2relu',
                       regularization=5e-4)
    tparams = {k: torch.tensor(v) for k, v in params.items()}
    tx, ty, tstate = torch.tensor(x), torch.tensor(labels), {}
    t_all = time.time()
    tnet.train_step(tparams, tx, ty, tstate)                    # warm-up (first-touch page faults, thread pools)
    times = []
    for _ in range(repeats):
        t0 = time.time()
        tnet.train_step(tparams, tx, ty, tstate)
        times.append(time.time() - t0)
    med = float(np.median(times))
    t_torch = time.time() - t_all

    def scipy_step():
        logits, cache = net.forward(params, x)
        loss, dlogits = net.loss(params, logits, labels)
        grads = net.backward(params, cache, dlogits)
        R.adam_tf_step(params, grads, {})
    t_all = time.time()
    scipy_step()
    t0 = time.time()
    scipy_step()
    dt_scipy = time.time() - t0
    t_scipy = time.time() - t_all
    # the REFERENCE's own NumPy/SciPy path (lib_new/graph.py:155-172 + models_gcn.py:611-616), timed by oracle/time_reference.py
    # in the build container -- the reference never travels to this box: static, labelled
    reference = None
    rpath = os.path.join(ROOT, 'profiles', 'r06_reference_cpu.json')
    if os.path.exists(rpath):
        rj = json.load(open(rpath))
        reference = {'static': True, 'source': 'profiles/r06_reference_cpu.json (oracle/time_reference.py, build container, not this host)',
                     'what': rj['what'], 'host': rj['host'],
                     'configs1_six_conv_layers_forward': rj['configs1_network_forward'],
                     'layer_s': {k: rj[k]['layer_s'] for k in ('configs1_layer1', 'configs1_layers2to6', 'configs3', 'configs4')}}
    return {'value': n_windows / med, 'unit': 'windows/s', 'cores': int(torch.get_num_threads()), 'kind': 'port', 'reference': reference,
            'sample': '%d windows per step, full 6-layer fwd+loss+bwd+Adam; torch-CPU restatement (oracle/torch_cpu_ref.py, '
                      'torch.sparse_csr SpMM, %d torch threads of %d host CPUs): 1 warm-up + median of %d steps, %.1f s in all'
                      % (n_windows, torch.get_num_threads(), os.cpu_count() or 0, repeats, t_torch),
            'step_s_median': med, 'step_s_all': [float(t) for t in times],
            'scipy': {'value': n_windows / dt_scipy, 'unit': 'windows/s', 'cores': 1, 'blas_threads': int(blas_threads),
                      'sample': '%d windows, NumPy/SciPy oracle (oracle/layers_ref.py; SciPy SpMM is single-threaded, BLAS uses '
                                '%d threads): 1 warm-up + 1 timed step, %.1f s in all' % (n_windows, blas_threads, t_scipy)}}


def kernel_leg(lib_graph, B, Fin, K, launches, what):
    """HIP-event timing of the recurrence kernels alone at a shape BASELINE.json names (the
    north-star shape K=5 / Fin=32 / batch 256, configs[3] K=25 / Fin=Fout=64): ``launches``
    back-to-back launches each, events recorded on the launch stream, algorithmic bytes per
    SURVEY.md 8(d): forward 4*M*Fin*K per window (T_0 in place, as the model runs it; the copy of
    x adds one slab), adjoint 4*M*Fin*(K+1)."""
    import torch
    from gcn_fmri_decoding_amd import _lib, ops
    g = lib_graph
    lib = _lib.lib()
    dev = g.device
    stack = torch.randn((K, B, Fin, g.Mp), device=dev)
    gstack = torch.randn((K, B, Fin, g.Mp), device=dev)
    dx = torch.empty((B, Fin, g.Mp), device=dev)
    xcopy = stack[0].clone()
    st, P = ops._stream(), ops._p
    calls = {
        'recurrence_fwd': (lambda: lib.chebgcn_recurrence_fwd(g.handle, P(stack), P(stack), B, Fin, K, st), 4.0 * g.M * Fin * K * B),
        'recurrence_fwd_copy_x': (lambda: lib.chebgcn_recurrence_fwd(g.handle, P(xcopy), P(stack), B, Fin, K, st),
                                  4.0 * g.M * Fin * (K + 1) * B),
        'recurrence_bwd': (lambda: lib.chebgcn_recurrence_bwd(g.handle, P(gstack), P(dx), B, Fin, K, st), 4.0 * g.M * Fin * (K + 1) * B),
    }
    out = {'shape': {'B': B, 'Fin': Fin, 'K': K, 'M': g.M}, 'launches': launches, 'what': what,
           'timing': 'HIP events around each launch on the launch stream, mean over the launches after 3 warm-up launches'}
    for name, (fn, nbytes) in calls.items():
        for _ in range(3):
            _lib.check(fn(), name)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for s, e in evs:
            s.record()
            _lib.check(fn(), name)
            e.record()
        torch.cuda.synchronize()
        ms = [s.elapsed_time(e) for s, e in evs]
        avg = float(np.mean(ms))
        out[name] = {'avg_ms': avg, 'min_ms': float(np.min(ms)), 'algorithmic_bytes': nbytes, 'GBps': nbytes / avg / 1e6,
                     'frac': nbytes / avg / 1e6 / HBM_PEAK_GBS, 'kernel': _lib.last_dispatch()}
    del stack, gstack, dx, xcopy
    torch.cuda.empty_cache()
    return out


def layer_leg(lib_graph, B, Fin, K, Fout, steps, what, precisions=('f32', 'bf16', 'bf16x3')):
    """One wide layer (per-vertex bias + ReLU) trained forward + backward -- recurrence, contraction, bias/ReLU
    gradient, contraction gradients (dW, d stack), adjoint recurrence -- with the contraction and its two
    gradients in fp32 MFMA, bf16 and split-bf16 (``ops.cheb_conv(precision=...)``): BASELINE configs[4]
    (block_dura = Fin = 60 -> Fout = 256, K = 5) and configs[3] (K = 25, 64 -> 64).  Reports the layer step
    time, the per-kernel HIP-event times and the fp32 parity of the mixed-precision results (max error relative
    to max|fp32 result| of y, dx and dW on the same layer without the ReLU); ``default_precision`` = what
    precision 'auto' (cgcnn's default) resolves to for this shape."""
    import torch
    from gcn_fmri_decoding_amd import ops
    g = lib_graph
    dev = g.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    x = torch.randn((B, Fin, g.Mp), generator=gen, device=dev)
    x[:, :, g.M:] = 0
    W = (torch.randn((Fin * K, Fout), generator=gen, device=dev) * (2.0 / (Fin * K)) ** 0.5).requires_grad_(True)
    bias = (torch.randn((Fout, g.Mp), generator=gen, device=dev) * 0.1).requires_grad_(True)
    gout = torch.randn((B, Fout, g.Mp), generator=gen, device=dev)
    gout[:, :, g.M:] = 0
    out = {'shape': {'B': B, 'Fin': Fin, 'K': K, 'Fout': Fout, 'M': g.M}, 'steps': steps, 'what': what,
           'default_precision': ops.resolve_precision('auto', Fin, K, Fout),
           'timing': 'wall clock over the steps between synchronisations; per-kernel HIP events on a separate instrumented pass'}
    ref = None
    for precision in precisions:
        def layer_step():
            xs = x.detach().requires_grad_(True)
            W.grad = bias.grad = None
            y = ops.cheb_conv(xs, W, bias, g, K, relu=True, bias_kind=ops.BIAS_VERTEX, precision=precision)
            y.backward(gout)
            return y, xs.grad
        for _ in range(2):
            layer_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y, dx = layer_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ops.timers = ops.KernelTimers()
        for _ in range(3):
            layer_step()
        kern = ops.timers.summary()
        ops.timers = None
        # parity pass WITHOUT the ReLU: with it, pre-activations within rounding of zero flip their mask and a
        # flipped element changes dx / dW by a whole gout -- that would measure the flips, not the arithmetic
        xs = x.detach().requires_grad_(True)
        W.grad = bias.grad = None
        y = ops.cheb_conv(xs, W, bias, g, K, relu=False, bias_kind=ops.BIAS_VERTEX, precision=precision)
        y.backward(gout)
        res = {'y': y[:, :, :g.M].detach(), 'dx': xs.grad[:, :, :g.M], 'dW': W.grad.clone()}
        leg = {'ms_per_step': 1e3 * dt, 'windows_per_s': B / dt,
               'kernels': {k: {'avg_ms': v['avg_ms'], 'GBps': v['bytes'] / (v['total_ms'] * 1e-3) / 1e9,
                               'TFLOPs': v['flops'] / (v['total_ms'] * 1e-3) / 1e12} for k, v in kern.items()}}
        if ref is None:
            ref = res
        else:
            leg['rel_err_vs_f32'] = {k: float((res[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in res}
        out[precision] = leg
        del y, dx, res
    del x, gout
    torch.cuda.empty_cache()
    return out

def refshape_leg(dev, n_nodes, steps, warmup, batch=128, korder=10, block_dura=15):
    """The shape the reference's own ``training.py`` builds (SURVEY.md 8 / VERDICT r2 item 7): an atlas-sized graph
    (MMP atlas: 360 regions, ``configure_fmri.py:11``; 1000 = the largest atlas the reference names), kNN-8,
    ``coarsening_levels = 1`` (``configure_fmri.py:41``, ``model.py:136-141``), ChebNet ``K = 10`` x 6 (``training.py:34``,
    ``model.py:271-280``), F = 32, p = 1, b2relu, FC 512-256-22, batch 128 (``configure_fmri.py:28``), block_dura 15.
    At these sizes a step is a chain of short kernels: timed eagerly and as ONE captured HIP graph
    (``cgcnn.enable_step_graph``)."""
    import torch
    from gcn_fmri_decoding_amd import graph, models_gcn, ops
    Ls, perm, _ = graph.synthetic_graph(n_nodes, k=8, levels=1)
    cfg = dict(F=[32] * 6, K=[korder] * 6, p=[1] * 6, M=[512, 256, 22])
    out = {'shape': {'N': n_nodes, 'M': int(Ls[0].shape[0]), 'K': korder, 'F': 32, 'layers': 6, 'batch': batch,
                     'block_dura': block_dura}, 'steps': steps, 'warmup': warmup,
           'what': "the reference's default training shape (training.py / model.py:271-280 / configure_fmri.py), full step "
                   'fwd+loss+bwd+Adam, batch gathered on device'}
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    S = 4 * batch
    data = torch.randn((S, n_nodes, block_dura), generator=gen, device=dev)
    labels = torch.randint(0, 21, (S,), generator=gen, device=dev)
    order = torch.stack([torch.randperm(S, generator=gen, device=dev)[:batch].to(torch.int32) for _ in range(steps + warmup)])
    for mode in ('eager', 'hip_graph'):
        torch.manual_seed(0)
        net = models_gcn.cgcnn({'device': dev}, [Ls[0]] * 6, cfg['F'], cfg['K'], cfg['p'], cfg['M'], filter='chebyshev5',
                               brelu='b2relu', pool='mpool1', initial='he', channel=block_dura, regularization=5e-4,
                               dropout=0.5, batch_size=batch, learning_rate=0.001, decay_rate=0.9, momentum=0.9, verbose=False)
        if mode == 'hip_graph':
            net.enable_step_graph(True)
        xbuf = ops.plane_empty(batch, block_dura, int(Ls[0].shape[0]), dev)
        perm_dev = net.compose_perm(perm)

        lab_order = labels[order.long()]      # (the labels of every step's batch, gathered once like fit()'s per-epoch label pool)

        def step(i):
            idx = order[i]
            xb = net.step_inputs()[0]          # the captured step's input buffer, once there is one: no copy into it
            x = net.as_internal(ops.perm_data(data, perm_dev, idx, out=xb if xb is not None else xbuf))
            return net.train_step(x, lab_order[i])
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            _, loss = step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[mode] = {'ms_per_step': 1e3 * dt, 'windows_per_s': batch / dt, 'final_loss': float(loss)}
        del net
    torch.cuda.empty_cache()
    return out


def pool6_leg(dev, steps, warmup, batch=64, block_dura=15, instrumented=5):
    """the pooling ChebNet of the legacy monolith (HCP_task_fmri_gcn_test8.py:1632-1635) at full size -- the
    N = 10000 graph coarsened six times (M = 12672 / 3168 / 792 at the conv layers), F = [32, 32, 64, 64, 128, 128],
    p = [1, 4, 1, 4, 1, 4], K = [20, 10, 10, 10, 5, 5], per-vertex biases, max pooling, FC 512-256-22 -- as a training
    step (fwd + loss + bwd + Adam, batch gathered and permuted on device): windows/s, the per-kernel table of a separate
    instrumented pass with the kernel template each entry dispatched, and the step's algorithmic bytes against 8 TB/s.
    Vertices stay in the coarsening's tree order (pooling is a max over adjacent vertices there)."""
    import torch
    from gcn_fmri_decoding_amd import _lib, models_gcn, ops
    Ls, perm = load_graph(10000, 6, 0, 1, None)
    F, K, p, Mfc = [32, 32, 64, 64, 128, 128], [20, 10, 10, 10, 5, 5], [1, 4, 1, 4, 1, 4], [512, 256, 22]
    torch.manual_seed(0)
    net = models_gcn.cgcnn({'device': dev}, Ls, F, K, p, Mfc, filter='chebyshev5', brelu='b2relu', pool='mpool1', initial='he',
                           channel=block_dura, regularization=5e-4, dropout=0.5, batch_size=batch, learning_rate=0.001,
                           decay_rate=0.9, momentum=0.9, verbose=False)
    out = {'shape': {'N': 10000, 'M': [int(g.M) for g in net.graphs], 'F': F, 'K': K, 'p': p, 'fc': Mfc, 'batch': batch,
                     'block_dura': block_dura}, 'steps': steps, 'warmup': warmup,
           'precision': [ops.resolve_precision(net.contraction, fi, k, fo) for fi, k, fo in zip([block_dura] + F[:-1], K, F)],
           'what': 'SURVEY 8(f)4: the monolith\'s pooling ChebNet at full size, full training step, inputs resident in HBM'}
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    S = 2 * batch
    data = torch.randn((S, 10000, block_dura), generator=gen, device=dev)
    labels = torch.randint(0, 21, (S,), generator=gen, device=dev)
    order = torch.stack([torch.randperm(S, generator=gen, device=dev)[:batch].to(torch.int32)
                         for _ in range(steps + warmup + instrumented)])
    perm_dev = net.compose_perm(perm)

    def step(i):
        idx = order[i]
        x = net.as_internal(ops.perm_data(data, perm_dev, idx))
        return net.train_step(x, labels[idx.long()])
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        _, loss = step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out.update({'ms_per_step': 1e3 * dt, 'windows_per_s': batch / dt, 'final_loss': float(loss)})
    if instrumented:
        overlap, ops.overlap_bwd_w = ops.overlap_bwd_w, False
        ops.timers = ops.KernelTimers(by_dispatch=True)
        for i in range(warmup + steps, warmup + steps + instrumented):
            step(i)
        kern = ops.timers.summary()
        ops.timers = None
        ops.overlap_bwd_w = overlap
        out['kernels'] = {k: {'ms_per_step': v['total_ms'] / instrumented, 'launches_per_step': v['launches'] / instrumented,
                              'GBps': v['bytes'] / (v['total_ms'] * 1e-3) / 1e9,
                              'frac_hbm': v['bytes'] / (v['total_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              'TFLOPs': v['flops'] / (v['total_ms'] * 1e-3) / 1e12} for k, v in kern.items()}
        step_bytes = sum(v['bytes'] for v in kern.values()) / instrumented
        out['step_roofline'] = {'algorithmic_bytes_per_step': step_bytes, 'frac': step_bytes / dt / 1e9 / HBM_PEAK_GBS}
        out['kernel_ms_sum'] = sum(v['total_ms'] for v in kern.values()) / instrumented
    del net, data
    torch.cuda.empty_cache()
    return out


DP_STEPS = 40


def fit_leg(dev, L, cfg, batch, n_train, n_val, epochs, tag):
    """The caller's own number: ``t_step`` exactly as the reference defines it (models_gcn.py:183-184: wall time of
    ``fit`` / number of steps, evaluation passes and checkpoints included), once with the training set already staged
    in HBM (``model.stage``) and once with the float64 NumPy array ``coarsening.perm_data_3d`` returns
    (coarsening.py:255) -- that run pays the host cast and the copy over PCIe inside ``fit``."""
    import contextlib
    import io
    import tempfile
    import torch
    from gcn_fmri_decoding_amd import models_gcn
    M = int(L.shape[0])
    rs = np.random.RandomState(5)
    train = rs.randn(n_train, M, cfg['channel'])                   # float64, like perm_data_3d's output
    tl = rs.randint(0, 21, n_train)
    val, vl = rs.randn(n_val, M, cfg['channel']), rs.randint(0, 21, n_val)
    steps = int(epochs * n_train / batch)
    out = {'shape': {'M': M, 'batch': batch, 'train': n_train, 'val': n_val, 'steps': steps, 'evaluations': 1},
           'what': 't_step of fit() as models_gcn.py:183-184 defines it (wall time of fit / steps; the evaluation pass over the '
                   'validation set and the checkpoint of the last step inside), ' + tag}
    old_home = os.environ.get('CHEBGCN_HOME')
    with tempfile.TemporaryDirectory() as tmp:
        os.environ['CHEBGCN_HOME'] = tmp
        try:
            for mode in ('warmup', 'numpy_float64', 'staged'):
                torch.manual_seed(0)
                np.random.seed(0)
                net = models_gcn.cgcnn({'device': dev}, [L] * len(cfg['F']), cfg['F'], cfg['K'], cfg['p'], cfg['M'],
                                       filter='chebyshev5', brelu='b2relu', pool='mpool1', initial='he', channel=cfg['channel'],
                                       regularization=5e-4, dropout=0.5, batch_size=batch, learning_rate=0.001, decay_rate=0.9,
                                       momentum=0.9, num_epochs=epochs if mode != 'warmup' else max(1, epochs // 8),
                                       eval_frequency=steps, dir_name='bench_fit',
                                       verbose=False)
                t0 = time.perf_counter()
                a, b = (net.stage(train), net.stage(val)) if mode == 'staged' else (train, val)
                torch.cuda.synchronize()
                t_stage = time.perf_counter() - t0
                with contextlib.redirect_stdout(io.StringIO()):
                    _, _, t_step = net.fit(a, tl, b, vl)
                if mode == 'warmup':                        # (first use of this shape: library and allocator warm-up, not reported)
                    del net, a, b
                    continue
                out[mode] = {'t_step_ms': 1e3 * t_step, 'windows_per_s': batch / t_step}
                if mode == 'staged':
                    out[mode]['staging_ms_outside_fit'] = 1e3 * t_stage
                del net, a, b
                torch.cuda.empty_cache()
        finally:
            if old_home is None:
                os.environ.pop('CHEBGCN_HOME', None)
            else:
                os.environ['CHEBGCN_HOME'] = old_home
    out['pcie_inclusive_over_staged'] = out['numpy_float64']['t_step_ms'] / out['staged']['t_step_ms']
    return out


def dp_overhead_leg(net, step, first, steps, plain_ms, dev):
    """Fixed cost of the data-parallel plumbing, visible at N = 1: the same training step with the model wrapped in
    ``dist.DataParallel`` on backend nccl (RCCL) with a world of ONE rank -- the broadcast, the gradient hooks and
    the three asynchronous all-reduces of ``bench.py --gpus N``, each a self-copy -- against the plain step."""
    import torch
    import torch.distributed as dist
    from gcn_fmri_decoding_amd import dist as gdist
    def timed(i0):
        for i in range(i0, i0 + 3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(i0 + 3, i0 + 3 + steps):
            step(i)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    plain_now = timed(first)                         # measured again, right in front of the wrapped steps (same clocks, same caches)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % free_port(), rank=0, world_size=1, device_id=dev)
    try:
        dp = gdist.DataParallel(net)
        ms = timed(first + steps + 3)
        dp.remove()
    finally:
        dist.destroy_process_group()
    return {'plain_ms_per_step': plain_now, 'dp_world1_ms_per_step': ms, 'overhead_ms': ms - plain_now, 'steps': steps,
            'timed_region_ms_per_step': plain_ms,
            'what': 'same step under dist.DataParallel on RCCL with world_size 1 (hooks + 5 async all-reduces + waits) vs plain, '
                    '%d steps each, back to back' % steps}


def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """``python bench.py --gpus N`` with N > 1 and no torchrun environment: start N fresh rank processes
    (``python -m torch.distributed.run``, rendezvous on 127.0.0.1) as a CHILD -- this process has not imported torch
    or touched a GPU, and is not replaced -- relay rank 0's JSON line and the child's exit code."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write('bench.py: the rank processes ended without a result line\n')
        rc = 1
    sys.exit(rc)


def ranks_seen(dist, dev):
    """Sum over ranks of 1: what the collective library itself counts (the driver can tell RCCL saw N ranks)."""
    import torch
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    return int(t.item())


def stub_main(args, world, rank):
    """``--stub``: the launcher / rendezvous / timing / reporting path with a stand-in step on CPU tensors (backend
    gloo) -- what tests/test_bench_launcher.py runs where there is no GPU.  Not a measurement."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(args.backend if args.backend != 'nccl' else 'gloo', rank=rank, world_size=world)
    w = torch.zeros(1000)
    for _ in range(args.warmup):
        w += 1
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g = torch.full((1000,), float(rank + 1))
        if world > 1:
            dist.all_reduce(g)
        w += g / world
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    seen = 1
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        seen = ranks_seen(dist, 'cpu')
    if rank == 0:
        print(json.dumps({'metric': METRIC, 'value': args.batch * world * args.steps / dt, 'unit': 'windows/s', 'n_gpus': world,
                          'n_ranks_seen': seen, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'config': {'workload': 'STUB (launcher test, no kernels)', 'global_batch': args.batch * world,
                                     'parallelism': 'dp%d' % world}, 'stub': True,
                          'checksum': float(w.sum())}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=64, help='windows per GPU')
    ap.add_argument('--nodes', type=int, default=10000)
    ap.add_argument('--korder', type=int, default=5)
    ap.add_argument('--block-dura', type=int, default=15)
    ap.add_argument('--cpu-windows', type=int, default=16, help='0 disables the CPU baseline leg')
    ap.add_argument('--kernel-legs', type=int, default=1,
                    help='1: also time the recurrence kernels alone at the north-star shape (K=5, Fin=32, batch 256) and at '
                         'configs[3] (K=25, Fin=64, batch 64), the wide layer of configs[4] (Fin=60, Fout=256) forward + '
                         'backward in fp32 / bf16 / split bf16, the reference\'s own training shape (atlas graph, K=10, batch 128) '
                         'and the data-parallel plumbing on a world of one; N=1 only')
    ap.add_argument('--overlap-bwd-w', type=int, default=-1,
                    help="contract_bwd_w on a second stream (ops.overlap_bwd_w): 1 / 0 force it, -1 = the library's default ('auto': wide "
                         'layers only)')
    ap.add_argument('--repeats', type=int, default=3,
                    help='the timed region (exactly --steps steps) is run this many times; `value` comes from the FIRST, the '
                         'others are reported beside it (ms_per_step_repeats)')
    ap.add_argument('--instrumented-steps', type=int, default=10,
                    help='steps of the separate instrumented pass AFTER the timed regions (HIP events around every hot-kernel '
                         'launch, no second stream): the source of `roofline` and `kernels`; 0 disables it')
    ap.add_argument('--step-graph', type=int, default=0,
                    help='1: capture the training step as one HIP graph (cgcnn.enable_step_graph; single GPU only)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL)')
    ap.add_argument('--stub', action='store_true', help='launcher / reporting path with a stand-in CPU step (tests)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args.gpus, sys.argv[1:])            # does not return
    if world != args.gpus:
        raise SystemExit('--gpus %d but the launcher started %d rank processes' % (args.gpus, world))
    if args.stub:
        return stub_main(args, world, rank)

    # stdout carries ONE line: the result.  Libraries write banners to file descriptor 1 (RCCL prints its version block
    # when the first communicator comes up): everything but the result line goes to stderr from here on.
    result_out = os.fdopen(os.dup(1), 'w')
    sys.stdout.flush()
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    from gcn_fmri_decoding_amd import models_gcn, ops
    from gcn_fmri_decoding_amd import dist as gdist
    if args.overlap_bwd_w >= 0:
        ops.overlap_bwd_w = bool(args.overlap_bwd_w)

    Ls, perm = load_graph(args.nodes, 1, rank, world, barrier)
    cfg = dict(F=[32] * 6, K=[args.korder] * 6, p=[1] * 6, M=[512, 256, 22], channel=args.block_dura)
    torch.manual_seed(0)
    # one Laplacian per conv layer (no pooling: the same matrix object six times, one device graph);
    # a shorter list makes the constructor print its consistency check, as the reference does
    net = models_gcn.cgcnn({'device': dev}, [Ls[0]] * len(cfg['F']), cfg['F'], cfg['K'], cfg['p'], cfg['M'], filter='chebyshev5',
                           brelu='b2relu', pool='mpool1', initial='he', channel=cfg['channel'], regularization=5e-4,
                           dropout=0.5, batch_size=args.batch, learning_rate=0.001, decay_rate=0.9, momentum=0.9,
                           verbose=False)
    n_seen = 1
    if args.step_graph and world == 1:
        net.enable_step_graph(True)
    if world > 1:
        gdist.DataParallel(net)
        n_seen = ranks_seen(dist, dev)

    # synthetic dataset resident in HBM: [S, N, block_dura] z-scored signals, uniform labels
    S = 4 * args.batch
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    data = torch.randn((S, args.nodes, args.block_dura), generator=g, device=dev)
    labels = torch.randint(0, 21, (S,), generator=g, device=dev)
    perm_dev = net.compose_perm(perm)                 # perm_data_3d's index map, composed with the model's internal vertex order
    repeats = max(1, args.repeats)
    n_dp = 2 * (DP_STEPS + 3) if (world == 1 and args.kernel_legs) else 0
    n_order = args.warmup + repeats * args.steps + args.instrumented_steps + n_dp
    order = torch.stack([torch.randperm(S, generator=g, device=dev)[:args.batch].to(torch.int32) for _ in range(n_order)])

    def step(i):
        idx = order[i]
        x = net.as_internal(ops.perm_data(data, perm_dev, idx))          # perm_data_3d + batch gather, on device
        return net.train_step(x, labels[idx.long()])

    for i in range(args.warmup):
        step(i)
    # ---- the timed region: exactly --steps steps of ONE step variant (no event timers), barrier + synchronize on both sides
    region_ms = []
    nxt = args.warmup
    for r in range(repeats):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nxt, nxt + args.steps):
            _, loss = step(i)
        torch.cuda.synchronize()
        barrier()
        dt_r = time.perf_counter() - t0
        nxt += args.steps
        if world > 1:
            t = torch.tensor([dt_r], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_r = float(t.item())
        region_ms.append(1e3 * dt_r / args.steps)
    dt = region_ms[0] * args.steps / 1e3
    loss = float(loss)
    # ---- separate instrumented pass: HIP events around every hot-kernel launch on the launch stream (no second stream
    # on these steps: a kernel's time must not include a neighbour)
    kern, sampled = {}, 0
    exposed_ms = None
    if args.instrumented_steps > 0:
        ops.timers = ops.KernelTimers(every=1, by_dispatch=True)
        if world > 1:
            net._dp.exposed_events = []
        for i in range(nxt, nxt + args.instrumented_steps):
            ops.timers.next_step()
            step(i)
        kern = ops.timers.summary()
        sampled = ops.timers.sampled_steps
        ops.timers = None
        nxt += args.instrumented_steps
        if world > 1:
            ev, net._dp.exposed_events = net._dp.exposed_events, None
            if ev:
                t = torch.tensor([float(np.mean([a.elapsed_time(b) for a, b in ev]))], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                exposed_ms = float(t.item())

    if rank == 0:
        global_batch = args.batch * world
        line = {
            'metric': METRIC, 'value': global_batch * args.steps / dt, 'unit': 'windows/s', 'n_gpus': world,
            'n_ranks_seen': n_seen,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: ChebNet K=%d x6 conv (F=32, p=1, b2relu), FC 512-256-22, '
                                   'synthetic N=%d kNN-8 graph -> M=%d, block_dura=%d, batch %d per GPU'
                                   % (args.korder, args.nodes, Ls[0].shape[0], args.block_dura, args.batch),
                       'global_batch': global_batch, 'parallelism': 'dp%d' % world,
                       'step': 'fwd+loss+bwd+allreduce+Adam, batch gathered on device'},
            'final_loss': loss,
            # HIP-event time between this rank's last backward kernel and the point where every gradient bucket has arrived
            # (max over ranks of the mean over the instrumented steps): the part of the all-reduce backward does not hide
            'allreduce_exposed_ms': exposed_ms,
            'ms_per_step_repeats': {'all': region_ms, 'median': float(np.median(region_ms)), 'min': float(np.min(region_ms)),
                                    'max': float(np.max(region_ms)),
                                    'what': '%d timed regions of %d steps each, back to back; `value` / `ms_per_step` are the first'
                                            % (repeats, args.steps)},
        }
        if kern:
            # `kern` is keyed 'op | kernel template(s) chebgcn_last_dispatch() reported'.  Two tables come out of it: per op
            # (`kernels`, as in earlier rounds) and per kernel SYMBOL (`kernels_by_symbol`): the forward recurrence kernel also
            # serves the input gradient (on dy with the transposed operator), the forward contraction kernel its contraction.
            # The dominant kernel of the step -- `roofline` -- is chosen by symbol, so that its average launch time can be
            # compared with the AverageNs rocprofv3 --kernel-trace --stats reports for that symbol on the same command
            # (profiles/*_bench_kernel_stats_serial.csv: --overlap-bwd-w 0, every kernel alone on the device)
            def merged(key_of):
                out = {}
                for k, v in kern.items():
                    d = out.setdefault(key_of(k), {'launches': 0, 'total_ms': 0.0, 'bytes': 0.0, 'flops': 0.0})
                    for f in d:
                        d[f] += v[f]
                for d in out.values():
                    d['avg_ms'] = d['total_ms'] / d['launches']
                return out
            raw_keys = list(kern)
            by_sym = merged(lambda k: k.split(' | ', 1)[1])
            kern = merged(lambda k: k.split(' | ', 1)[0])
            dom = max(by_sym, key=lambda k: by_sym[k]['total_ms'])
            d = by_sym[dom]
            achieved = d['bytes'] / (d['total_ms'] * 1e-3) / 1e9
            traffic = traffic_note = None
            tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                traffic = tj.get('by_kernel', {}).get(dom)
                traffic_note = tj.get('_note')
            line['roofline'] = {'kernel': dom, 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                                'traffic_source': 'profiles/traffic.json by_kernel[%r] (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE '
                                                  'passes of tools/pmc_traffic.sh over this kernel at its Fin = Fout = 32 launch of the '
                                                  'step, not measured in this run)' % dom,
                                'avg_launch_ms': d['avg_ms'], 'launches': d['launches'],
                                'algorithmic_bytes_per_launch': d['bytes'] / d['launches'],
                                'chosen_by': 'largest total time per kernel symbol over the instrumented steps',
                                'ops': sorted(k.split(' | ', 1)[0] for k in raw_keys if k.split(' | ', 1)[1] == dom)}
            step_ms = 1e3 * dt / args.steps            # the same region `value` comes from
            step_bytes = sum(v['bytes'] for v in kern.values()) / max(sampled, 1)
            line['step_roofline'] = {'algorithmic_bytes_per_step': step_bytes, 'frac': step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     'what': 'algorithmic bytes of the hot kernels of one step (SURVEY.md 8d) / ms_per_step / 8 TB/s'}
            table = lambda src: {k: {'avg_ms': v['avg_ms'], 'launches': v['launches'],
                                     'GBps': v['bytes'] / (v['total_ms'] * 1e-3) / 1e9,
                                     'frac_hbm': v['bytes'] / (v['total_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     'TFLOPs': v['flops'] / (v['total_ms'] * 1e-3) / 1e12,
                                     'share_of_step': v['total_ms'] / max(sampled, 1) / step_ms}
                                 for k, v in src.items()}
            line['kernels'] = table(kern)
            line['kernels_by_symbol'] = table(by_sym)
            line['kernel_timing'] = ('HIP events around every launch of these kernels on the launch stream, in a separate '
                                     'instrumented pass of %d steps AFTER the timed regions (the timed steps carry no events)' % sampled)
        if world == 1 and args.kernel_legs:
            g0 = net.graphs[0]
            line['northstar'] = kernel_leg(g0, 256, 32, 5, 100, 'north-star shape of BASELINE.json: K=5 recurrence, Fin=32, batch 256, '
                                                                'M=10466; target frac >= 0.40')
            line['config4'] = kernel_leg(g0, 64, 64, 25, 30, 'BASELINE configs[3]: K=25, Fin=Fout=64, batch 64 -- these entries time the two '
                                                             'recurrence launches alone; the whole layer is in "layer"')
            line['config4']['layer'] = layer_leg(g0, 64, 64, 25, 64, 5, 'BASELINE configs[3]: the whole K=25, 64 -> 64 layer forward + backward, '
                                                 'HIP kernels only (no head, no optimizer)', ('f32', 'bf16x3'))
            line['config5'] = layer_leg(g0, 64, 60, 5, 256, 10, 'BASELINE configs[4]: single wide layer forward + backward, HIP kernels only '
                                        '(no head, no optimizer)')
            line['dp_overhead'] = dp_overhead_leg(net, step, nxt, DP_STEPS, 1e3 * dt / args.steps, dev)
            del net, data
            torch.cuda.empty_cache()
            line['refshape'] = {'n360': refshape_leg(dev, 360, 100, 10), 'n1000': refshape_leg(dev, 1000, 100, 10)}
            line['pool6'] = pool6_leg(dev, 10, 3)
            from gcn_fmri_decoding_amd import graph as G
            line['fit'] = {'configs1': fit_leg(dev, Ls[0], cfg, args.batch, 8 * args.batch, 2 * args.batch, 25,
                                               'BASELINE configs[1] (M = %d, batch %d)' % (Ls[0].shape[0], args.batch)),
                           'refshape_n360': fit_leg(dev, G.synthetic_graph(360, k=8, levels=1)[0][0],
                                                    dict(F=[32] * 6, K=[10] * 6, p=[1] * 6, M=[512, 256, 22], channel=15), 128, 1024,
                                                    256, 25, "the reference's training shape (N = 360, K = 10, batch 128)")}
        if world == 1 and args.cpu_windows > 0:
            line['cpu_baseline'] = cpu_baseline(Ls[:1], cfg, args.cpu_windows)
        result_out.write(json.dumps(line) + '\n')
        result_out.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
