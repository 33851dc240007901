"""Round 6: pooling between two vertex orders and the pooling network on relabelled levels.  Needs an MI355X: ``-m gpu``.

Reference semantics: ``mpool1`` / ``apool1`` (lib_new/models_gcn.py:631-648) pool ``p`` CONSECUTIVE vertices of the tree order
``coarsening.compute_perm`` builds (lib_new/coarsening.py:168-215).  ``cgcnn`` keeps every level that an ordered recurrence
kernel serves in descending-row-length order, so a pooled layer there pools through index maps
(``ops.pool_maps`` -> ``chebgcn_pool_gather_fwd`` / ``chebgcn_pool_scatter_bwd``): bit-exact against a NumPy restatement of the
reference's pooling composed with the two permutations.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import csr_from, load_golden, record_measured

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


def _P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pool_ref(y_ref, p, kind, relu):
    """Reference pooling on [B, F, M] in the REFERENCE order: (out, grad-selector [B, F, M]) -- the selector is d(out_j)/d(y_v)
    for v in cluster j (first maximum wins; zero where ``relu`` and the maximum is not positive / the member was not positive)."""
    B, F, M = y_ref.shape
    c = y_ref.reshape(B, F, M // p, p)
    if kind == 0:
        out = c.max(axis=3)
        arg = c.argmax(axis=3)                               # first maximum
        sel = np.zeros_like(c)
        np.put_along_axis(sel, arg[..., None], 1.0, axis=3)
        if relu:
            sel *= (out > 0)[..., None]
    else:
        out = c[..., 0].copy()
        for i in range(1, p):                                # members added in order, like the kernels (and TF's AvgPool window)
            out = out + c[..., i]
        out = out * np.float32(1.0 / p)
        sel = np.full_like(c, 1.0 / p)
        if relu:
            sel *= (c > 0)
    return out.astype(np.float32), sel.reshape(B, F, M).astype(np.float32)


@pytest.mark.parametrize('M,p,F,B', [(12672, 4, 8, 5), (3168, 4, 16, 7), (792, 2, 5, 3), (20480, 8, 3, 2), (40, 4, 3, 9)])
@pytest.mark.parametrize('kind', [0, 1])
@pytest.mark.parametrize('bias_kind', [0, 1, 2])
@pytest.mark.parametrize('mapped', [True, False])
def test_pool_gather_scatter_vs_numpy(dev, M, p, F, B, kind, bias_kind, mapped):
    """chebgcn_pool_gather_fwd / chebgcn_pool_scatter_bwd against NumPy: random source and pooled orders (or the identity),
    max and average pooling, the three bias kinds; pads poisoned.  Outputs and dy bit-exact (selection, no arithmetic but the
    1/p of the average); bias sums to fp32 round-off (another, fixed, order)."""
    from gcn_fmri_decoding_amd import _lib, ops
    lib = _lib.lib()
    if bias_kind and (M, kind) not in ((12672, 0), (3168, 0), (792, 1), (40, 0), (40, 1)) and not mapped:
        pytest.skip('bias kinds on a subset of the unmapped shapes')
    rs = np.random.RandomState(M + 7 * p + kind)
    Mo, Mp, Mpo = M // p, ops.plane_stride(M), ops.plane_stride(M // p)
    src = rs.permutation(M) if mapped else None
    dst = rs.permutation(Mo) if mapped else None
    relu = 1
    y_ref = rs.randn(B, F, M).astype(np.float32)
    y_ref = np.maximum(y_ref, 0)                             # what contract_fwd(pool = 1, relu) leaves
    y_ref[0, 0, :2 * p] = 0                                  # whole clusters at zero: no gradient through the ReLU
    if M >= 16:
        y_ref[0, 0, 4 * p:4 * p + p] = 1.5                   # a tie: the first member (in the reference's order) wins
    y_int = y_ref[:, :, src] if mapped else y_ref
    y = torch.full((B, F, Mp), float('nan'), device=dev)
    y[:, :, :M] = torch.as_tensor(y_int).to(dev)
    pmap = smap = None
    if mapped:
        pmap, smap = ops.pool_maps(p, src, dst, M, dev)
        # the maps against their definition
        pm, sm = pmap.cpu().numpy(), smap.cpu().numpy()
        inv_src = np.argsort(src)
        assert np.array_equal(pm.reshape(Mo, p), inv_src[p * dst[:, None] + np.arange(p)[None, :]])
        assert np.array_equal(sm[pm], np.arange(M))
    out = torch.full((B, F, Mpo), float('nan'), device=dev)
    sel = torch.zeros((B, F, Mpo), dtype=torch.uint8, device=dev)
    _lib.check(lib.chebgcn_pool_gather_fwd(_P(y), _P(pmap), _P(out), _P(sel), B, M, F, p, kind, relu, _stream()), 'pool_gather_fwd')
    assert _lib.last_dispatch() == ('pool_gather_fwd_kernel<map>' if mapped else 'pool_gather_fwd_kernel')
    o_ref, s_ref = _pool_ref(y_ref, p, kind, relu)
    o_int = o_ref[:, :, dst] if mapped else o_ref
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :, :Mo], o_int), 'pooled output differs'
    assert np.all(got[:, :, Mo:] == 0)                       # the padding of the pooled planes is zeroed
    # backward
    do_ref = rs.randn(B, F, Mo).astype(np.float32)
    dout = torch.full((B, F, Mpo), float('nan'), device=dev)
    dout[:, :, :Mo] = torch.as_tensor(do_ref[:, :, dst] if mapped else do_ref).to(dev)
    dy = torch.full((B, F, Mp), float('nan'), device=dev)
    dbias = None if bias_kind == 0 else torch.full((F,) if bias_kind == 1 else (F, Mp), float('nan'), device=dev)
    nws = lib.chebgcn_pool_scatter_bwd_workspace(B, M, F, p, bias_kind)
    assert (nws > 0) == (bias_kind != 0)
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
    _lib.check(lib.chebgcn_pool_scatter_bwd(_P(dout), _P(sel), _P(smap), _P(dy), _P(dbias), bias_kind, B, M, F, p, kind, relu, _P(ws),
                                            nws, _stream()), 'pool_scatter_bwd')
    kname = ['CHEBGCN_BIAS_NONE', 'CHEBGCN_BIAS_FILTER', 'CHEBGCN_BIAS_VERTEX'][bias_kind]
    want = 'pool_scatter_bwd_kernel<%s>%s' % (kname, '<map>' if mapped else '') + (' + pool_bias_reduce_kernel<%s>' % kname if bias_kind else '')
    assert _lib.last_dispatch() == want, _lib.last_dispatch()
    dy_ref = (np.repeat(do_ref, p, axis=2) * s_ref).astype(np.float32)           # [B, F, M], reference order
    dy_int = dy_ref[:, :, src] if mapped else dy_ref
    gdy = dy.cpu().numpy()
    assert np.array_equal(gdy[:, :, :M], dy_int), 'dy differs'
    assert np.all(gdy[:, :, M:] == 0)
    if bias_kind == 2:
        db = dbias.cpu().numpy()
        ref = dy_int.astype(np.float64).sum(axis=0)
        assert np.abs(db[:, :M] - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-30) * np.sqrt(B)
        assert np.all(db[:, M:] == 0)
    elif bias_kind == 1:
        db = dbias.cpu().numpy()
        ref = dy_int.astype(np.float64).sum(axis=(0, 2))
        assert np.abs(db - ref).max() <= 2e-6 * np.abs(dy_int).sum(axis=(0, 2)).max()
    # the same gradient from chebgcn_brelu_pool_bwd (tree order only): from 2048 vertices and with the workspace it asks for -> this
    # kernel; smaller planes or no workspace -> the scalar kernel of rounds 1-5.  Both bit-identical in dy.
    if not mapped and bias_kind != 1:
        for give_ws in (True, False):
            dy2 = torch.full((B, F, Mp), float('nan'), device=dev)
            db2 = None if bias_kind == 0 else torch.zeros((F, Mp), device=dev)
            n2 = lib.chebgcn_brelu_pool_bwd_workspace(B, M, F, p, bias_kind) if give_ws else 0
            w2 = torch.empty(max(n2, 1), dtype=torch.uint8, device=dev)
            # (the fused contraction epilogue's byte: the winning member / the positive-member mask, and `out` for the ReLU)
            arg = sel.clone()
            if kind == 0:
                arg[arg == 0xFF] = 0
            _lib.check(lib.chebgcn_brelu_pool_bwd(_P(dout), _P(out), _P(arg), _P(dy2), _P(db2), bias_kind, B, M, F, p, kind, relu,
                                                  _P(w2) if give_ws else None, n2, _stream()), 'brelu_pool_bwd')
            name = _lib.last_dispatch()
            assert name.startswith('pool_scatter_bwd_kernel<' if ((give_ws or bias_kind == 0) and Mp >= 2048) else 'brelu_pool_bwd_kernel<'), name
            assert np.array_equal(dy2.cpu().numpy()[:, :, :M], dy_int), name
            if bias_kind == 2:
                ref = dy_int.astype(np.float64).sum(axis=0)
                assert np.abs(db2.cpu().numpy()[:, :M] - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-30) * np.sqrt(B)


@pytest.mark.parametrize('name', ['inference_pool_n212', 'inference_pool6_n512'])
@pytest.mark.parametrize('pool_kind', ['mpool1', 'apool1'])
def test_relabelled_levels_are_invisible_with_pooling(dev, name, pool_kind):
    """A pooling network whose levels are FORCED into the length order (``vertex_order = 'length!'``: these fixtures' graphs are
    too small for an ordered kernel, the relabelling and the pooling maps are exercised all the same) against the same network
    in the reference's order: logits, every gradient and the variables after a step agree to fp32 summation order, biases and
    the first FC layer's rows cross the boundary in the reference's order, checkpoints are interchangeable."""
    from gcn_fmri_decoding_amd import models_gcn, ops
    z = load_golden(name)
    Ls = [csr_from(z, 'L%d' % i) for i in range(int(z['nlevels']))]
    F, K, p, M = z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist()
    params = {k[len('param:'):]: z[k].copy() for k in z.files if k.startswith('param:')}
    x = torch.as_tensor(z['x']).to(dev)
    B = x.shape[0]
    labels = torch.as_tensor(np.arange(B) % M[-1]).to(dev)
    nets = {}
    import os
    for mode in ('reference', 'length!'):
        os.environ['CHEBGCN_VERTEX_ORDER'] = mode
        try:
            net = models_gcn.cgcnn({'device': dev}, Ls, F, K, p, M, channel=int(z['channel']), brelu=str(z['brelu']), pool=pool_kind,
                                   batch_size=B, regularization=5e-4, dropout=1, verbose=False)
        finally:
            os.environ.pop('CHEBGCN_VERTEX_ORDER', None)
        net.contraction = 'f32'
        for k, v in params.items():
            net.set_variable(k, v)
        nets[mode] = net
    ref, rel = nets['reference'], nets['length!']
    assert ref.vertex_order == 'reference' and not ref._relabelled
    assert rel.vertex_order == 'length' and all(o is not None for o in rel._orders)
    assert any(m is not None for m in rel._pool_maps) and all(m is None for m in ref._pool_maps)
    for k in params:                                         # what was set is what is read, in the reference's shape and order
        assert np.array_equal(rel.get_var(k), params[k]), k
    with torch.no_grad():
        la = ref.inference(x, 1).cpu().numpy()
        lb = rel.inference(x, 1).cpu().numpy()
    if pool_kind == 'mpool1' and 'logits' in z.files:
        assert np.abs(la - z['logits']).max() <= 2e-5 * np.abs(z['logits']).max()       # the reference-generated vector
    e_logits = np.abs(la - lb).max() / np.abs(la).max()
    assert e_logits <= 1e-5, e_logits
    xs = ops.plane_storage(x)
    ref.train_step(xs, labels)
    rel.train_step(xs, labels)
    worst = {}
    for k in params:
        ga, gb = ref.gradient(k).cpu().numpy().astype(np.float64), rel.gradient(k).cpu().numpy().astype(np.float64)
        worst[k] = float(np.abs(ga - gb).max() / max(np.abs(ga).max(), 1e-30))
        assert worst[k] <= 5e-5, (k, worst[k])
    record_measured('relabelled_levels_invisible[%s,%s]' % (name, pool_kind), logits=float(e_logits), **worst)
    sd = rel.state_dict()
    twin = models_gcn.cgcnn.from_checkpoint(sd, config={'device': dev})
    for k in params:
        assert np.array_equal(twin.get_var(k), rel.get_var(k)), k
    ref.load_state_dict(sd)                                  # a relabelled model's checkpoint in a reference-order model
    with torch.no_grad():
        lc = ref.inference(x, 1).cpu().numpy()
        ld = rel.inference(x, 1).cpu().numpy()
    assert np.abs(lc - ld).max() <= 1e-5 * np.abs(lc).max()
    # the relabelled network's step captured as a HIP graph (the mapped pooling kernels, their workspaces and the batched weight
    # re-indexing inside the capture) replays bit-identically to the eager step of a twin
    os.environ['CHEBGCN_VERTEX_ORDER'] = 'length!'
    try:
        twin = models_gcn.cgcnn.from_checkpoint(rel.state_dict(), config={'device': dev})
    finally:
        os.environ.pop('CHEBGCN_VERTEX_ORDER', None)
    assert twin._relabelled
    twin._loss_ema = None if rel._loss_ema is None else rel._loss_ema.clone()
    twin.global_step = rel.global_step
    rel.enable_step_graph(True)
    for _ in range(4):
        a = rel.train_step(xs, labels)[1]
        b = twin.train_step(xs, labels)[1]
    torch.cuda.synchronize()
    assert rel._sg is not None and torch.equal(rel._flat, twin._flat) and float(a) == float(b)


def test_contraction_value_is_validated(dev):
    from gcn_fmri_decoding_amd import models_gcn
    z = load_golden('inference_flat_n212')
    Ls = [csr_from(z, 'L%d' % i) for i in range(int(z['nlevels']))]
    net = models_gcn.cgcnn({'device': dev}, Ls, z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist(),
                           channel=int(z['channel']), brelu=str(z['brelu']), batch_size=2, verbose=False)
    with pytest.raises(ValueError):
        net.contraction = 'bf16x2'
    net.contraction = 'bf16x3'
    assert net.layer_precisions() == ['bf16x3'] * len(z['F'])


def test_loss_average_reading_switch(dev):
    """``ema_zero_debias`` (models_gcn.py:269-275 read two ways, see cgcnn): default = TF >= 1.0 (the zero-initialised shadow as
    is: 0.1 * loss after one step), True = TF 0.12 (debiased: the loss itself)."""
    from gcn_fmri_decoding_amd import models_gcn, ops
    z = load_golden('inference_flat_n212')
    Ls = [csr_from(z, 'L%d' % i) for i in range(int(z['nlevels']))]
    x = ops.plane_storage(torch.as_tensor(z['x']).to(dev))
    labels = torch.as_tensor(np.arange(x.shape[0]) % int(z['M'][-1])).to(dev)
    vals = {}
    for zd in (False, True):
        torch.manual_seed(0)
        net = models_gcn.cgcnn({'device': dev}, Ls, z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist(),
                               channel=int(z['channel']), brelu=str(z['brelu']), batch_size=x.shape[0], dropout=1, verbose=False)
        assert net.ema_zero_debias is False
        net.ema_zero_debias = zd
        vals[zd] = [float(net.train_step(x, labels)[1]) for _ in range(3)]
    raw, deb = np.array(vals[False]), np.array(vals[True])
    t = np.arange(1, 4)
    np.testing.assert_allclose(raw, deb * (1 - 0.9 ** t), rtol=2e-6)


@pytest.mark.parametrize('contraction', ['auto', 'f32'])
def test_fit_tracks_float64_oracle_under_auto(dev, contraction, tmp_path, monkeypatch):
    """What the default arithmetic costs a training run ("Arithmetic"): a 20-step ``fit`` of the pooling ChebNet
    fixture (F up to 128: split bf16 in the wide layers under ``contraction = 'auto'``) against the FLOAT64 oracle's loop --
    the reported ``loss_average`` series and the validation losses.  Under 'auto' single gradient elements differ by flipped
    ReLU / max-pool decisions (tests/test_gpu_bench_shapes.py: 99 % quantile bound 2e-3); over 20 Adam steps the loss
    trajectory stays as close to the float64 one as the exact-product path ('f32') does -- fp32 itself drifts from float64 at
    the 1e-3 level over 20 Adam steps (the first updates are +-lr whatever the gradient's size, and decisions flip).  Measured
    (profiles/r06_parity_measured.jsonl): 'auto' 1.5e-3 (loss_average series) / 1.2e-3 (validation losses), 'f32' 1.8e-4 / 2.3e-3;
    bound 5e-3 for both."""
    from gcn_fmri_decoding_amd import models_gcn
    from oracle import layers_ref as R
    from oracle import loop_ref as LR
    monkeypatch.setenv('CHEBGCN_HOME', str(tmp_path))
    z = load_golden('inference_pool6_n512')
    Ls = [csr_from(z, 'L%d' % i) for i in range(int(z['nlevels']))]
    F, K, p, M = z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist()
    params = {k[len('param:'):]: z[k].copy() for k in z.files if k.startswith('param:')}
    batch, S, Sv, epochs, reg = 4, 16, 6, 5, 5e-4
    rs = np.random.RandomState(11)
    M0, C = Ls[0].shape[0], int(z['channel'])
    data, labels = rs.randn(S, M0, C), rs.randint(0, M[-1], S)
    vdata, vlabels = rs.randn(Sv, M0, C), rs.randint(0, M[-1], Sv)

    class Seeded(models_gcn.cgcnn):
        seed_params = None

        def _init_variables(self):
            super()._init_variables()
            if self.seed_params is not None and self.device.type == 'cuda':
                for k, v in self.seed_params.items():
                    self.set_variable(k, v)
    net = Seeded({'device': dev}, Ls, F, K, p, M, channel=C, brelu=str(z['brelu']), num_epochs=epochs, eval_frequency=5,
                 batch_size=batch, regularization=reg, dropout=1, dir_name='auto_fit', verbose=False)
    net.contraction = contraction
    assert ('bf16x3' in net.layer_precisions()) == (contraction == 'auto')
    net.seed_params = params
    net.record_fit = True
    np.random.seed(7)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        acc, losses, _ = net.fit(data, labels, vdata, vlabels)
    onet = R.Net(Ls, F, K, p, M, channel=C, brelu=str(z['brelu']), regularization=reg, dtype=np.float64)
    oparams = {k: v.astype(np.float64) for k, v in params.items()}
    np.random.seed(7)
    log = LR.fit(onet, oparams, data, labels, vdata, vlabels, epochs, batch, 5)
    assert log['num_steps'] == 20 and len(net.fit_log['loss_average']) == 20
    la, lo = np.array(net.fit_log['loss_average']), np.array(log['loss_average'])
    e_series = float(np.abs(la - lo).max() / np.abs(lo).max())
    e_val = float(np.abs(np.array(losses) - np.array(log['losses'])).max() / np.abs(np.array(log['losses'])).max())
    record_measured('fit_tracks_float64_oracle[%s]' % contraction, loss_average_series=e_series, validation_losses=e_val,
                    final_loss_average=float(la[-1]))
    assert e_series <= 5e-3 and e_val <= 5e-3, (contraction, e_series, e_val)


def test_reindex_weights_batch_is_the_index_map(dev):
    """chebgcn_reindex_weights_batch: every layer's W'[fo*K + k][fin] = W[fin*K + k][fo] in one launch, bit for bit (more than
    16 layers: two launches)."""
    from gcn_fmri_decoding_amd import _lib, ops
    shapes = [(32, 5, 32), (64, 25, 64), (128, 5, 96), (15, 20, 8), (1, 1, 1)] + [(3 + i, 2, 2 + i) for i in range(14)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    Ws = [torch.randn((fi * k, fo), generator=gen, device=dev) for fi, k, fo in shapes]
    outs = ops.reindex_weights_batch(Ws, shapes)
    assert _lib.last_dispatch() == 'reindex_weights_batch_kernel' and len(outs) == len(shapes)
    for W, Wt, (fi, k, fo) in zip(Ws, outs, shapes):
        ref = W.view(fi, k, fo).permute(2, 1, 0).reshape(fo * k, fi)
        assert torch.equal(Wt, ref), (fi, k, fo)


# ---------------------------------------------------------------------------------------
# the ReluGrad of a layer in the epilogue of the input gradient of the layer above (ops.GateLink)
# ---------------------------------------------------------------------------------------

@pytest.mark.parametrize('B,M,Fin,K,Fout', [(25, 10466, 32, 5, 32), (27, 10466, 32, 5, 15), (110, 2500, 8, 3, 32), (64, 4111, 20, 7, 7)])
def test_contract_fwd_gated_is_contract_fwd_then_the_mask(dev, B, M, Fin, K, Fout):
    """chebgcn_contract_fwd_gated against chebgcn_contract_fwd (no bias, no ReLU) followed by the mask in torch: bit for bit
    (same products, same order), random masks with whole rows / quads of zeros."""
    from gcn_fmri_decoding_amd import _lib, ops
    lib = _lib.lib()
    assert lib.chebgcn_contract_fwd_gated_supported(B, M, Fin, K, Fout) == 1
    assert lib.chebgcn_contract_fwd_gated_supported(B, M, Fin, K, 33) == 0
    assert lib.chebgcn_contract_fwd_gated_supported(1, 300, Fin, K, Fout) == 0          # a small launch: another kernel's shape
    Mp = ops.plane_stride(M)
    gen = torch.Generator(device=dev)
    gen.manual_seed(M + Fout)
    stack = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    W = torch.randn((Fin * K, Fout), generator=gen, device=dev) * 0.2
    gate = torch.randint(0, 16, (B, Fout, Mp // 4), generator=gen, device=dev, dtype=torch.uint8)
    gate[0, 0] = 0
    gate[-1, -1, ::3] = 15
    ref = torch.full((B, Fout, Mp), 7.0, device=dev)
    got = torch.full((B, Fout, Mp), 7.0, device=dev)
    _lib.check(lib.chebgcn_contract_fwd(_P(stack), _P(W), None, 0, _P(ref), None, B, M, Fin, K, Fout, 1, 0, 0, _stream()), 'fwd')
    assert _lib.last_dispatch() == 'contract_fwd_ring_kernel'
    _lib.check(lib.chebgcn_contract_fwd_gated(_P(stack), _P(W), _P(gate), _P(got), B, M, Fin, K, Fout, _stream()), 'gated')
    assert _lib.last_dispatch() == 'contract_fwd_ring_kernel<gated>'
    bits = torch.stack([(gate >> r) & 1 for r in range(4)], dim=-1).reshape(B, Fout, Mp).bool()
    # (whole padded planes: both kernels store every quad of a plane, the pad's values come from the stack's pad)
    assert torch.equal(got, torch.where(bits, ref, torch.zeros_like(ref)))
    with pytest.raises(RuntimeError):
        _lib.check(lib.chebgcn_contract_fwd_gated(_P(stack), _P(W), _P(gate), _P(got), 1, 300, Fin, K, Fout, _stream()), 'gated')


def test_gate_links_are_invisible_in_the_benchmark_network(dev, monkeypatch):
    """BASELINE configs[1] at batch 26 (the smallest launches the big-launch contraction kernels take at this size are 25
    windows) with the layers linked (default) and not: every gradient bit-identical -- the linked
    step stores dy of layers 2..5 from the epilogue of the layer above (`contract_fwd_ring_kernel<gated>`), their ReluGrad
    pass is the bias reduction alone; and the linked network captured as a HIP graph replays the eager step."""
    import bench
    from gcn_fmri_decoding_amd import models_gcn, ops
    Ls, _ = bench.load_graph(10000, 1, 0, 1, None)
    L = Ls[0]
    M = L.shape[0]
    F, K, p, Mfc, C, B = [32] * 6, [5] * 6, [1] * 6, [512, 256, 22], 15, 26

    def make():
        torch.manual_seed(5)
        return models_gcn.cgcnn({'device': dev}, [L] * 6, F, K, p, Mfc, filter='chebyshev5', brelu='b2relu', pool='mpool1',
                                initial='he', channel=C, regularization=5e-4, dropout=1, batch_size=B, verbose=False)
    a, b = make(), make()
    b.load_state_dict(a.state_dict())
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    x = a.as_internal(ops.plane_storage(torch.randn((B, M, C), generator=gen, device=dev)))
    labels = (torch.arange(B, device=dev) * 5) % 21
    names = {}
    for net, on in ((a, True), (b, False)):
        monkeypatch.setattr(ops, 'gate_links', on)
        ops.timers = ops.KernelTimers(every=1, by_dispatch=True)
        try:
            ops.timers.next_step()
            net.train_step(x, labels)
            names[on] = sorted(ops.timers.summary())
        finally:
            ops.timers = None
    gated = [n for n in names[True] if 'contract_fwd_ring_kernel<gated>' in n]
    assert gated and gated[0].startswith('contract_bwd_x'), names[True]
    assert not [n for n in names[False] if '<gated>' in n], names[False]
    # with the links the full ReluGrad pass (op brelu_pool_bwd: dy written) is gone from the step, the bias reductions remain
    assert not [n for n in names[True] if n.startswith('brelu_pool_bwd')], names[True]
    assert [n for n in names[False] if n.startswith('brelu_pool_bwd')], names[False]
    assert torch.equal(a._grad, b._grad)
    assert torch.equal(a._flat, b._flat)
    monkeypatch.setattr(ops, 'gate_links', True)
    twin = make()
    twin.load_state_dict(a.state_dict())
    twin._loss_ema = None if a._loss_ema is None else a._loss_ema.clone()
    twin.global_step = a.global_step
    twin._adam_m.copy_(a._adam_m)
    twin._adam_v.copy_(a._adam_v)
    a.enable_step_graph(True)
    for _ in range(3):
        la = a.train_step(x, labels)[1]
        lb = twin.train_step(x, labels)[1]
    torch.cuda.synchronize()
    assert a._sg is not None and torch.equal(a._flat, twin._flat) and float(la) == float(lb)


@pytest.mark.parametrize('B,M,F', [(7, 10466, 5), (9, 300, 6), (130, 1044, 32)])
@pytest.mark.parametrize('bias_kind', [1, 2])
def test_bias_grad_sum_vs_float64(dev, B, M, F, bias_kind):
    """chebgcn_brelu_pool_bwd(pool = 1, relu = 0, dy = NULL): the bias gradient as the plain sum of dout over the windows (per
    vertex and filter) or over windows and vertices (per filter) -- `bias_grad_sum_kernel`; pads of the planes poisoned."""
    from gcn_fmri_decoding_amd import _lib, ops
    lib = _lib.lib()
    Mp = ops.plane_stride(M)
    gen = torch.Generator(device=dev)
    gen.manual_seed(B + M)
    g = torch.randn((B, F, Mp), generator=gen, device=dev)
    g[:, :, M:] = float('nan')
    n = lib.chebgcn_brelu_pool_bwd_workspace(B, M, F, 1, bias_kind)
    ws = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    db = torch.full((F, Mp) if bias_kind == 2 else (F,), 3.0, device=dev)
    _lib.check(lib.chebgcn_brelu_pool_bwd(_P(g), None, None, None, _P(db), bias_kind, B, M, F, 1, 0, 0, _P(ws) if n else None, n,
                                          _stream()), 'bias sum')
    kind = 'CHEBGCN_BIAS_VERTEX' if bias_kind == 2 else 'CHEBGCN_BIAS_FILTER'
    assert _lib.last_dispatch().startswith('bias_grad_sum_kernel<%s,' % kind), _lib.last_dispatch()
    g64 = g[:, :, :M].double()
    if bias_kind == 2:
        ref = g64.sum(0)
        got = db[:, :M].double()
        assert torch.equal(db[:, M:], torch.full_like(db[:, M:], 3.0)) or not torch.isnan(db[:, M:]).any()
    else:
        ref = g64.sum((0, 2))
        got = db.double()
    err = float((got - ref).abs().max() / ref.abs().max())
    record_measured('bias_grad_sum[%d,%d,%d,%d]' % (B, M, F, bias_kind), rel=err)
    assert err <= 2e-6, err


@pytest.mark.parametrize('B,M,Fin,K,Fout', [(128, 376, 32, 10, 32), (128, 422, 32, 10, 32), (40, 1044, 32, 5, 24), (7, 260, 20, 4, 30)])
def test_bwd_w_relu_bias_merged_is_the_two_calls(dev, B, M, Fin, K, Fout):
    """chebgcn_contract_bwd_w_relu_bias (small launches: the per-vertex bias gradient in the launch that adds the weight gradient's
    partials) against chebgcn_contract_bwd_w_relu + chebgcn_brelu_pool_bwd(dout, NULL, mask, NULL, dbias): dW and dbias bit for
    bit (the same code and order); not served on a big launch."""
    from gcn_fmri_decoding_amd import _lib, ops
    lib = _lib.lib()
    assert lib.chebgcn_contract_bwd_w_relu_bias_merged(B, M, Fin, K, Fout) == 1
    assert lib.chebgcn_contract_bwd_w_relu_bias_merged(64, 10466, 32, 5, 32) == 0
    Mp = ops.plane_stride(M)
    gen = torch.Generator(device=dev)
    gen.manual_seed(B + M + Fin)
    stack = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    stack[:, :, :, M:] = 0
    dout = torch.randn((B, Fout, Mp), generator=gen, device=dev)
    dout[:, :, M:] = float('nan')                            # the pad of a plane carries no gradient: masked out below, never summed
    mask = torch.randint(0, 16, (B, Fout, Mp // 4), generator=gen, device=dev, dtype=torch.uint8)
    q0 = M // 4
    if M % 4:
        mask[:, :, q0] &= (1 << (M % 4)) - 1                 # (the forward never sets bits of pad vertices)
    mask[:, :, q0 + (1 if M % 4 else 0):] = 0
    nws = lib.chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    dW1, dW2 = torch.full((Fin * K, Fout), 5.0, device=dev), torch.full((Fin * K, Fout), 6.0, device=dev)
    db1, db2 = torch.full((Fout, Mp), 5.0, device=dev), torch.full((Fout, Mp), 6.0, device=dev)
    _lib.check(lib.chebgcn_contract_bwd_w_relu(_P(stack), _P(dout), _P(mask), _P(dW1), _P(ws), nws, B, M, Fin, K, Fout, _stream()), 'bwd_w')
    assert _lib.last_dispatch().endswith('reduce_partials_small'), _lib.last_dispatch()
    _lib.check(lib.chebgcn_brelu_pool_bwd(_P(dout), None, _P(mask), None, _P(db1), 2, B, M, Fout, 1, 0, 1, None, 0, _stream()), 'bias')
    assert _lib.last_dispatch() == 'bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX,16>', _lib.last_dispatch()
    _lib.check(lib.chebgcn_contract_bwd_w_relu_bias(_P(stack), _P(dout), _P(mask), _P(dW2), _P(db2), _P(ws), nws, B, M, Fin, K, Fout,
                                                    _stream()), 'merged')
    assert _lib.last_dispatch().endswith('reduce_partials_small_bias_kernel'), _lib.last_dispatch()
    assert torch.equal(dW1, dW2)
    assert torch.equal(db1[:, :M], db2[:, :M]) and torch.isfinite(db2[:, :M]).all()
    with pytest.raises(RuntimeError):
        _lib.check(lib.chebgcn_contract_bwd_w_relu_bias(_P(stack), _P(dout), _P(mask), _P(dW2), _P(db2), _P(ws), nws, 64, 10466, 32, 5,
                                                        32, _stream()), 'merged big')
