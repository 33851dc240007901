"""Parity of the HIP path (through the C ABI of libchebgcn.so) with the CPU oracle and the
golden vectors produced by the reference.  Needs an MI355X: run with ``-m gpu``.

Tolerance: BASELINE.json's north star asks for <= 1e-5 relative fp32; we assert
max|hip - ref| <= 1e-5 * max|ref| per tensor (REL below), bit-exactness for index maps.
"""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import assert_adam_params_close, csr_from, load_golden
from oracle import coarsening_ref as CR
from oracle import graph_ref as GR
from oracle import layers_ref as R

pytestmark = pytest.mark.gpu
REL = 1e-5


def close(got, ref, rel=REL, what=''):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max() / scale
    assert err <= rel, '%s: rel err %.3e > %.1e' % (what, err, rel)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops():
    from gcn_fmri_decoding_amd import ops as o
    return o


def to_storage(ops, x_bmf, dev):
    """numpy [B, M, F] -> plane storage tensor [B, F, Mp] with a NaN-poisoned pad (the pad
    must never leak into results)."""
    B, M, F = x_bmf.shape
    st = torch.full((B, F, ops.plane_stride(M)), float('nan'), device=dev)
    st[:, :, :M] = torch.as_tensor(np.ascontiguousarray(x_bmf.transpose(0, 2, 1))).to(dev)
    return st


def from_storage(st, M):
    return st[:, :, :M].permute(0, 2, 1).cpu().numpy()


def stack_to_ref(stack, M):
    """[K, B, F, Mp] device -> oracle layout T[K, M, F, B]."""
    return stack[:, :, :, :M].permute(0, 3, 2, 1).cpu().numpy()


def levels(name='layers_n212'):
    z = load_golden(name)
    return [csr_from(z, 'L%d' % i) for i in range(4)]


# ---------------------------------------------------------------------------------------
# recurrence
# ---------------------------------------------------------------------------------------

@pytest.mark.parametrize('lvl,B,Fin,K', [(0, 1, 1, 2), (0, 3, 5, 5), (1, 2, 3, 9), (0, 2, 1, 25), (3, 5, 7, 4), (0, 4, 15, 1)])
def test_recurrence_fwd_vs_oracle(ops, dev, lvl, B, Fin, K):
    from gcn_fmri_decoding_amd import _lib
    L = levels()[lvl]
    M = L.shape[0]
    g = ops.Graph(L, dev)
    rs = np.random.RandomState(lvl * 100 + K)
    x = rs.randn(B, M, Fin).astype(np.float32)
    xs = to_storage(ops, x, dev)
    stack = torch.full((K, B, Fin, g.Mp), float('nan'), device=dev)
    _lib.check(_lib.lib().chebgcn_recurrence_fwd(g.handle, ops._p(xs), ops._p(stack), B, Fin, K, ops._stream()), 'fwd')
    ref = R.cheb_stack(R.rescaled_laplacian(L, np.float32), x, K)        # [K, M, Fin, B]
    close(stack_to_ref(stack, M), ref, what='stack')
    # in-place form: x already is slab 0
    stack2 = torch.empty((K, B, Fin, g.Mp), device=dev)
    stack2[0] = xs
    _lib.check(_lib.lib().chebgcn_recurrence_fwd(g.handle, ops._p(stack2), ops._p(stack2), B, Fin, K, ops._stream()), 'fwd')
    assert torch.equal(stack2[:, :, :, :M], stack[:, :, :, :M])


@pytest.mark.parametrize('lvl,B,Fin,K', [(0, 1, 1, 2), (0, 3, 5, 5), (1, 2, 3, 9), (2, 3, 2, 3), (0, 2, 2, 1)])
def test_recurrence_bwd_vs_oracle(ops, dev, lvl, B, Fin, K):
    from gcn_fmri_decoding_amd import _lib
    L = levels()[lvl]
    M = L.shape[0]
    g = ops.Graph(L, dev)
    rs = np.random.RandomState(7 + K)
    G = rs.randn(K, M, Fin * B).astype(np.float32)                       # oracle layout [K, M, Fin*B]
    Lt = sp.csr_matrix(R.rescaled_laplacian(L, np.float32).T)
    ref = G.copy()
    for k in range(K - 1, 1, -1):
        ref[k - 1] += 2 * Lt.dot(ref[k])
        ref[k - 2] -= ref[k]
    if K > 1:
        ref[0] += Lt.dot(ref[1])
    gs = torch.full((K, B, Fin, g.Mp), float('nan'), device=dev)
    gs[:, :, :, :M] = torch.as_tensor(G.reshape(K, M, Fin, B).transpose(0, 3, 2, 1).copy()).to(dev)
    dx = torch.full((B, Fin, g.Mp), float('nan'), device=dev)
    _lib.check(_lib.lib().chebgcn_recurrence_bwd(g.handle, ops._p(gs), ops._p(dx), B, Fin, K, ops._stream()), 'bwd')
    got = dx[:, :, :M].permute(2, 1, 0).cpu().numpy().reshape(M, Fin * B)
    close(got, ref[0], what='dx')


def test_recurrence_adjoint_identity(ops, dev):
    """<stack(x), G> == <x, adjoint(G)> at the benchmark size (size-independent property)."""
    from gcn_fmri_decoding_amd import _lib, graph
    Ls, perm, _ = graph.synthetic_graph(10000, k=8, levels=1)
    g = ops.Graph(Ls[0], dev)
    assert g.M == 10466 and g.on_chip
    B, Fin, K = 4, 6, 5
    torch.manual_seed(0)
    x = torch.randn(B, Fin, g.Mp, device=dev)
    x[:, :, g.M:] = 0
    G = torch.randn(K, B, Fin, g.Mp, device=dev)
    G[:, :, :, g.M:] = 0
    stack = torch.empty_like(G)
    dx = torch.empty_like(x)
    lib = _lib.lib()
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, ops._p(x), ops._p(stack), B, Fin, K, ops._stream()), 'fwd')
    _lib.check(lib.chebgcn_recurrence_bwd(g.handle, ops._p(G), ops._p(dx), B, Fin, K, ops._stream()), 'bwd')
    lhs = (stack[:, :, :, :g.M].double() * G[:, :, :, :g.M].double()).sum().item()
    rhs = (x[:, :, :g.M].double() * dx[:, :, :g.M].double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0)
    # against the oracle on two planes of the full-size graph
    Lr = R.rescaled_laplacian(Ls[0], np.float32)
    xp = x[1, 2:4, :g.M].cpu().numpy().T.copy()                          # [M, 2]
    ref = GR.chebyshev(Lr, xp, K)                                          # [K, M, 2]
    got = stack[:, 1, 2:4, :g.M].permute(0, 2, 1).cpu().numpy()
    close(got, ref, what='full-size stack')
    # linearity
    y = torch.randn_like(x)
    y[:, :, g.M:] = 0
    s2 = torch.empty_like(G)
    s3 = torch.empty_like(G)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, ops._p(y), ops._p(s2), B, Fin, K, ops._stream()), 'fwd')
    z = (2.0 * x - y).contiguous()
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, ops._p(z), ops._p(s3), B, Fin, K, ops._stream()), 'fwd')
    lin = 2.0 * stack - s2
    err = (s3 - lin)[:, :, :, :g.M].abs().max().item() / lin[:, :, :, :g.M].abs().max().item()
    assert err <= 1e-5


# ---------------------------------------------------------------------------------------
# one layer, forward and backward, against the oracle / reference vectors
# ---------------------------------------------------------------------------------------

def test_chebyshev5_layer_vs_reference_vectors(ops, dev):
    """chebyshev5 (+ b1relu / b2relu + mpool1 / apool1) on the golden inputs: compared with
    the reference's own layer methods (executed under the TF stand-in)."""
    z = load_golden('layers_n212')
    Ls = levels()
    for tag in 'abcde':
        x, W = z['cheb_%s_x' % tag], z['cheb_%s_W' % tag]
        K, lvl = int(z['cheb_%s_K' % tag]), int(z['cheb_%s_lvl' % tag])
        L = Ls[lvl]
        M = L.shape[0]
        g = ops.Graph(L, dev)
        xs = to_storage(ops, x, dev)
        Wd = torch.as_tensor(W).to(dev)
        y = ops.cheb_conv(xs, Wd, None, g, K)
        close(from_storage(y, M), z['cheb_%s_y' % tag], what='cheb ' + tag)
        b1 = torch.as_tensor(z['cheb_%s_b1' % tag].reshape(-1)).to(dev)
        b2 = torch.zeros((W.shape[1], g.Mp), device=dev)
        b2[:, :M] = torch.as_tensor(z['cheb_%s_b2' % tag][0].T.copy()).to(dev)
        y1 = ops.cheb_conv(xs, Wd, b1, g, K, relu=True, bias_kind=ops.BIAS_FILTER)
        close(from_storage(y1, M), z['cheb_%s_y1' % tag], what='b1relu ' + tag)
        for p in (1, 2, 4):
            ymp = ops.cheb_conv(xs, Wd, b2, g, K, pool=p, pool_kind=ops.POOL_MAX, relu=True, bias_kind=ops.BIAS_VERTEX)
            close(from_storage(ymp, M // p), z['cheb_%s_mp%d' % (tag, p)], what='mpool %s %d' % (tag, p))
            yap = ops.cheb_conv(xs, Wd, b2, g, K, pool=p, pool_kind=ops.POOL_AVG, relu=True, bias_kind=ops.BIAS_VERTEX)
            close(from_storage(yap, M // p), z['cheb_%s_ap%d' % (tag, p)], what='apool %s %d' % (tag, p))
            # unfused kernels give the same numbers (the same contraction kernel with and without its epilogue: the
            # on-chip layer of csrc/fused_small.hip, which serves p == 1 on graphs of this size, sums in another order)
            fs, ops.fused_small = ops.fused_small, False
            try:
                y2 = ops.cheb_conv(xs, Wd, None, g, K)
                if p == 1:
                    ymp = ops.cheb_conv(xs, Wd, b2, g, K, pool=p, pool_kind=ops.POOL_MAX, relu=True, bias_kind=ops.BIAS_VERTEX)
            finally:
                ops.fused_small = fs
            yu = ops.BiasReluPool.apply(y2, b2, M, p, ops.POOL_MAX, True, ops.BIAS_VERTEX)
            assert torch.equal(yu[:, :, :M // p], ymp[:, :, :M // p])


@pytest.mark.parametrize('lvl,B,Fin,Fout,K,p,pool_kind,bias', [
    (0, 3, 5, 8, 5, 1, 0, 2), (0, 2, 3, 40, 3, 2, 0, 1), (1, 2, 4, 33, 2, 4, 0, 2), (0, 2, 2, 4, 1, 1, 0, 1),
    (0, 2, 6, 70, 4, 2, 1, 2), (0, 3, 3, 5, 3, 8, 0, 2), (0, 1, 33, 32, 2, 1, 0, 0), (1, 2, 4, 6, 3, 2, 1, 1)])
def test_layer_gradients_vs_oracle(ops, dev, lvl, B, Fin, Fout, K, p, pool_kind, bias):
    L = levels()[lvl]
    M = L.shape[0]
    g = ops.Graph(L, dev)
    rs = np.random.RandomState(11 * lvl + Fout)
    x = rs.randn(B, M, Fin).astype(np.float32)
    W = (rs.randn(Fin * K, Fout) * 0.3).astype(np.float32)
    if bias == 1:
        b = (rs.randn(1, 1, Fout) * 0.5).astype(np.float32)
    elif bias == 2:
        b = (rs.randn(1, M, Fout) * 0.5).astype(np.float32)
    else:
        b = np.zeros((1, 1, Fout), np.float32)
    relu = bias != 0
    # oracle
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
    dx_ref, dW_ref = R.chebyshev5_bwd(dy.astype(np.float32), L, W, K, T)
    # HIP
    xs = to_storage(ops, x, dev).requires_grad_(True)
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
    close(from_storage(out.detach(), M // p), o, what='out')
    gout = torch.full(out.shape, float('nan'), device=dev)
    gout[:, :, :M // p] = torch.as_tensor(np.ascontiguousarray(do.transpose(0, 2, 1))).to(dev)
    out.backward(gout)
    close(from_storage(xs.grad, M), dx_ref, what='dx', rel=2e-5)
    close(Wd.grad.cpu().numpy(), dW_ref, what='dW', rel=2e-5)
    if bias == 1:
        close(bd.grad.cpu().numpy(), db.reshape(-1), what='db1', rel=2e-5)
    elif bias == 2:
        close(bd.grad[:, :M].t().cpu().numpy(), db[0], what='db2')
        assert float(bd.grad[:, M:].abs().sum()) == 0.0


@pytest.mark.parametrize('lvl,B,Fin,Fout,K,bias', [
    (0, 3, 5, 32, 3, 2), (0, 2, 32, 32, 5, 2), (1, 2, 4, 40, 2, 1), (0, 1, 7, 70, 4, 0), (2, 5, 3, 9, 3, 2)])
def test_relu_grad_fold_is_bit_identical(ops, dev, lvl, B, Fin, Fout, K, bias):
    """pool == 1 layers with ReLU: the gradients computed by chebgcn_contract_bwd_w_relu /
    _bwd_x_relu from (gout, ReLU bit mask) are the SAME numbers as the separate ReluGrad pass
    (chebgcn_brelu_pool_bwd -> dy) followed by the plain gradients: gating a product's operand to
    zero in registers or in memory feeds the matrix cores identical values."""
    L = levels()[lvl]
    M = L.shape[0]
    g = ops.Graph(L, dev)
    rs = np.random.RandomState(17 + Fout)
    x = rs.randn(B, M, Fin).astype(np.float32)
    W = torch.as_tensor((rs.randn(Fin * K, Fout) * 0.3).astype(np.float32)).to(dev)
    if bias == 1:
        b, kind = torch.as_tensor((rs.randn(Fout) * 0.5).astype(np.float32)).to(dev), ops.BIAS_FILTER
    elif bias == 2:
        b, kind = torch.zeros((Fout, g.Mp), device=dev), ops.BIAS_VERTEX
        b[:, :M] = torch.as_tensor((rs.randn(Fout, M) * 0.5).astype(np.float32)).to(dev)
    else:
        b, kind = None, ops.BIAS_NONE
    gout = torch.zeros((B, Fout, g.Mp), device=dev)
    gout[:, :, :M] = torch.as_tensor(rs.randn(B, Fout, M).astype(np.float32)).to(dev)
    res = {}
    try:
        for fold in (False, True):
            ops.fold_relu_grad = fold
            xs = to_storage(ops, x, dev).requires_grad_(True)
            Wp = W.clone().requires_grad_(True)
            bp = b.clone().requires_grad_(True) if b is not None else None
            out = ops.cheb_conv(xs, Wp, bp, g, K, relu=True, bias_kind=kind)
            out.backward(gout)
            res[fold] = (out.detach()[:, :, :M], xs.grad[:, :, :M], Wp.grad, None if bp is None else
                         (bp.grad[:, :M] if bias == 2 else bp.grad))
    finally:
        ops.fold_relu_grad = True
    for name, a, c in zip(('out', 'dx', 'dW', 'dbias'), res[False], res[True]):
        if a is None:
            continue
        if name == 'dbias':
            # the two bias reductions split the batch differently (and the per-filter one is an atomic sum over
            # workgroups): same terms, another fp32 summation order
            close(c.cpu().numpy(), a.cpu().numpy(), what=name)
        else:
            assert torch.equal(a, c), '%s differs between the folded and the separate ReluGrad' % name
    # and against the oracle (the fold is what every other layer test runs through as well)
    y = R.chebyshev5_fwd(x, L, W.cpu().numpy(), K)
    if bias == 2:
        bb = b[:, :M].cpu().numpy().T[None]
    elif bias == 1:
        bb = b.cpu().numpy().reshape(1, 1, Fout)
    else:
        bb = np.zeros((1, 1, Fout), np.float32)
    a_ref = R.brelu_fwd(y, bb)
    close(from_storage(res[True][0], M), a_ref, what='out')


# ---------------------------------------------------------------------------------------
# bf16 matrix-core contraction (BASELINE config 5).  Tolerances (relative to max|ref|): one pass
# rounds both operands to bf16 (2^-9 each) -> 1e-2; three passes keep hi*hi + hi*lo + lo*hi of
# the two-bf16 split of each operand (dropped lo*lo and the rounding of lo: ~2^-16 per product)
# -> 5e-5, i.e. fp32-grade.
# ---------------------------------------------------------------------------------------
BF16_REL = {'bf16': 1e-2, 'bf16x3': 5e-5}


@pytest.mark.parametrize('precision', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('lvl,B,Fin,Fout,K,p,pool_kind,bias', [
    (0, 2, 12, 256, 5, 1, 0, 2), (0, 2, 3, 40, 3, 2, 0, 1), (1, 2, 4, 33, 2, 4, 0, 2), (0, 1, 7, 300, 5, 1, 0, 0),
    (0, 2, 6, 70, 4, 2, 1, 2), (0, 3, 3, 5, 3, 8, 0, 2), (0, 1, 16, 64, 1, 1, 0, 1)])
def test_bf16_contraction_vs_oracle(ops, dev, precision, lvl, B, Fin, Fout, K, p, pool_kind, bias):
    L = levels()[lvl]
    M = L.shape[0]
    g = ops.Graph(L, dev)
    rs = np.random.RandomState(5 * lvl + Fout)
    x = rs.randn(B, M, Fin).astype(np.float32)
    W = (rs.randn(Fin * K, Fout) * 0.3).astype(np.float32)
    if bias == 1:
        b = (rs.randn(1, 1, Fout) * 0.5).astype(np.float32)
    elif bias == 2:
        b = (rs.randn(1, M, Fout) * 0.5).astype(np.float32)
    else:
        b = np.zeros((1, 1, Fout), np.float32)
    relu = bias != 0
    y = R.chebyshev5_fwd(x, L, W, K)
    a = R.brelu_fwd(y, b) if relu else y + b
    # pooling picks may flip under rounding; values still agree to the bound
    o = R.mpool1_fwd(a, p)[0] if pool_kind == 0 else R.apool1_fwd(a, p)
    xs = to_storage(ops, x, dev)
    Wd = torch.as_tensor(W).to(dev)
    if bias == 1:
        bd, kind = torch.as_tensor(b.reshape(-1)).to(dev), ops.BIAS_FILTER
    elif bias == 2:
        bd = torch.zeros((Fout, g.Mp), device=dev)
        bd[:, :M] = torch.as_tensor(b[0].T.copy()).to(dev)
        kind = ops.BIAS_VERTEX
    else:
        bd, kind = None, ops.BIAS_NONE
    out = ops.cheb_conv(xs, Wd, bd, g, K, pool=p, pool_kind=pool_kind, relu=relu, bias_kind=kind, precision=precision)
    # scale of the pre-activation sums: the rounding error does not shrink with the ReLU / bias
    scale = np.abs(y).max()
    err = np.abs(from_storage(out, M // p).astype(np.float64) - o).max() / scale
    assert err <= BF16_REL[precision], '%s: rel err %.3e' % (precision, err)
    if precision == 'bf16':
        assert err > 1e-6, 'bf16 path suspiciously exact: is it running the fp32 kernel?'
    # (the gradients of a mixed-precision layer have their own value tests: test_bf16_contraction_gradients,
    # test_bf16_layer_gradients_vs_fp32_layer)


def test_bf16_contraction_config5_shape(ops, dev):
    """BASELINE config 5 (block_dura 60 -> Fin = 60, Fout = 256, K = 5) on the benchmark graph, two windows: the
    forward contraction in fp32 MFMA, bf16 and split bf16 against a FLOAT64 product of the same operands (the
    Chebyshev stack the recurrence kernel left, W, the per-vertex bias) computed on the device -- not against
    another kernel of this library."""
    import bench
    from gcn_fmri_decoding_amd import _lib
    Ls, _ = bench.load_graph(10000, 1, 0, 1, None)
    g = ops.Graph(Ls[0], dev)
    B, Fin, Fout, K, M = 2, 60, 256, 5, g.M
    torch.manual_seed(3)
    x = torch.randn(B, Fin, g.Mp, device=dev)
    x[:, :, M:] = 0
    W = torch.randn(Fin * K, Fout, device=dev) * 0.1
    bias = torch.randn(Fout, g.Mp, device=dev) * 0.5
    stack = torch.empty((K, B, Fin, g.Mp), device=dev)
    _lib.check(_lib.lib().chebgcn_recurrence_fwd(g.handle, ops._p(x), ops._p(stack), B, Fin, K, ops._stream()), 'recurrence_fwd')
    # pre[b, o, m] = sum_{fin, k} W[fin*K + k, o] * stack[k, b, fin, m]   (models_gcn.py:611-617), float64
    pre = torch.einsum('fko,kbfm->bom', W.double().view(Fin, K, Fout), stack[..., :M].double())
    ref = torch.relu(pre + bias[:, :M].double())
    scale = float(pre.abs().max())
    for precision, rel in [('f32', 1e-5)] + list(BF16_REL.items()):
        got = ops.cheb_conv(x, W, bias, g, K, relu=True, bias_kind=ops.BIAS_VERTEX, precision=precision)[:, :, :M].double()
        err = float((got - ref).abs().max()) / scale
        assert err <= rel, '%s: rel err %.3e' % (precision, err)


@pytest.mark.parametrize('precision', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('B,M,Fin,K,Fout', [
    (2, 10466, 60, 5, 256),      # BASELINE config 5 on the benchmark graph: the one-pass wide bwd_w, bwd_x on five waves
    (64, 10466, 60, 5, 256),     # ... and at the batch bench.py's config5 object times (41920 chunks over 256 workgroups)
    (3, 200, 3, 3, 40),          # one ragged row tile, two ragged column tiles
    (2, 1000, 7, 5, 300),        # Fout beyond one 256-filter group of bwd_x; RT = 2
    (1, 64, 16, 1, 16),          # K = 1, a single chunk, CT = 1
    (5, 333, 13, 4, 33),         # RT = 2 (52 rows), ragged everything
    (2, 96, 40, 5, 64),          # RT = 5 with two row groups (200 rows), CT = 2
    (2, 500, 70, 5, 300),        # one-pass wide kernel with 2 x 2 workgroup tiles (350 rows, 300 columns)
    (3, 77, 33, 5, 65),          # wide kernel, ragged tiles, chunks of 16 vertices with a tail of 13
    (70, 40, 60, 5, 256),        # wide kernel with more windows than vertices chunks per window
    (2, 300, 120, 5, 64),        # 600 rows: bwd_x on five waves with two groups of 320; bwd_w with four row groups
    (1, 130, 20, 5, 96)])        # 100 rows (one group of 256 on four waves), 96 reduction rows = 6 k-steps
def test_bf16_contraction_gradients(ops, dev, precision, B, M, Fin, K, Fout):
    """chebgcn_contract_bwd_x_bf16 / chebgcn_contract_bwd_w_bf16 (the MatMul gradients of
    models_gcn.py:616 on the bf16 matrix cores) against float64 sums of the same operands.  The
    padding of the planes is filled with huge values: no gradient may see it."""
    from gcn_fmri_decoding_amd import _lib
    lib = _lib.lib()
    Mp = (M + 31) // 32 * 32
    passes = ops.PRECISIONS[precision]
    gen = torch.Generator(device='cpu').manual_seed(B * 1000 + M + Fout)
    stack = torch.randn(K, B, Fin, Mp, generator=gen)
    dy = torch.randn(B, Fout, Mp, generator=gen)
    W = torch.randn(Fin * K, Fout, generator=gen) * 0.2
    stack[..., M:] = 1e30
    dy[..., M:] = -1e30
    sd, dd, Wd = stack.to(dev), dy.to(dev), W.to(dev)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- bwd_w
    n = lib.chebgcn_contract_bwd_w_bf16_workspace(B, M, Fin, K, Fout)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    dW = torch.full((Fin * K, Fout), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_w_bf16(ptr(sd), ptr(dd), ptr(dW), ptr(ws), n, B, M, Fin, K, Fout, passes, stream), 'bwd_w_bf16')
    s64 = stack[..., :M].double().permute(2, 0, 1, 3).reshape(Fin * K, B * M)            # rows fin*K + k
    d64 = dy[..., :M].double().permute(1, 0, 2).reshape(Fout, B * M)
    ref = s64 @ d64.T
    err = float((dW.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err <= BF16_REL[precision], 'bwd_w %s: rel err %.3e' % (precision, err)
    if precision == 'bf16':
        assert err > 1e-6, 'suspiciously exact: is the fp32 kernel running?'
    dW2 = torch.empty_like(dW)
    _lib.check(lib.chebgcn_contract_bwd_w_bf16(ptr(sd), ptr(dd), ptr(dW2), ptr(ws), n, B, M, Fin, K, Fout, passes, stream), 'bwd_w_bf16')
    assert torch.equal(dW, dW2), 'bwd_w_bf16 is not deterministic'

    # ---- bwd_x
    n = lib.chebgcn_contract_bwd_x_bf16_workspace(Fin, K, Fout)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    gs = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_x_bf16(ptr(dd), ptr(Wd), ptr(gs), B, M, Fin, K, Fout, passes, ptr(ws), n, stream), 'bwd_x_bf16')
    refx = torch.einsum('ro,bom->rbm', W.double(), dy[..., :M].double())                  # [Fin*K, B, M]
    refx = refx.reshape(Fin, K, B, M).permute(1, 2, 0, 3)                                # [K, B, Fin, M]
    got = gs[..., :M].cpu().double()
    err = float((got - refx).abs().max() / refx.abs().max())
    assert err <= BF16_REL[precision], 'bwd_x %s: rel err %.3e' % (precision, err)


@pytest.mark.parametrize('precision', ['bf16', 'bf16x3'])
def test_bf16_layer_gradients_vs_fp32_layer(ops, dev, precision):
    """A whole layer (recurrence + bf16 contraction, no ReLU / pooling so that no pick can flip):
    dx, dW and dbias of the mixed-precision layer against the fp32 layer on the same inputs."""
    L = levels()[0]
    M = L.shape[0]
    g = ops.Graph(L, dev)
    B, Fin, K, Fout = 3, 12, 5, 72
    rs = np.random.RandomState(11)
    x = rs.randn(B, M, Fin).astype(np.float32)
    W = torch.as_tensor((rs.randn(Fin * K, Fout) * 0.3).astype(np.float32)).to(dev)
    b = torch.as_tensor((rs.randn(Fout) * 0.5).astype(np.float32)).to(dev)
    gout = None
    grads = {}
    for prec in ('f32', precision):
        xs = to_storage(ops, x, dev).requires_grad_(True)
        Wp, bp = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
        out = ops.cheb_conv(xs, Wp, bp, g, K, bias_kind=ops.BIAS_FILTER, precision=prec)
        if gout is None:
            gout = torch.zeros_like(out)
            gout[:, :, :M] = torch.randn(B, Fout, M, device=dev)
        out.backward(gout)
        grads[prec] = (xs.grad[:, :, :M].double(), Wp.grad.double(), bp.grad.double())
    for name, ref, got in zip(('dx', 'dW', 'dbias'), grads['f32'], grads[precision]):
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err <= BF16_REL[precision], '%s %s: rel err %.3e' % (name, precision, err)


# ---------------------------------------------------------------------------------------
# staging, head, optimizer
# ---------------------------------------------------------------------------------------

def test_perm_data_bit_exact(ops, dev):
    z = load_golden('coarsen_n212')
    x3, perm = z['pd_x3'], z['perm']
    ref = z['pd_y3']                                                      # float64 [S, M, F]
    xd = torch.as_tensor(x3).to(dev)
    pd = torch.as_tensor(perm.astype(np.int32)).to(dev)
    out = ops.perm_data(xd, pd)
    M = len(perm)
    assert np.array_equal(from_storage(out, M).astype(np.float64), ref)
    assert float(out[:, :, M:].abs().sum()) == 0.0
    smp = torch.as_tensor(np.array([2, 0, 2], np.int32)).to(dev)
    out2 = ops.perm_data(xd, pd, smp)
    assert np.array_equal(from_storage(out2, M).astype(np.float64), ref[[2, 0, 2]])
    # layout round trip
    back = ops.from_plane(out, M)
    assert np.array_equal(back.cpu().numpy().astype(np.float64), ref)
    assert torch.equal(ops.plane_storage(back)[:, :, :M], out[:, :, :M])
    assert ops.plane_storage(ops.plane_view(out, M)).data_ptr() == out.data_ptr()


def test_feature_mean_and_adam(ops, dev):
    rs = np.random.RandomState(3)
    x = rs.randn(3, 50, 7).astype(np.float32)
    xs = to_storage(ops, x, dev).requires_grad_(True)
    y = ops.FeatureMean.apply(xs, 50)
    close(y.detach().cpu().numpy(), x.mean(-1), what='mean')
    gy = rs.randn(3, 50).astype(np.float32)
    y.backward(torch.as_tensor(gy).to(dev))
    close(from_storage(xs.grad, 50), np.repeat(gy[:, :, None] / 7, 7, axis=2), what='dmean')
    # Adam (TF form) with L2 folded into the gradient
    p0 = rs.randn(1000).astype(np.float32)
    params, state = {'w': p0.copy()}, {}
    p = torch.as_tensor(p0.copy()).to(dev)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for t in range(1, 4):
        g = rs.randn(1000).astype(np.float32)
        R.adam_tf_step(params, {'w': 0.5 * g + 5e-4 * params['w']}, state)
        lr_t = 0.001 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.adam_step(p, torch.as_tensor(g).to(dev), m, v, lr_t, grad_scale=0.5, l2=5e-4)
        close(p.cpu().numpy(), params['w'], what='adam step %d' % t)


# ---------------------------------------------------------------------------------------
# whole network through the cgcnn drop-in
# ---------------------------------------------------------------------------------------

def build_model(z, dev, **kw):
    from gcn_fmri_decoding_amd import models_gcn
    Ls = [csr_from(z, 'L%d' % i) for i in range(int(z['nlevels']))]
    net = models_gcn.cgcnn({'device': dev}, Ls, z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist(),
                           channel=int(z['channel']), brelu=str(z['brelu']), batch_size=int(z['x'].shape[0]),
                           verbose=False, **kw)
    params = {k[len('param:'):]: z[k] for k in z.files if k.startswith('param:')}
    for k, v in params.items():
        net.set_variable(k, v)
    return net, Ls, params


@pytest.mark.parametrize('name', ['inference_pool_n212', 'inference_flat_n212', 'inference_config1_n512', 'inference_pool6_n512'])
def test_inference_vs_reference_vectors(ops, dev, name):
    z = load_golden(name)
    net, Ls, params = build_model(z, dev)
    assert {k: tuple(net.variable(k).shape) for k in net.variables()} == {k: v.shape for k, v in params.items()}
    x = torch.as_tensor(z['x']).to(dev)
    with torch.no_grad():
        logits = net.inference(x, 1)
    close(logits.cpu().numpy(), z['logits'], rel=2e-5, what='logits')
    # string-dispatched, unfused layer methods (a subclass overriding one of them switches
    # the fused fast path off) give the same logits
    from gcn_fmri_decoding_amd import models_gcn

    class Unfused(models_gcn.cgcnn):
        def mpool1(self, x, p):
            return super().mpool1(x, p)

    net2 = Unfused({'device': dev}, Ls, z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist(),
                   channel=int(z['channel']), brelu=str(z['brelu']), batch_size=int(z['x'].shape[0]), verbose=False)
    assert net._fusable() and not net2._fusable()
    for k, v in params.items():
        net2.set_variable(k, v)
    with torch.no_grad():
        logits2 = net2.inference(x, 1)
    close(logits2.cpu().numpy(), logits.cpu().numpy(), rel=1e-6, what='unfused logits')


@pytest.mark.parametrize('name', ['inference_pool_n212', 'inference_flat_n212', 'inference_pool6_n512', 'inference_config1_n512'])
def test_train_step_vs_oracle(ops, dev, name):
    """Gradients of the full network and three TF-form Adam steps against the oracle (``inference_config1_n512`` =
    BASELINE configs[0]: 1stGCN K=1, N=512, block_dura=1, batch 4 -- "loss, one Adam step")."""
    z = load_golden(name)
    reg = 5e-4
    net, Ls, params = build_model(z, dev, regularization=reg, dropout=1)
    onet = R.Net(Ls, z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist(), channel=int(z['channel']),
                 brelu=str(z['brelu']), regularization=reg)
    params = {k: v.copy() for k, v in params.items()}
    x = z['x']
    labels = np.arange(x.shape[0]) % int(z['M'][-1])
    xs = to_storage(ops, x, dev)
    ld = torch.as_tensor(labels).to(dev)
    state, ill = {}, {}
    for step in range(3):
        logits, cache = onet.forward(params, x)
        loss, dlogits = onet.loss(params, logits, labels)
        grads = onet.backward(params, cache, dlogits)
        _, loss_avg = net.train_step(xs, ld)
        if step == 0:
            for k in params:                                   # net._grad holds d(CE); add the L2 part
                ref = grads[k] - (reg * params[k] if onet.regularized(k) else 0)
                got = net_grad_in_ref_shape(net, k)
                close(got, ref, rel=5e-5, what='grad ' + k)
            assert abs(float(loss_avg) - 0.1 * loss) <= 2e-5 * abs(0.1 * loss)      # (first read of the zero-initialised 0.9-EMA)
        R.adam_tf_step(params, grads, state)
        for k in params:
            assert_adam_params_close(net.get_var(k), params[k], state['v/' + k], step, ill, k, rel=2e-5 if step == 0 else 1e-4, quantile=1.0 if step == 0 else 0.999)


def test_second_stream_for_bwd_w_is_bit_identical(ops, dev):
    """ops.overlap_bwd_w: contract_bwd_w on a second stream beside contract_bwd_x / recurrence_bwd.
    Three training steps must leave bit-identical variables."""
    z = load_golden('inference_flat_n212')
    x = to_storage(ops, z['x'], dev)
    labels = torch.as_tensor(np.arange(z['x'].shape[0]) % int(z['M'][-1])).to(dev)
    results = []
    before = ops.overlap_bwd_w                 # ('auto' since round 6: the second stream for wide layers only)
    for flag in (False, True):
        ops.overlap_bwd_w = flag
        try:
            net, _, params = build_model(z, dev, regularization=5e-4, dropout=1)
            for _ in range(3):
                net.train_step(x, labels)
            torch.cuda.synchronize()
            results.append({k: net.get_var(k).copy() for k in params})
        finally:
            ops.overlap_bwd_w = before
    for k in results[0]:
        assert np.array_equal(results[0][k], results[1][k]), k


def net_grad_in_ref_shape(net, name):
    return net.gradient(name).cpu().numpy()


def test_predict_evaluate_fit_smoke(ops, dev, tmp_path, monkeypatch):
    """fit / evaluate / predict call contract (models_gcn.py:31-184) on a tiny problem."""
    monkeypatch.setenv('CHEBGCN_HOME', str(tmp_path))
    z = load_golden('inference_pool_n212')
    net, Ls, _ = build_model(z, dev, num_epochs=3, eval_frequency=2, dir_name='t', dropout=0.5, regularization=5e-4)
    M0 = Ls[0].shape[0]
    rs = np.random.RandomState(0)
    S = 3 * 7 + 1
    data = rs.randn(S, M0, int(z['channel']))                    # float64 like perm_data_3d output
    labels = rs.randint(0, int(z['M'][-1]), S)
    np.random.seed(0)
    acc, losses, t_step = net.fit(data, labels, data[:5], labels[:5])
    assert len(acc) == len(losses) and t_step > 0 and net.global_step == int(3 * S / 3)
    pred = net.predict(data)
    assert pred.shape == (S,)
    pred2, loss = net.predict(data, labels)
    assert np.array_equal(pred, pred2) and np.isfinite(loss)
    string, accuracy, f1, loss2 = net.evaluate(data, labels)
    assert 0 <= accuracy <= 100 and 'accuracy' in string


def test_missing_gpu_path_fails_loudly(ops):
    from gcn_fmri_decoding_amd import _lib
    with pytest.raises(_lib.ChebgcnError):
        ops.cheb_conv(torch.zeros(1, 1, 32), torch.zeros(1, 1), None, None, 1)


# ---------------------------------------------------------------------------------------
# alternative kernel shapes: 4 planes per workgroup (opt-in) and the out-of-LDS fallback
# ---------------------------------------------------------------------------------------

def _run_fwd_bwd(ops, g, x, G, K):
    from gcn_fmri_decoding_amd import _lib
    lib = _lib.lib()
    B, Fin = x.shape[0], x.shape[1]
    stack = torch.full((K, B, Fin, g.Mp), float('nan'), device=x.device)
    dx = torch.full((B, Fin, g.Mp), float('nan'), device=x.device)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, ops._p(x), ops._p(stack), B, Fin, K, ops._stream()), 'fwd')
    _lib.check(lib.chebgcn_recurrence_bwd(g.handle, ops._p(G), ops._p(dx), B, Fin, K, ops._stream()), 'bwd')
    return stack, dx


@pytest.mark.parametrize('lvl,B,Fin,K', [(0, 3, 3, 5), (1, 2, 5, 6), (3, 1, 1, 2)])
def test_four_plane_kernel_matches_two_plane(ops, dev, lvl, B, Fin, K):
    """chebgcn_graph_create_planes(..., 4): only active vertices on chip, isolated ("fake") vertices
    patched in by the streaming code.  Same summation order as the default kernel, so the two agree
    to the last few ulps."""
    L = levels()[lvl]
    M = L.shape[0]
    torch.manual_seed(lvl)
    g2 = ops.Graph(L, dev, planes=2)
    g4 = ops.Graph(L, dev, planes=4)
    assert g2.query(6) == 2
    assert g4.query(6) == 4 and g4.query(7) <= M
    assert ops.Graph(L, dev).query(6) == 4               # the automatic choice for a small graph
    x = torch.randn(B, Fin, g2.Mp, device=dev)
    G = torch.randn(K, B, Fin, g2.Mp, device=dev)
    s2, d2 = _run_fwd_bwd(ops, g2, x, G, K)
    s4, d4 = _run_fwd_bwd(ops, g4, x, G, K)
    close(s4[:, :, :, :M].cpu().numpy(), s2[:, :, :, :M].cpu().numpy(), rel=1e-6, what='4-plane vs 2-plane stack')
    close(d4[:, :, :M].cpu().numpy(), d2[:, :, :M].cpu().numpy(), rel=1e-6, what='4-plane vs 2-plane dx')
    ref = R.cheb_stack(R.rescaled_laplacian(L, np.float32), from_storage(x, M), K)
    close(stack_to_ref(s4, M), ref, what='4-plane stack')


@pytest.mark.parametrize('nodes,B,Fin,K', [(10000, 3, 3, 5), (4000, 2, 5, 4), (10000, 1, 2, 25)])
def test_four_plane_large_graph_kernel(ops, dev, nodes, B, Fin, K):
    """recurrence4.hip (more than 2048 ranked rows): 16-byte LDS entries for the active vertices
    only, isolated ("fake") vertices stored at staging; B*Fin is not a multiple of 4, x and G are
    non-zero at the fake vertices.  Checked against the two-plane kernel, the adjoint identity
    and the oracle on two planes."""
    from gcn_fmri_decoding_amd import _lib, graph
    Ls, perm, _ = graph.synthetic_graph(nodes, k=8, levels=1)
    L = Ls[0]
    M = L.shape[0]
    g2 = ops.Graph(L, dev, planes=2)
    assert g2.query(6) == 2
    g4 = ops.Graph(L, dev, planes=4)
    assert g4.query(6) == 4 and 2048 < g4.query(7) < M        # some vertices are isolated
    assert ops.Graph(L, dev).query(6) == 4                    # and four planes are the automatic choice
    torch.manual_seed(nodes + K)
    x = torch.randn(B, Fin, g2.Mp, device=dev)
    G = torch.randn(K, B, Fin, g2.Mp, device=dev)
    x[:, :, M:] = 0
    G[:, :, :, M:] = 0
    s2, d2 = _run_fwd_bwd(ops, g2, x, G, K)
    s4, d4 = _run_fwd_bwd(ops, g4, x, G, K)
    close(s4[:, :, :, :M].cpu().numpy(), s2[:, :, :, :M].cpu().numpy(), what='4-plane vs 2-plane stack')
    close(d4[:, :, :M].cpu().numpy(), d2[:, :, :M].cpu().numpy(), what='4-plane vs 2-plane dx')
    lhs = (s4[:, :, :, :M].double() * G[:, :, :, :M].double()).sum().item()
    rhs = (x[:, :, :M].double() * d4[:, :, :M].double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0)
    Lr = R.rescaled_laplacian(L, np.float32)
    xp = x[B - 1, Fin - 2:Fin, :M].cpu().numpy().T.copy()                 # the last (partial) plane group
    ref = GR.chebyshev(Lr, xp, K)
    got = s4[:, B - 1, Fin - 2:Fin, :M].permute(0, 2, 1).cpu().numpy()
    close(got, ref, what='4-plane stack vs oracle')
    # in-place T_0 (x is slab 0 of the stack)
    s4b = torch.full((K, B, Fin, g4.Mp), float('nan'), device=dev)
    s4b[0] = x
    _lib.check(_lib.lib().chebgcn_recurrence_fwd(g4.handle, ops._p(s4b[0]), ops._p(s4b), B, Fin, K, ops._stream()), 'fwd')
    assert torch.equal(s4b[:, :, :, :M], s4[:, :, :, :M])


@pytest.mark.parametrize('planes', [0, 2])
@pytest.mark.parametrize('nodes', [1500, 2600, 6000, 13000])
def test_recurrence_other_kernel_shapes(ops, dev, nodes, planes):
    """Graph sizes that select the other workgroup shapes of the on-chip kernels -- two planes
    (256/512/768 threads, 8..32 rows per thread, with and without LDS-resident id records) and four
    planes (generic kernel up to 2048 rows; recurrence4.hip with 10 or 20 rows per thread beyond):
    forward against the oracle on two planes, adjoint by the identity <T(x), G> = <x, T*(G)>."""
    from gcn_fmri_decoding_amd import _lib, graph
    Ls, perm, _ = graph.synthetic_graph(nodes, k=8, levels=1)
    L = Ls[0]
    M = L.shape[0]
    if nodes > 10000:                                       # 16 bytes per active vertex do not fit the LDS
        with pytest.raises(_lib.ChebgcnError):
            ops.Graph(L, dev, planes=4)
    g = ops.Graph(L, dev, planes=planes)
    # automatic: four planes wherever 16 bytes per active vertex fit the LDS (up to ~10.2k active vertices)
    assert g.query(3) == 1 and g.query(6) == (2 if planes == 2 or nodes > 10000 else 4)
    B, Fin, K = 2, 3, 5
    torch.manual_seed(nodes)
    x = torch.randn(B, Fin, g.Mp, device=dev)
    G = torch.randn(K, B, Fin, g.Mp, device=dev)
    x[:, :, M:] = 0
    G[:, :, :, M:] = 0
    stack, dx = _run_fwd_bwd(ops, g, x, G, K)
    Lr = R.rescaled_laplacian(L, np.float32)
    ref = GR.chebyshev(Lr, x[0, :2, :M].cpu().numpy().T.copy(), K)
    close(stack[:, 0, :2, :M].permute(0, 2, 1).cpu().numpy(), ref, what='stack vs oracle')
    lhs = (stack[:, :, :, :M].double() * G[:, :, :, :M].double()).sum().item()
    rhs = (x[:, :, :M].double() * dx[:, :, :M].double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0)


def test_operator_layout_statistics(ops, dev):
    """graph_query 9..11: modelled LDS cycles of one gather pass in the caller's entry order, after
    the bank-aware placement of build_ell, and without any conflict."""
    from gcn_fmri_decoding_amd import graph
    Ls, _, _ = graph.synthetic_graph(4000, k=8, levels=1)
    g = ops.Graph(Ls[0], dev)
    before, after, ideal = g.query(9), g.query(10), g.query(11)
    assert ideal > 0 and ideal <= after <= before
    assert after < 0.75 * before                      # the placement removes a good part of the conflicts
    assert g.query(8) <= 160 * 1024 and g.query(5) >= 8


def test_out_of_lds_fallback(ops, dev):
    """A graph too large for the LDS image (M > 20480) takes the kernel-per-step path."""
    rs = np.random.RandomState(5)
    M, deg = 21000, 6
    rows = np.repeat(np.arange(M), deg)
    cols = rs.randint(0, M, M * deg)
    W = sp.coo_matrix((rs.rand(M * deg).astype(np.float32), (rows, cols)), shape=(M, M)).tocsr()
    W = W + W.T
    W.setdiag(0)
    W.eliminate_zeros()
    from gcn_fmri_decoding_amd import graph
    L = graph.laplacian(W.astype(np.float32), normalized=True)
    g = ops.Graph(L, dev)
    assert not g.on_chip
    B, Fin, K = 2, 3, 4
    torch.manual_seed(1)
    x = torch.randn(B, Fin, g.Mp, device=dev)
    x[:, :, M:] = 0
    G = torch.randn(K, B, Fin, g.Mp, device=dev)
    G[:, :, :, M:] = 0
    stack, dx = _run_fwd_bwd(ops, g, x, G, K)
    ref = R.cheb_stack(R.rescaled_laplacian(L, np.float32), from_storage(x, M), K)
    close(stack_to_ref(stack, M), ref, what='fallback stack')
    lhs = (stack[:, :, :, :M].double() * G[:, :, :, :M].double()).sum().item()
    rhs = (x[:, :, :M].double() * dx[:, :, :M].double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0)
