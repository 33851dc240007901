"""Every arm of the library's dispatchers, reached by a VALUE test that names it.  Needs an MI355X: ``-m gpu``.

The dispatchers of libchebgcn.so choose kernel templates by shape and by the device's CU count
(contract.hip ``small_launch`` / ring / LDS conditions, ``bw_rt``, recurrence.hip ``dispatch_onchip``, ``pick_ell``,
pointwise.hip ``brelu_bwd_blocks``).  ``chebgcn_last_dispatch()`` reports which templates the last launching call
enqueued; each case below asserts that name -- a change of a dispatch condition that silently moves a test onto
another kernel fails here -- and then compares the results with float64 products of the same operands computed on the
device (pads poisoned with NaN), or with the CPU oracle.  All numbers assume the 256 CUs of an MI355X.

=============================================  ===============================================================
case (M = 10466 unless stated)                 kernel templates asserted
=============================================  ===============================================================
bench step, B=64, 32->32, K=5                  contract_fwd_ring_kernel, contract_bwd_x_lds_kernel<true|false>,
                                               contract_bwd_w_kernel<5,true|false> (768 workgroups, in-kernel reduce)
config 4 big launch, B=25, 64->64, K=25        contract_fwd_kernel<2>, contract_bwd_x_kernel<false,*,false>,
                                               contract_bwd_w_kernel<5,*> with gy = 10
config 5 in fp32, B=64, 60->256, K=5           contract_fwd_kernel<2> (4 filter blocks), contract_bwd_x_kernel<false,*,false>,
                                               contract_bwd_w_kernel<5,*> gy = 2, gz = 8   (the reference leg of rel_err_vs_f32)
W beyond 48 KB, B=25, 32->32, K=25             contract_fwd_kernel<1>, contract_bwd_x_kernel<true,*,false>
small launch, B=3, 32->32, K=5                 contract_fwd_splitk_kernel, contract_bwd_x_kernel<true,*,true>
small launch, B=3, 64->64, K=25                contract_fwd_kernel<2>, contract_bwd_x_kernel<false,*,true>
recurrence, B*Fin = 8192 / 2048 / 960 planes   cheb4_kernel<10240,20,6,512,*>, cheb_onchip_kernel<2,14,4,768,*>;
                                               cheb_ord_kernel<10240,6,5,512,*> on the length-ordered graph
configs[1] network at batch 64                 the set of templates of one training step, fused feature mean on and off
=============================================  ===============================================================

Tolerance: ``max|hip - ref| <= 1e-5 max|ref|`` forward, 2e-5 gradients (north star: 1e-5 relative fp32), per tensor;
the network test states its own.  Measured errors go to gpurun_out/parity_measured.jsonl (conftest.record_measured).
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import assert_adam_params_close, record_measured

pytestmark = pytest.mark.gpu
REL = 1e-5
GREL = 2e-5


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    assert torch.cuda.get_device_properties(0).multi_processor_count == 256, 'the dispatch arms asserted here assume 256 CUs'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops():
    from gcn_fmri_decoding_amd import ops as o
    return o


@pytest.fixture(scope='module')
def lib():
    from gcn_fmri_decoding_amd import _lib
    return _lib.lib()


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def rel_err(got, ref, scale=None):
    scale = ref.abs().max() if scale is None else scale
    return float((got.double() - ref).abs().max() / scale)


EPS32 = float(np.finfo(np.float32).eps)


def elementwise_eps(got, ref, mag):
    """max over the ELEMENTS of |got - ref| / (eps32 * mag), mag = the sum of the magnitudes of the products an element adds up
    (the scale of its own rounding error: the textbook bound is n * eps * mag for n terms, ~sqrt(n) in practice).  A per-tensor
    max norm hides a wrong element wherever the tensor has larger ones; this does not."""
    d = (got.double() - ref).abs()
    return float((d / (EPS32 * mag.clamp(min=1e-30))).max())


CASES = {
    # name: (B, M, Fin, K, Fout, forward, bwd_x stem, bwd_w RT, bwd_w reduce)
    'bench_b64': (64, 10466, 32, 5, 32, 'contract_fwd_ring_kernel', 'contract_bwd_x_lds_kernel<%s>', 5, 'big'),
    # (round 6: the weight gradient launches two workgroups per CU over ALL its row-tile groups and column tiles: launches of several
    # groups have at most 256 workgroups per group and their partials go through reduce_partials_small)
    'config4_b25': (25, 10466, 64, 25, 64, 'contract_fwd_kernel<2>', 'contract_bwd_x_kernel<false,%s,false>', 5, 'small'),
    'config5_f32_b64': (64, 10466, 60, 5, 256, 'contract_fwd_kernel<2>', 'contract_bwd_x_kernel<false,%s,false>', 5, 'small'),
    'wide_w_b25': (25, 10466, 32, 25, 32, 'contract_fwd_kernel<1>', 'contract_bwd_x_kernel<true,%s,false>', 5, 'small'),
    'small_b3': (3, 10466, 32, 5, 32, 'contract_fwd_splitk_kernel', 'contract_bwd_x_kernel<true,%s,true>', 5, 'small'),
    'small_config4_b3': (3, 10466, 64, 25, 64, 'contract_fwd_kernel<2>', 'contract_bwd_x_kernel<false,%s,true>', 5, 'small'),
    'first_layer_b64': (64, 10466, 15, 5, 32, 'contract_fwd_ring_kernel', 'contract_bwd_x_kernel<true,%s,false>', 3, 'big'),
}


@pytest.mark.parametrize('case', sorted(CASES))
def test_contraction_arm_vs_float64(ops, dev, lib, case):
    """Forward (per-vertex bias, ReLU, bit mask), both gradients with the ReluGrad folded in (``_relu`` entry points) and
    without (plain entry points on the gated gradient): dispatcher arm asserted, values against float64 products
    (models_gcn.py:611-617 and its autodiff :297-303)."""
    from gcn_fmri_decoding_amd import _lib
    B, M, Fin, K, Fout, fwd_name, bwx_stem, rt, red = CASES[case]
    Mp = ops.plane_stride(M)
    gen = torch.Generator(device=dev)
    gen.manual_seed(sum(map(ord, case)))
    stack = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    stack[..., M:] = float('nan')
    W = torch.randn((Fin * K, Fout), generator=gen, device=dev) * (0.5 / np.sqrt(Fin * K))
    bias = torch.zeros((Fout, Mp), device=dev)
    bias[:, :M] = torch.randn((Fout, M), generator=gen, device=dev) * 0.3
    st = stream()
    got = {}

    out = torch.full((B, Fout, Mp), float('nan'), device=dev)
    mask = torch.zeros((B, Fout, Mp // 4), dtype=torch.uint8, device=dev)
    _lib.check(lib.chebgcn_contract_fwd(P(stack), P(W), P(bias), ops.BIAS_VERTEX, P(out), P(mask), B, M, Fin, K, Fout, 1, 0, 1, st), 'fwd')
    assert _lib.last_dispatch() == fwd_name
    S = stack[..., :M].permute(2, 0, 1, 3).reshape(Fin * K, B, M).double()          # rows fin*K + k
    pre = torch.einsum('rbm,ro->bom', S, W.double()) + bias[:, :M].double()
    got['fwd'] = rel_err(out[..., :M], pre.clamp(min=0), pre.abs().max())
    assert got['fwd'] <= REL, 'contract_fwd (%s): %.3e' % (fwd_name, got['fwd'])
    mag = torch.einsum('rbm,ro->bom', S.abs(), W.double().abs()) + bias[:, :M].double().abs()
    got['fwd_elementwise_eps'] = elementwise_eps(out[..., :M], pre.clamp(min=0), mag)
    assert got['fwd_elementwise_eps'] <= 16, 'contract_fwd (%s): an element is %.1f eps of its own terms off' % (fwd_name, got['fwd_elementwise_eps'])
    del mag
    bits = torch.stack([(mask >> r) & 1 for r in range(4)], -1).reshape(B, Fout, Mp)[..., :M].bool()
    assert torch.equal(bits, out[..., :M] > 0), 'ReLU bit mask disagrees with the output'
    del pre

    gout = torch.randn((B, Fout, Mp), generator=gen, device=dev)
    gated = gout.clone()
    gated[..., :M] *= bits
    gated[..., M:] = 0.0                      # plain entry points: the pad columns of dy are the caller's zeros
    gout[..., M:] = float('nan')
    dy = gated[..., :M].double()
    n = lib.chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    dW_ref = torch.einsum('rbm,bom->ro', S, dy)
    dW_mag = torch.einsum('rbm,bom->ro', S.abs(), dy.abs())
    del S
    ntiles = (Fin * K + 31) // 32
    assert min(ntiles, 5) == rt
    groups = ((ntiles + rt - 1) // rt) * ((Fout + 31) // 32)            # the launcher's arithmetic (contract.hip bw_grid_x)
    gx = min(max((512 // groups + 63) // 64 * 64, 64), (B * ((M + 63) // 64) + 2) // 3)
    assert ('small' if gx <= 256 else 'big') == red, (gx, groups)
    dWs = {}
    for folded in (True, False):
        dW = torch.full((Fin * K, Fout), float('nan'), device=dev)
        if folded:
            _lib.check(lib.chebgcn_contract_bwd_w_relu(P(stack), P(gout), P(mask), P(dW), P(ws), n, B, M, Fin, K, Fout, st), 'bwd_w_relu')
        else:
            _lib.check(lib.chebgcn_contract_bwd_w(P(stack), P(gated), P(dW), P(ws), n, B, M, Fin, K, Fout, st), 'bwd_w')
        name = _lib.last_dispatch()
        assert name.startswith('contract_bwd_w_kernel<%d,%s>' % (rt, 'true' if folded else 'false')), name
        assert name.endswith(BWD_W_TAIL[red]), name
        key = 'bwd_w_relu' if folded else 'bwd_w'
        got[key] = rel_err(dW, dW_ref)
        assert got[key] <= GREL, '%s (%s): %.3e' % (key, name, got[key])
        # a sum over B*M ~ 1e5..1e6 products: fixed-order partial sums of partial sums, errors ~ sqrt(n) eps of the terms' magnitudes
        got[key + '_elementwise_eps'] = elementwise_eps(dW, dW_ref, dW_mag)
        assert got[key + '_elementwise_eps'] <= 16, '%s (%s): an element is %.1f eps of its own terms off' % (key, name, got[key + '_elementwise_eps'])
        dWs[folded] = dW
    assert torch.equal(dWs[True], dWs[False]), 'folded and plain dW differ'            # the same products in the same order

    gs_ref = torch.einsum('ro,bom->rbm', W.double(), dy).reshape(Fin, K, B, M).permute(1, 2, 0, 3)
    gs_mag = torch.einsum('ro,bom->rbm', W.double().abs(), dy.abs()).reshape(Fin, K, B, M).permute(1, 2, 0, 3)
    gss = {}
    for folded in (True, False):
        gstack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
        if folded:
            _lib.check(lib.chebgcn_contract_bwd_x_relu(P(gout), P(mask), P(W), P(gstack), B, M, Fin, K, Fout, st), 'bwd_x_relu')
        else:
            _lib.check(lib.chebgcn_contract_bwd_x(P(gated), P(W), P(gstack), B, M, Fin, K, Fout, st), 'bwd_x')
        name = _lib.last_dispatch()
        assert name == bwx_stem % ('true' if folded else 'false'), name
        key = 'bwd_x_relu' if folded else 'bwd_x'
        got[key] = rel_err(gstack[..., :M], gs_ref)
        assert got[key] <= GREL, '%s (%s): %.3e' % (key, name, got[key])
        got[key + '_elementwise_eps'] = elementwise_eps(gstack[..., :M], gs_ref, gs_mag)
        assert got[key + '_elementwise_eps'] <= 16, '%s (%s): an element is %.1f eps of its own terms off' % (key, name, got[key + '_elementwise_eps'])
        gss[folded] = gstack[..., :M]
    assert torch.equal(gss[True], gss[False]), 'folded and plain dstack differ'
    record_measured('contraction_arm_vs_float64[%s]' % case, **got)


# ---------------------------------------------------------------------------------------
# wide layers: the split-bf16 contraction ('bf16x3', what precision 'auto' resolves to above 32 filters) at 1e-5
# ---------------------------------------------------------------------------------------

SPLIT_CASES = {
    # name: (B, Fin, K, Fout, forward kernel, bwd_x kernel, bwd_w kernel)
    'config4_b16': (16, 64, 25, 64, 'contract_fwd_bf16_kernel<3,4,tiles4>', 'contract_fwd_bf16_kernel<3,5>', 'contract_bwd_w_bf16_kernel<5,2,3>'),
    'config5_b32': (32, 60, 5, 256, 'contract_fwd_bf16_kernel<3,4>', 'contract_fwd_bf16_kernel<3,5>', 'contract_bwd_w_bf16_wide_kernel<3>'),
    'f128_b32': (32, 128, 5, 128, 'contract_fwd_bf16_kernel<3,4,tiles2>', 'contract_fwd_bf16_kernel<3,5>', 'contract_bwd_w_bf16_wide_kernel<3>'),
    'pool6_l2_b32': (32, 32, 10, 64, 'contract_fwd_bf16_kernel<3,4,tiles4>', 'contract_fwd_bf16_kernel<3,5>', 'contract_bwd_w_bf16_kernel<5,2,3>'),
    'pool6_l5_b32': (32, 64, 5, 128, 'contract_fwd_bf16_kernel<3,4,tiles2>', 'contract_fwd_bf16_kernel<3,5>', 'contract_bwd_w_bf16_wide_kernel<3>'),
}


@pytest.mark.parametrize('bias_kind', ['vertex', 'filter'])
@pytest.mark.parametrize('case', sorted(SPLIT_CASES))
def test_split_bf16_arm_vs_float64(ops, dev, lib, case, bias_kind):
    """chebgcn_contract_fwd_bf16 / _bwd_w_bf16 / _bwd_x_bf16 with passes = 3 (each fp32 operand split into two bf16, hi*hi +
    hi*lo + lo*hi accumulated in fp32) against float64 products of the SAME fp32 operands, at the bound the fp32 kernels are
    held to -- 1e-5 of the tensor's scale (north star: "1e-5 relative fp32"), not a bf16-style bound: this is the arithmetic
    precision 'auto' gives every layer of more than 32 filters (ops.resolve_precision).  Kernel templates asserted: the forward
    of 64 / 128 filters runs four / two vertex tiles per work item (contract_bf16.hip Bf16Cfg NVT), config 4's K = 25 sums
    are the longest of the BASELINE configs (1600 terms).  models_gcn.py:611-617, :297-303."""
    from gcn_fmri_decoding_amd import _lib
    B, Fin, K, Fout, fwd_name, bwx_name, bww_name = SPLIT_CASES[case]
    if bias_kind == 'filter' and case not in ('config4_b16', 'f128_b32'):
        pytest.skip('per-filter bias on one shape per tile geometry')
    M = 10466
    Mp = ops.plane_stride(M)
    gen = torch.Generator(device=dev)
    gen.manual_seed(sum(map(ord, case)))
    stack = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    stack[..., M:] = float('nan')
    W = torch.randn((Fin * K, Fout), generator=gen, device=dev) * (0.5 / np.sqrt(Fin * K))
    if bias_kind == 'vertex':
        bias = torch.zeros((Fout, Mp), device=dev)
        bias[:, :M] = torch.randn((Fout, M), generator=gen, device=dev) * 0.3
        bk, bias64 = ops.BIAS_VERTEX, bias[:, :M].double()
    else:
        bias = torch.randn((Fout,), generator=gen, device=dev) * 0.3
        bk, bias64 = ops.BIAS_FILTER, bias.double()[:, None]
    st = stream()
    got = {}
    out = torch.full((B, Fout, Mp), float('nan'), device=dev)
    mask = torch.zeros((B, Fout, Mp // 4), dtype=torch.uint8, device=dev)
    nws = lib.chebgcn_contract_fwd_bf16_workspace(Fin, K, Fout)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    _lib.check(lib.chebgcn_contract_fwd_bf16(P(stack), P(W), P(bias), bk, P(out), P(mask), B, M, Fin, K, Fout, 1, 0, 1, 3, P(ws), nws, st), 'fwd')
    assert _lib.last_dispatch() == 'pack_w_bf16_kernel + ' + fwd_name, _lib.last_dispatch()
    S = stack[..., :M].permute(2, 0, 1, 3).reshape(Fin * K, B, M).double()
    pre = torch.einsum('rbm,ro->bom', S, W.double()) + bias64
    got['fwd'] = rel_err(out[..., :M], pre.clamp(min=0), pre.abs().max())
    assert got['fwd'] <= REL, '%s: %.3e' % (fwd_name, got['fwd'])
    bits = torch.stack([(mask >> r) & 1 for r in range(4)], -1).reshape(B, Fout, Mp)[..., :M].bool()
    assert torch.equal(bits, out[..., :M] > 0), 'ReLU bit mask disagrees with the output'
    del pre
    dyf = torch.randn((B, Fout, Mp), generator=gen, device=dev)
    dyf[..., :M] *= bits
    dyf[..., M:] = 0.0
    dy = dyf[..., :M].double()
    n = lib.chebgcn_contract_bwd_w_bf16_workspace(B, M, Fin, K, Fout)
    wsw = torch.empty(n, dtype=torch.uint8, device=dev)
    dW = torch.full((Fin * K, Fout), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_w_bf16(P(stack), P(dyf), P(dW), P(wsw), n, B, M, Fin, K, Fout, 3, st), 'bwd_w')
    assert _lib.last_dispatch().startswith(bww_name), _lib.last_dispatch()
    got['bwd_w'] = rel_err(dW, torch.einsum('rbm,bom->ro', S, dy))
    assert got['bwd_w'] <= REL, '%s: %.3e' % (bww_name, got['bwd_w'])
    dW2 = torch.full_like(dW, float('nan'))
    _lib.check(lib.chebgcn_contract_bwd_w_bf16(P(stack), P(dyf), P(dW2), P(wsw), n, B, M, Fin, K, Fout, 3, st), 'bwd_w')
    assert torch.equal(dW, dW2), 'dW differs between two runs'
    del S
    n = lib.chebgcn_contract_bwd_x_bf16_workspace(Fin, K, Fout)
    wsx = torch.empty(n, dtype=torch.uint8, device=dev)
    gstack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_x_bf16(P(dyf), P(W), P(gstack), B, M, Fin, K, Fout, 3, P(wsx), n, st), 'bwd_x')
    assert _lib.last_dispatch() == 'pack_w_bf16_kernel<transposed> + ' + bwx_name, _lib.last_dispatch()
    gs_ref = torch.einsum('ro,bom->rbm', W.double(), dy).reshape(Fin, K, B, M).permute(1, 2, 0, 3)
    got['bwd_x'] = rel_err(gstack[..., :M], gs_ref)
    assert got['bwd_x'] <= REL, '%s: %.3e' % (bwx_name, got['bwd_x'])
    del gs_ref, gstack
    # one pass (operands rounded to bf16) through the same tile geometry: the bf16 bound
    out1 = torch.full((B, Fout, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_fwd_bf16(P(stack), P(W), P(bias), bk, P(out1), None, B, M, Fin, K, Fout, 1, 0, 1, 1, P(ws), nws, st), 'fwd1')
    assert _lib.last_dispatch() == 'pack_w_bf16_kernel + ' + fwd_name.replace('<3,', '<1,'), _lib.last_dispatch()
    got['fwd_one_pass_vs_split'] = rel_err(out1[..., :M], out[..., :M].double())
    assert got['fwd_one_pass_vs_split'] <= 1e-2
    record_measured('split_bf16_arm_vs_float64[%s,%s]' % (case, bias_kind), **got)


# how contract_bwd_w's partials are reduced, by launch size (contract.hip launch_bwd_w)
BWD_W_TAIL = {'big': ' + reduce_partials_wide', 'small': ' + reduce_partials_small'}


ORD_F, ORD_A = 'cheb_ord_kernel<10240,6,5,512,false>', 'cheb_ord_kernel<10240,6,5,512,true>'


@pytest.mark.parametrize('B,Fin,K,fwd,adj', [
    (256, 32, 5, 'cheb4_kernel<10240,20,6,512,false,true>', 'cheb4_kernel<10240,20,6,512,true,false>'),       # north star
    (64, 32, 5, 'cheb_onchip_kernel<2,14,4,768,false>', 'cheb_onchip_kernel<2,14,4,768,true>'),               # batch 64, layers 2-6
    (64, 15, 5, 'cheb_onchip_kernel<2,14,4,768,false>', 'cheb_onchip_kernel<2,14,4,768,true>'),               # batch 64, layer 1
    (64, 64, 25, 'cheb4_kernel<10240,20,6,512,false,true>', 'cheb4_kernel<10240,20,6,512,true,false>'),       # config 4
    # the same launches on the graph relabelled by descending row length (graph.length_order: what cgcnn builds for a
    # network without pooling, i.e. what the bench step runs): csrc/recurrence_ord.hip whatever the launch size
    (256, 32, 5, ORD_F, ORD_A), (64, 32, 5, ORD_F, ORD_A), (64, 15, 5, ORD_F, ORD_A), (64, 64, 25, ORD_F, ORD_A), (1, 3, 2, ORD_F, ORD_A),
])
def test_recurrence_arm(ops, dev, lib, B, Fin, K, fwd, adj):
    """Which recurrence kernel a launch of B*Fin planes on the benchmark graph reaches (common.h ``pick_ell``: four planes
    per workgroup from four plane groups per CU; the ordered kernels where the rows come sorted by length) -- the values of
    these launches are checked by test_gpu_bench_shapes.py::test_northstar_launch_properties; here every plane of one
    window against the oracle."""
    import bench
    from gcn_fmri_decoding_amd import _lib, graph
    from oracle import graph_ref as GR
    Ls, _ = bench.load_graph(10000, 1, 0, 1, None)
    L = Ls[0]
    M = L.shape[0]
    if fwd == ORD_F:
        order = graph.length_order(L)
        g = ops.Graph(L, dev, order=order)
        assert g.ordered
        L = graph.permute(L, order)
    else:
        g = ops.graph_for(L, dev)
        assert not g.ordered
    Mp = g.Mp
    gen = torch.Generator(device=dev)
    gen.manual_seed(B + Fin)
    x = torch.randn((B, Fin, Mp), generator=gen, device=dev)
    x[..., M:] = float('nan')
    stack = torch.empty((K, B, Fin, Mp), device=dev)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(x), P(stack), B, Fin, K, stream()), 'recurrence_fwd')
    assert _lib.last_dispatch() == fwd
    G = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    G[..., M:] = float('nan')
    dx = torch.empty((B, Fin, Mp), device=dev)
    _lib.check(lib.chebgcn_recurrence_bwd(g.handle, P(G), P(dx), B, Fin, K, stream()), 'recurrence_bwd')
    assert _lib.last_dispatch() == adj
    # oracle on the planes of one window from the middle of the batch (graph.chebyshev, graph.py:155-172, and its adjoint)
    Lr = GR.rescale_L(L, 2)
    b = B // 2
    xv = x[b, :, :M].cpu().numpy().T.astype(np.float32)                      # [M, Fin]
    T = [xv, (Lr @ xv).astype(np.float32)]
    for k in range(2, K):
        T.append((2 * (Lr @ T[-1]) - T[-2]).astype(np.float32))
    ref = np.stack(T).transpose(0, 2, 1)
    e_f = np.abs(stack[:, b, :, :M].cpu().numpy() - ref).max() / np.abs(ref).max()
    Gb = G[:, b, :, :M].cpu().numpy().transpose(0, 2, 1).astype(np.float64)   # [K, M, Fin]
    LT = Lr.T.tocsr().astype(np.float64)
    c1, c2 = Gb[K - 1], np.zeros_like(Gb[0])
    for j in range(K - 2, 0, -1):
        c1, c2 = Gb[j] + 2 * (LT @ c1) - c2, c1
    dref = Gb[0] + LT @ c1 - c2
    e_a = np.abs(dx[b, :, :M].cpu().numpy().T - dref).max() / np.abs(dref).max()
    record_measured('recurrence_arm[%d,%d,%d]' % (B, Fin, K), fwd=e_f, adjoint=e_a)
    assert e_f <= REL, 'forward %s: %.3e' % (fwd, e_f)
    # K = 25 grows the terms by 2^k-like factors before they cancel: the K-term sums are compared at the scale of the result
    assert e_a <= GREL, 'adjoint %s: %.3e' % (adj, e_a)


def test_bwd_x_mask_at_the_end_of_an_allocation(ops, dev, lib):
    """contract_bwd_x_lds_kernel reads the ReLU mask one dword per lane; a mask row is Mp / 4 bytes and Mp is padded to 32
    vertices, not to the 128 of a tile: with Mp % 128 == 32 the last tile's dwords 2..7 lie beyond the row.  The mask sits at
    the very end of its (exact-size) allocation here; results against float64."""
    from gcn_fmri_decoding_amd import _lib
    B, M, Fin, K, Fout = 512, 1050, 32, 5, 32
    Mp = ops.plane_stride(M)
    assert Mp % 128 == 32 and (B * Fout * (Mp // 4)) % 512 == 0
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    W = torch.randn((Fin * K, Fout), generator=gen, device=dev) * 0.1
    gout = torch.randn((B, Fout, Mp), generator=gen, device=dev)
    keep = torch.rand((B, Fout, Mp), generator=gen, device=dev) > 0.5
    keep[..., M:] = False
    packed = (keep.reshape(B, Fout, Mp // 4, 4).to(torch.uint8) * torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=dev)).sum(-1).to(torch.uint8)
    torch.cuda.empty_cache()
    mask = torch.empty(B * Fout * (Mp // 4), dtype=torch.uint8, device=dev)          # its own block, exact size
    mask.copy_(packed.reshape(-1))
    gstack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_x_relu(P(gout), P(mask), P(W), P(gstack), B, M, Fin, K, Fout, stream()), 'bwd_x_relu')
    assert _lib.last_dispatch() == 'contract_bwd_x_lds_kernel<true>'
    torch.cuda.synchronize()
    dy = (gout[..., :M] * keep[..., :M]).double()
    ref = torch.einsum('ro,bom->rbm', W.double(), dy).reshape(Fin, K, B, M).permute(1, 2, 0, 3)
    err = rel_err(gstack[..., :M], ref)
    assert err <= GREL, 'contract_bwd_x_relu at Mp %% 128 == 32: %.3e' % err


def test_fwd_mean_not_served_below_four_filters(lib):
    """chebgcn_contract_fwd_mean's ring kernel loads 16 bytes around a per-filter bias: layers of fewer than four filters are
    declined (the caller runs chebgcn_contract_fwd + chebgcn_feature_mean_fwd)."""
    assert lib.chebgcn_contract_fwd_mean_supported(64, 10466, 32, 5, 3) == 0
    assert lib.chebgcn_contract_fwd_mean_supported(64, 10466, 32, 5, 4) == 1


# ---------------------------------------------------------------------------------------
# BASELINE configs[1] at the batch the benchmark times
# ---------------------------------------------------------------------------------------

def _oracle_worker(args):
    """Runs in a fresh process (spawn): the NumPy oracle of one training step in the given precision."""
    import numpy as np
    import bench
    from oracle import layers_ref as R
    dtype, seed, B = args
    dt = np.dtype(dtype).type
    Ls, _ = bench.load_graph(10000, 1, 0, 1, None)
    L = Ls[0].astype(dt)
    F, K, p, Mfc, C = [32] * 6, [5] * 6, [1] * 6, [512, 256, 22], 15
    onet = R.Net([L], F, K, p, Mfc, channel=C, brelu='b2relu', regularization=5e-4, dtype=dt)
    params, x, labels = _network_inputs(onet, L.shape[0], C, B, seed)
    params = {k: v.astype(dt) for k, v in params.items()}
    logits, cache = onet.forward(params, x.astype(dt))
    loss, dlogits = onet.loss(params, logits, labels)
    grads = onet.backward(params, cache, dlogits)
    return logits, float(loss), grads


def _network_inputs(onet, M, C, B, seed):
    rs = np.random.RandomState(seed)
    params = {}
    for k, s in onet.param_shapes().items():
        params[k] = ((0.2 + 0.05 * rs.randn(*s)) if k.endswith('bias') else rs.randn(*s) * np.sqrt(2.0 / s[0])).astype(np.float32)
    x = rs.randn(B, M, C).astype(np.float32)
    labels = rs.randint(0, 21, B)
    return params, x, labels


STEP_KERNELS = {
    # the Clenshaw form of the input gradient (ops.dx_by_forward = False): contract_bwd_x + recurrence_bwd
    'clenshaw': {
        'recurrence_fwd': {'cheb_ord_kernel<10240,6,5,512,false>'},          # (cgcnn relabels the vertices: graph.length_order)
        'recurrence_bwd': {'cheb_ord_kernel<10240,6,5,512,true>'},
        'contract_fwd': {'contract_fwd_ring_kernel'},
        'contract_bwd_x_relu': {'contract_bwd_x_lds_kernel<true>'},
        'brelu_pool_bwd': {'bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX,4>'},
    },
    # the default: the forward recurrence on dy with the transposed operator + the forward contraction on the re-indexed weights
    'forward': {
        'recurrence_fwd': {'cheb_ord_kernel<10240,6,5,512,false>'},
        'recurrence_fwd_t': {'cheb_ord_kernel<10240,6,5,512,false>'},
        'contract_fwd': {'contract_fwd_ring_kernel'},
        # layers 3-6 store their input gradient gated by the mask of the layer below (ops.GateLink), whose ReluGrad pass is then
        # the plain sum for the bias; layer 2 feeds layer 1, which folds its ReluGrad into the weight gradient
        'contract_bwd_x': {'contract_fwd_ring_kernel<gated>', 'contract_fwd_ring_kernel'},
        'brelu_pool_bwd': {'bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX,4>', 'bias_grad_sum_kernel<CHEBGCN_BIAS_VERTEX,4>'},
    },
}


@pytest.mark.parametrize('dx_form', ['forward', 'clenshaw'])
def test_config2_network_b64_vs_oracle(ops, dev, dx_form, monkeypatch):
    """BASELINE configs[1] as bench.py times it -- 6 x [K=5, F=32, p=1, b2relu], FC 512-256-22, block_dura 15, batch 64 on
    the M = 10466 graph -- logits, loss, EVERY gradient and one TF-form Adam step against the oracle
    (oracle/layers_ref.Net <-> lib_new/models_gcn.py:658-682, :253-276, :296), with the last layer fused with
    tf.reduce_mean (:673) and without.  The kernels of the step are the ring / LDS / ordered-recurrence instantiations
    (asserted from the dispatch log), not the small-launch ones the batch-2 test reaches; the model keeps its vertices in
    graph.length_order internally, inputs, variables and gradients cross the boundary in the reference's order.

    Ground truth is the oracle in float64; the fp32 oracle run beside it gives the error fp32 arithmetic itself makes on
    this network (another summation order, another side of zero for a ReLU within round-off).  Bounds: logits 1e-5 of
    max (north star) against float64; gradients at the 99.9 % quantile (per-vertex biases 99 %) within 5e-5 of the
    gradient's scale or three times the fp32 oracle's own error, whichever is larger (measured, round 4: weights 2.4e-5 ..
    2.2e-4 on the GPU against 1.6e-5 .. 1.5e-4 for NumPy's fp32 -- the first layer, whose gradient passes six layers of
    ReLU decisions, is the worst in both; FC head 1e-7)."""
    import concurrent.futures as cf
    import multiprocessing as mp
    import bench
    from gcn_fmri_decoding_amd import _lib, models_gcn
    from oracle import layers_ref as R
    monkeypatch.setattr(ops, 'dx_by_forward', dx_form == 'forward')
    B, seed, reg = 64, 4, 5e-4
    with cf.ProcessPoolExecutor(2, mp_context=mp.get_context('spawn')) as ex:
        futs = [ex.submit(_oracle_worker, ('float32', seed, B)), ex.submit(_oracle_worker, ('float64', seed, B))]
        Ls, _ = bench.load_graph(10000, 1, 0, 1, None)
        L = Ls[0]
        M = L.shape[0]
        F, K, p, Mfc, C = [32] * 6, [5] * 6, [1] * 6, [512, 256, 22], 15
        onet = R.Net([L], F, K, p, Mfc, channel=C, brelu='b2relu', regularization=reg)
        params, x, labels = _network_inputs(onet, M, C, B, seed)
        xs = torch.full((B, C, ops.plane_stride(M)), float('nan'), device=dev)
        xs[:, :, :M] = torch.as_tensor(np.ascontiguousarray(x.transpose(0, 2, 1))).to(dev)
        ld = torch.as_tensor(labels).to(dev)
        results = {}
        for fused in (True, False):
            net = models_gcn.cgcnn({'device': dev}, [L] * 6, F, K, p, Mfc, filter='chebyshev5', brelu='b2relu', pool='mpool1',
                                   initial='he', channel=C, regularization=reg, dropout=1, batch_size=B, verbose=False)
            net.fuse_feature_mean = fused
            for k, v in params.items():
                net.set_variable(k, v)
            with torch.no_grad():
                logits = net._inference_storage(xs, 1).cpu().numpy()
            _lib.dispatch_log = log = []
            try:
                _, loss_avg = net.train_step(xs, ld)
                torch.cuda.synchronize()
            finally:
                _lib.dispatch_log = None
            seen = {}
            for what, name in log:
                seen.setdefault(what, set()).add(name)
            expect = dict(STEP_KERNELS[dx_form])
            if fused:
                expect['contract_fwd_mean'] = {'contract_fwd_ring_kernel<mean>'}
                if dx_form == 'clenshaw':
                    expect['contract_bwd_x_relu_mean'] = {'contract_bwd_x_lds_kernel<true>'}
                    expect['bias_grad_relu_mean'] = {'bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX,4><mean>'}
                else:
                    expect['relu_grad_mean'] = {'bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX,4><mean>'}
            for what, names in expect.items():
                assert seen.get(what) == names, (what, seen.get(what), names)
            absent = ('recurrence_bwd', 'contract_bwd_x_relu', 'contract_bwd_x_relu_mean') if dx_form == 'forward' else ('recurrence_fwd_t', 'contract_bwd_x')
            assert not any(a in seen for a in absent), sorted(seen)
            # layers 2-6 under the forward form read a materialised dy (plain kernel); layer 1 (no input gradient) and the Clenshaw form fold the ReluGrad
            wnames = ('contract_bwd_w_kernel<5,false>', 'contract_bwd_w_kernel<3,true>') if dx_form == 'forward' else ('contract_bwd_w_kernel<5,true>', 'contract_bwd_w_kernel<3,true>')
            assert {n.split(' + ')[0] for n in seen['contract_bwd_w']} == set(wnames), seen['contract_bwd_w']
            grads = {}
            for k in params:
                grads[k] = net.gradient(k).cpu().numpy().astype(np.float64)
            # (loss_average after ONE step = 0.1 * loss: the first read of the zero-initialised 0.9-EMA, cgcnn.ema_zero_debias)
            results[fused] = (logits, float(loss_avg) / 0.1, grads, {k: net.get_var(k) for k in params})
            del net
        (l32, loss32, g32), (l64, loss64, g64) = [f.result() for f in futs]
    assert l64.dtype == np.float64
    measured = {}
    for fused, (logits, loss, grads, after) in results.items():
        tag = 'fused' if fused else 'unfused'
        e = np.abs(logits - l64).max() / np.abs(l64).max()
        e32 = np.abs(l32 - l64).max() / np.abs(l64).max()
        measured['logits_' + tag] = e
        measured['logits_oracle32'] = e32
        assert e <= max(REL, 3 * e32), 'logits (%s): %.3e of max against float64 (fp32 oracle: %.3e)' % (tag, e, e32)
        assert abs(loss - loss64) <= GREL * abs(loss64), 'loss (%s): %.8f against %.8f' % (tag, loss, loss64)
        for k in params:
            l2 = reg * params[k].astype(np.float64) if onet.regularized(k) else 0
            ref = g64[k] - l2
            scale = max(np.abs(ref).max(), 1e-30)
            e_gpu = np.abs(grads[k] - ref) / scale
            e_o32 = np.abs(g32[k].astype(np.float64) - l2 - ref) / scale
            qq = 0.99 if (k.startswith('conv') and k.endswith('bias')) else 0.999
            q_gpu, q_o32 = float(np.quantile(e_gpu, qq)), float(np.quantile(e_o32, qq))
            measured['grad_%s_%s' % (k, tag)] = [q_gpu, float(e_gpu.max())]
            measured['grad_%s_oracle32' % k] = [q_o32, float(e_o32.max())]
            assert q_gpu <= max(5e-5, 3 * q_o32), 'grad %s (%s): %.1f %% quantile %.3e of scale, max %.3e (fp32 oracle %.3e, %.3e)' % (
                k, tag, 100 * qq, q_gpu, e_gpu.max(), q_o32, e_o32.max())
            if not (k.startswith('conv') and k.endswith('bias')):
                assert e_gpu.max() <= max(10 * GREL, 3 * e_o32.max()), 'grad %s (%s): max %.3e (fp32 oracle %.3e)' % (
                    k, tag, e_gpu.max(), e_o32.max())
        # one TF-form Adam step from the float32 oracle's gradients
        p32 = {k: v.copy() for k, v in params.items()}
        state, ill = {}, {}
        R.adam_tf_step(p32, {k: g32[k] for k in params}, state)
        for k in params:
            assert_adam_params_close(after[k], p32[k], state['v/' + k], 0, ill, k, rel=GREL, lr=2e-3, quantile=0.999)
    record_measured('config2_network_b64_vs_oracle[%s]' % dx_form, **measured)
    # fused and unfused differ only in the order of the sum over the 32 filters of the last layer
    lf, lu = results[True][0], results[False][0]
    assert np.abs(lf - lu).max() <= REL * np.abs(lu).max()


def test_internal_vertex_order_is_invisible(ops, dev, monkeypatch):
    """cgcnn relabels the vertices of a network without pooling (vertex_order = 'length': graph.length_order) and keeps
    activations, per-vertex biases and the first FC layer's rows in that order.  Nothing of it shows at the boundary: with
    the same variables (set by name in the reference's shapes) a relabelled and a reference-order model give the same
    logits, the same gradients and the same variables after a training step (1e-5: another summation order inside the
    rows), and a checkpoint of either restores the other bit for bit."""
    import bench
    from gcn_fmri_decoding_amd import models_gcn
    Ls, _ = bench.load_graph(10000, 1, 0, 1, None)
    L = Ls[0]
    M = L.shape[0]
    F, K, p, Mfc, C, B = [8, 8], [4, 3], [1, 1], [16, 5], 3, 6
    nets = {}
    for mode in ('length', 'reference'):
        monkeypatch.setenv('CHEBGCN_VERTEX_ORDER', mode)
        nets[mode] = models_gcn.cgcnn({'device': dev}, [L] * 2, F, K, p, Mfc, filter='chebyshev5', brelu='b2relu', pool='mpool1',
                                      initial='he', channel=C, regularization=5e-4, dropout=1, batch_size=B, verbose=False)
        assert nets[mode].vertex_order == mode
    a, b = nets['length'], nets['reference']
    assert a.graphs[0].ordered and not b.graphs[0].ordered
    rs = np.random.RandomState(3)
    for k in a.variables():
        v = (rs.randn(*a._spec(k).ref_shape) * (0.3 if k.endswith('bias') else 0.05)).astype(np.float32)
        a.set_variable(k, v)
        b.set_variable(k, v)
        assert np.array_equal(a.get_var(k), v) and np.array_equal(b.get_var(k), v)
    x = torch.full((B, C, ops.plane_stride(M)), float('nan'), device=dev)
    x[:, :, :M] = torch.as_tensor(rs.randn(B, C, M).astype(np.float32)).to(dev)
    ld = torch.as_tensor(rs.randint(0, 5, B)).to(dev)
    with torch.no_grad():
        la, lb = a._inference_storage(x, 1).cpu().numpy(), b._inference_storage(x, 1).cpu().numpy()
    assert np.abs(la - lb).max() <= REL * np.abs(lb).max()
    # The order a batch is in is carried by a wrapper (models_gcn.InternalPlanes), never by the tensor: whatever tensor
    # operations produce is a plain tensor = the caller's order; slices / clones OF THE WRAPPER stay in the internal order.
    with torch.no_grad():
        for xv in (x.clone(), x[:], x.contiguous(), x[0:B]):
            assert np.array_equal(a._inference_storage(xv, 1).cpu().numpy(), la)
        raw = torch.as_tensor(rs.randn(B + 2, M, C).astype(np.float32)).to(dev)
        xi = a._gather(raw, torch.arange(B, dtype=torch.int32, device=dev))             # staged: wrapped, internal order
        assert isinstance(xi, models_gcn.InternalPlanes) and xi.shape == x.shape
        li = a._inference_storage(xi, 1).cpu().numpy()
        xr = raw[:B].permute(0, 2, 1).contiguous()                                        # the same batch, caller's order
        xc = torch.zeros_like(x)
        xc[:, :, :M] = xr
        assert np.array_equal(a._inference_storage(xc, 1).cpu().numpy(), li)
        for xv in (xi.clone(), xi[:], xi.detach(), xi[0:B]):
            assert isinstance(xv, models_gcn.InternalPlanes)
            assert np.array_equal(a._inference_storage(xv, 1).cpu().numpy(), li)
        l13 = a._inference_storage(xi[1:3], 1).cpu().numpy()                              # (a smaller launch: other kernel arms)
        assert np.abs(l13 - li[1:3]).max() <= REL * np.abs(li).max()
        with pytest.raises(IndexError):
            xi[:, 0]
        with pytest.raises(ValueError):
            b._inference_storage(xi, 1)              # another model's internal order
        lb2 = b._inference_storage(b._gather(raw, torch.arange(B, dtype=torch.int32, device=dev)), 1).cpu().numpy()
        assert np.abs(li - lb2).max() <= REL * np.abs(lb2).max()
    _, loss_a = a.train_step(x, ld)
    _, loss_b = b.train_step(x, ld)
    assert abs(float(loss_a) - float(loss_b)) <= REL * abs(float(loss_b))
    for k in a.variables():
        ga, gb = a.gradient(k).cpu().numpy(), b.gradient(k).cpu().numpy()
        assert ga.shape == gb.shape == tuple(a._spec(k).ref_shape)
        assert np.abs(ga - gb).max() <= GREL * max(np.abs(gb).max(), 1e-30), k
    sd = a.state_dict()
    b.load_state_dict(sd)
    for k in a.variables():
        assert np.array_equal(a.get_var(k), b.get_var(k)), k
        assert torch.equal(a._ref_view(a._adam_m, k), b._ref_view(b._adam_m, k)), k
    a.load_state_dict(b.state_dict())
    assert all(np.array_equal(a.get_var(k), b.get_var(k)) for k in a.variables())


@pytest.mark.parametrize('N,B,Fin,K,Fout,bias_kind,split', [
    (360, 128, 32, 10, 32, 'vertex', 2),        # the reference's training shape (training.py:34, configure_fmri.py:11, 28): layers 2-6
    (360, 128, 15, 10, 32, 'vertex', 2),        # ... layer 1 (block_dura = 15 input planes)
    (360, 200, 32, 4, 32, 'filter', 1),         # more than 3/4 of a window per CU: one workgroup per window
    (246, 7, 5, 3, 20, 'none', 2),              # M = 260 (eight waves); no bias, no ReLU: plain gradients
    (400, 9, 7, 4, 9, 'vertex', 2),             # M = 422 > 384: not served by the fused kernel (sixteen waves would have 128
                                                # registers per lane; built and measured in round 5: slower than the separate kernels)
])
def test_fused_atlas_layer_vs_oracle(ops, dev, lib, N, B, Fin, K, Fout, bias_kind, split):
    """csrc/fused_small.hip: recurrence + contraction of a layer in one on-chip launch (atlas-sized graphs), and the gradient
    wrt its input (G_j = dy W_j^T feeding the adjoint recurrence) -- against the CPU oracle (oracle/layers_ref.py <->
    lib_new/models_gcn.py:587-629 and its autodiff): output, ReLU mask, the stack it leaves for the weight gradient, dx."""
    from gcn_fmri_decoding_amd import _lib, graph
    from oracle import layers_ref as R
    Ls, _, _ = graph.synthetic_graph(N, k=8, levels=1)
    L = Ls[0]
    g = ops.Graph(L, dev)
    M, Mp = g.M, g.Mp
    if Mp > 384:
        assert lib.chebgcn_fused_layer_supported(g.handle, B, Fin, K, Fout) == 0
        return
    assert lib.chebgcn_fused_layer_supported(g.handle, B, Fin, K, Fout) == 1
    rs = np.random.RandomState(N + B)
    x = rs.randn(B, M, Fin).astype(np.float32)
    W = (rs.randn(Fin * K, Fout) * (0.5 / np.sqrt(Fin * K))).astype(np.float32)
    relu = bias_kind != 'none'
    kind = {'vertex': ops.BIAS_VERTEX, 'filter': ops.BIAS_FILTER, 'none': ops.BIAS_NONE}[bias_kind]
    bref = (rs.randn(1, M if bias_kind == 'vertex' else 1, Fout) * 0.3).astype(np.float32) if relu else np.zeros((1, 1, Fout), np.float32)
    y, T = R.chebyshev5_fwd(x, L, W, K, return_stack=True)               # T: [K, M, Fin, B]
    act = R.brelu_fwd(y, bref) if relu else y
    do = rs.randn(B, M, Fout).astype(np.float32)
    dy = R.brelu_bwd(do, act, bref.shape)[0] if relu else do
    dx_ref, _ = R.chebyshev5_bwd(dy.astype(np.float32), L, W, K, T, need_dx=True)

    xs = torch.full((B, Fin, Mp), float('nan'), device=dev)
    xs[:, :, :M] = torch.as_tensor(np.ascontiguousarray(x.transpose(0, 2, 1))).to(dev)
    Wd = torch.as_tensor(W).to(dev)
    bd = None
    if bias_kind == 'vertex':
        bd = torch.zeros((Fout, Mp), device=dev)
        bd[:, :M] = torch.as_tensor(bref[0].T.copy()).to(dev)
    elif bias_kind == 'filter':
        bd = torch.as_tensor(bref.reshape(-1)).to(dev)
    nws = lib.chebgcn_fused_layer_workspace(g.handle, B, Fin, K, Fout)
    assert (nws > 0) == (split == 2)
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
    stack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    out = torch.full((B, Fout, Mp), float('nan'), device=dev)
    mask = torch.zeros((B, Fout, Mp // 4), dtype=torch.uint8, device=dev)
    nw = 8 if Mp <= 256 else 12
    _lib.check(lib.chebgcn_fused_layer_fwd(g.handle, P(xs), P(Wd), P(bd), kind, P(stack), P(out), P(mask) if relu else None, P(ws), nws,
                                           B, Fin, K, Fout, int(relu), stream()), 'fused_layer_fwd')
    want = 'fused_layer_kernel<%d,%d,false>' % (nw, 16 // split) + (' + fused_combine_kernel' if split == 2 else '')
    assert _lib.last_dispatch() == want, _lib.last_dispatch()
    got = {}
    o = out[:, :, :M].permute(0, 2, 1).cpu().numpy()
    got['out'] = np.abs(o - act).max() / np.abs(y).max()
    assert got['out'] <= REL, 'output: %.3e' % got['out']
    st = stack[:, :, :, :M].permute(0, 3, 2, 1).cpu().numpy()             # [K, M, Fin, B]
    got['stack'] = np.abs(st - T).max() / np.abs(T).max()
    assert got['stack'] <= REL, 'stack: %.3e' % got['stack']
    if relu:
        bits = torch.stack([(mask >> r) & 1 for r in range(4)], -1).reshape(B, Fout, Mp)[..., :M].bool()
        assert torch.equal(bits, out[..., :M] > 0), 'ReLU bit mask disagrees with the output'
    # inference form: no stack is written, same output bits
    out2 = torch.full((B, Fout, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_fused_layer_fwd(g.handle, P(xs), P(Wd), P(bd), kind, None, P(out2), None, P(ws), nws, B, Fin, K, Fout, int(relu),
                                           stream()), 'fused_layer_fwd')
    assert torch.equal(out2[..., :M], out[..., :M])
    dout = torch.full((B, Fout, Mp), float('nan'), device=dev)
    dout[:, :, :M] = torch.as_tensor(np.ascontiguousarray(do.transpose(0, 2, 1))).to(dev)
    if not relu:
        dout[..., M:] = 0.0
    dx = torch.full((B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_fused_layer_bwd_x(g.handle, P(dout), P(mask) if relu else None, P(Wd), P(dx), B, Fin, K, Fout, stream()),
               'fused_layer_bwd_x')
    assert _lib.last_dispatch() == 'fused_layer_kernel<%d,%d,true>' % (nw, 16 // split)
    d = dx[:, :, :M].permute(0, 2, 1).cpu().numpy()
    got['dx'] = np.abs(d - dx_ref).max() / np.abs(dx_ref).max()
    assert got['dx'] <= GREL, 'dx: %.3e' % got['dx']
    record_measured('fused_atlas_layer_vs_oracle[%d,%d,%d,%d,%d]' % (N, B, Fin, K, Fout), **got)


# ------------------------------------------------------------------------------------------------------------------
# bf16 gradients of wide layers with dy handed over as bf16 (chebgcn_relu_grad_bf16 + chebgcn_contract_bwd_*_bf16_dy16)
BF16_REL = 1e-2


@pytest.mark.parametrize('B,M,Fin,K,Fout,bias_kind', [
    (2, 500, 70, 5, 300, 1),         # 2 x 2 workgroup tiles of the one-pass weight gradient (350 rows, 300 columns); per-filter bias
    (3, 77, 33, 5, 65, 2),           # ragged tiles, chunks of 16 vertices with a tail of 13; per-vertex bias
    (70, 40, 60, 5, 256, 0),         # more windows than vertex chunks per window; no bias
    (2, 300, 120, 5, 80, 2),         # 600 rows: bwd_x on five waves with two groups of 320
    (4, 10466, 60, 5, 256, 2)])      # BASELINE configs[4] (batch 4 of 64)
def test_bf16_dy16_gradients(ops, dev, lib, B, M, Fin, K, Fout, bias_kind):
    """ReluGrad writing dy as bf16 and the two one-pass bf16 contraction gradients reading it (models_gcn.py:616, 619-629 under
    TF autodiff): BIT-IDENTICAL to the fp32-dy path (chebgcn_brelu_pool_bwd + chebgcn_contract_bwd_*_bf16, passes = 1), and
    within the bf16 bound of float64 products of the same operands.  Pads of every plane are NaN."""
    from gcn_fmri_decoding_amd import _lib
    assert lib.chebgcn_bf16_dy16_supported(B, M, Fin, K, Fout) == 1
    Mp = ops.plane_stride(M)
    gen = torch.Generator(device=dev)
    gen.manual_seed(B * 1000 + M + Fout)
    stack = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
    gout = torch.randn((B, Fout, Mp), generator=gen, device=dev)
    W = torch.randn((Fin * K, Fout), generator=gen, device=dev) * 0.2
    keep = torch.rand((B, Fout, Mp), generator=gen, device=dev) > 0.4
    stack[..., M:] = float('nan')
    gout[..., M:] = float('nan')
    mask = (keep.reshape(B, Fout, Mp // 4, 4).to(torch.uint8) * torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=dev)).sum(-1).to(torch.uint8).contiguous()
    bshape = {0: None, 1: (Fout,), 2: (Fout, Mp)}[bias_kind]
    nb = lib.chebgcn_brelu_pool_bwd_workspace(B, M, Fout, 1, bias_kind)
    bws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)

    # ---- fp32-dy path (what the round-3 layer ran)
    dy32 = torch.full((B, Fout, Mp), float('nan'), device=dev)
    db32 = torch.full(bshape, float('nan'), device=dev) if bshape else None
    _lib.check(lib.chebgcn_brelu_pool_bwd(P(gout), None, P(mask), P(dy32), P(db32), bias_kind, B, M, Fout, 1, 0, 1, P(bws), nb, stream()), 'brelu_pool_bwd')
    nw = lib.chebgcn_contract_bwd_w_bf16_workspace(B, M, Fin, K, Fout)
    ws = torch.empty(nw, dtype=torch.uint8, device=dev)
    dW32 = torch.full((Fin * K, Fout), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_w_bf16(P(stack), P(dy32), P(dW32), P(ws), nw, B, M, Fin, K, Fout, 1, stream()), 'bwd_w_bf16')
    assert _lib.last_dispatch().startswith('contract_bwd_w_bf16_wide_kernel<1> + ')
    nx = lib.chebgcn_contract_bwd_x_bf16_workspace(Fin, K, Fout)
    wsx = torch.empty(nx, dtype=torch.uint8, device=dev)
    gs32 = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_x_bf16(P(dy32), P(W), P(gs32), B, M, Fin, K, Fout, 1, P(wsx), nx, stream()), 'bwd_x_bf16')
    nw5 = (Fin * K + 319) // 320 * 320 < (Fin * K + 255) // 256 * 256
    assert _lib.last_dispatch() == 'pack_w_bf16_kernel<transposed> + contract_fwd_bf16_kernel<1,%d>' % (5 if nw5 else 4)

    # ---- bf16-dy path
    dy16 = torch.full((B, Fout, Mp), float('nan'), dtype=torch.bfloat16, device=dev)
    db16 = torch.full(bshape, float('nan'), device=dev) if bshape else None
    _lib.check(lib.chebgcn_relu_grad_bf16(P(gout), P(mask), P(dy16), P(db16), bias_kind, B, M, Fout, P(bws), nb, stream()), 'relu_grad_bf16')
    assert 'bias_grad_relu_kernel<' in _lib.last_dispatch() and ',bf16>' in _lib.last_dispatch()
    dW16 = torch.full((Fin * K, Fout), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_w_bf16_dy16(P(stack), P(dy16), P(dW16), P(ws), nw, B, M, Fin, K, Fout, stream()), 'bwd_w_bf16_dy16')
    assert _lib.last_dispatch().startswith('contract_bwd_w_bf16_wide_kernel<1,dy16> + ')
    gs16 = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_contract_bwd_x_bf16_dy16(P(dy16), P(W), P(gs16), B, M, Fin, K, Fout, P(wsx), nx, stream()), 'bwd_x_bf16_dy16')
    assert _lib.last_dispatch() == 'pack_w_bf16_kernel<transposed> + contract_fwd_bf16_kernel<1,%d,x16>' % (5 if nw5 else 4)
    torch.cuda.synchronize()

    assert torch.equal(dy16[..., :M], dy32[..., :M].to(torch.bfloat16)), 'dy16 is not the RNE rounding of the fp32 dy'
    if db32 is not None:
        v = (slice(None), slice(0, M)) if bias_kind == 2 else (slice(None),)
        assert torch.equal(db16[v], db32[v]), 'bias gradient differs'
    assert torch.equal(dW16, dW32), 'weight gradient differs from the fp32-dy path: max %.3e' % float((dW16 - dW32).abs().max())
    assert torch.equal(gs16[..., :M], gs32[..., :M]), 'stack gradient differs from the fp32-dy path'

    # ---- against float64 products of the same operands
    d64 = (gout[..., :M] * keep[..., :M]).double()
    refw = stack[..., :M].double().permute(2, 0, 1, 3).reshape(Fin * K, B * M) @ d64.permute(1, 0, 2).reshape(Fout, B * M).T
    refx = torch.einsum('ro,bom->rbm', W.double(), d64).reshape(Fin, K, B, M).permute(1, 2, 0, 3)
    e_w, e_x = rel_err(dW16, refw), rel_err(gs16[..., :M], refx)
    record_measured('bf16_dy16_gradients[%d,%d,%d,%d,%d]' % (B, M, Fin, K, Fout), bwd_w=e_w, bwd_x=e_x)
    assert e_w <= BF16_REL and e_x <= BF16_REL, 'bwd_w %.3e, bwd_x %.3e' % (e_w, e_x)
    assert e_w > 1e-6 and e_x > 1e-6, 'suspiciously exact: is an fp32 kernel running?'


def test_bf16_layer_takes_the_dy16_path(ops, dev, lib):
    """A wide ReLU layer in 'bf16' precision: ChebConv.backward hands dy over as bf16 (the kernels are named), and every
    gradient is bit-identical to the same layer with ``ops.bf16_dy16 = False``."""
    import scipy.sparse
    from gcn_fmri_decoding_amd import _lib, graph
    rs = np.random.RandomState(4)
    N = 900
    A = scipy.sparse.random(N, N, density=6.0 / N, random_state=rs, format='csr', dtype=np.float32)
    A = A + A.T
    A.setdiag(0)
    A.eliminate_zeros()
    L = graph.laplacian(A.tocsr(), normalized=True)
    g = ops.Graph(L, dev)
    M = L.shape[0]
    B, Fin, K, Fout = 3, 40, 5, 96
    x = rs.randn(B, Fin, g.Mp).astype(np.float32)
    x[..., M:] = 0
    W = torch.as_tensor((rs.randn(Fin * K, Fout) * 0.1).astype(np.float32)).to(dev)
    b = torch.as_tensor((rs.randn(Fout, g.Mp) * 0.1).astype(np.float32)).to(dev)
    gout = torch.zeros((B, Fout, g.Mp), device=dev)
    gout[..., :M] = torch.as_tensor(rs.randn(B, Fout, M).astype(np.float32)).to(dev)
    grads, logs = {}, {}
    old = ops.bf16_dy16
    try:
        for flag in (True, False):
            ops.bf16_dy16 = flag
            xs = torch.as_tensor(x).to(dev).requires_grad_(True)
            Wp, bp = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
            out = ops.cheb_conv(xs, Wp, bp, g, K, relu=True, bias_kind=ops.BIAS_VERTEX, precision='bf16')
            _lib.dispatch_log = []
            out.backward(gout)
            torch.cuda.synchronize()
            logs[flag] = dict(_lib.dispatch_log)
            _lib.dispatch_log = None
            grads[flag] = (xs.grad[..., :M].clone(), Wp.grad.clone(), bp.grad[..., :M].clone())
    finally:
        ops.bf16_dy16 = old
        _lib.dispatch_log = None
    assert 'relu_grad_bf16' in logs[True] and ',bf16>' in logs[True]['relu_grad_bf16']
    assert 'dy16' in logs[True]['contract_bwd_w_bf16'] and 'x16' in logs[True]['contract_bwd_x_bf16']
    assert 'brelu_pool_bwd' in logs[False] and 'dy16' not in logs[False]['contract_bwd_w_bf16']
    for name, a, c in zip(('dx', 'dW', 'dbias'), grads[True], grads[False]):
        assert torch.equal(a, c), '%s differs between the bf16-dy and the fp32-dy layer' % name


@pytest.mark.parametrize('B,C,dtype', [(128, 22, torch.int64), (64, 21, torch.int32), (1, 2, torch.int64), (1000, 7, torch.int32), (3, 1, torch.int64),
                                       (70, 32, torch.int32), (33, 45, torch.int64)])
def test_softmax_xent_vs_float64(ops, dev, B, C, dtype):
    """chebgcn_softmax_xent (tf.nn.sparse_softmax_cross_entropy_with_logits + tf.reduce_mean, models_gcn.py:257-259, and its
    gradient) against float64 NumPy; logits with a wide range (the max-shift must hold); bit-identical from run to run."""
    from gcn_fmri_decoding_amd import _lib
    rs = np.random.RandomState(B + C)
    z = (rs.randn(B, C) * 8).astype(np.float32)
    y = rs.randint(0, C, B)
    zd, yd = torch.as_tensor(z).to(dev), torch.as_tensor(y).to(dev).to(dtype)
    loss, dz = ops.softmax_xent(zd, yd)
    assert _lib.last_dispatch() == 'softmax_xent_kernel<%s>' % ('int64' if dtype == torch.int64 else 'int32')
    z64 = z.astype(np.float64)
    m = z64.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(z64 - m).sum(1))
    ref = float((lse - z64[np.arange(B), y]).mean())
    sm = np.exp(z64 - lse[:, None])
    sm[np.arange(B), y] -= 1
    dref = sm / B
    e_l = abs(float(loss) - ref) / max(abs(ref), 1e-30)
    e_d = np.abs(dz.cpu().numpy() - dref).max() / np.abs(dref).max() if C > 1 else float(np.abs(dz.cpu().numpy()).max())
    record_measured('softmax_xent[%d,%d]' % (B, C), loss=e_l, dlogits=e_d)
    assert e_l <= 1e-6 and e_d <= 2e-6, (e_l, e_d)
    loss2, dz2 = ops.softmax_xent(zd, yd)
    assert torch.equal(loss, loss2) and torch.equal(dz, dz2)
    # a label outside [0, C) (e.g. 1-based labels): NaN loss and NaN gradient for that row, like TensorFlow's GPU kernel --
    # never a finite loss against a clamped class; the other rows keep their gradients
    for bad in (C, -1):
        yb = y.copy()
        yb[B // 2] = bad
        loss3, dz3 = ops.softmax_xent(zd, torch.as_tensor(yb).to(dev).to(dtype))
        assert torch.isnan(loss3).all() and torch.isnan(dz3[B // 2]).all()
        keep = [b for b in range(B) if b != B // 2]
        assert torch.equal(dz3[keep], dz[keep])


@pytest.mark.parametrize('n,dev_scalars', [(1000, False), (700001, True), (2500000, False)])
def test_adam_with_squares_and_loss_bookkeeping(ops, dev, n, dev_scalars):
    """chebgcn_adam_step_sq = chebgcn_adam_step bit for bit, plus the partial sums of squares of the PRE-update variables;
    chebgcn_loss_bookkeeping: loss = ce + reg/2 * sum p^2 (models_gcn.py:262-266), the EMA(0.9) of :269-275 and its debiased
    read, against float64."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(n)
    p0 = torch.randn(n, generator=gen, device=dev) * 0.1
    g = torch.randn(n, generator=gen, device=dev)
    m0 = torch.randn(n, generator=gen, device=dev) * 0.01
    v0 = torch.rand(n, generator=gen, device=dev) * 0.01
    lr_t, reg = 1.7e-3, 5e-4
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    ops.adam_step(pa, g, ma, va, lr_t, grad_scale=0.5, l2=reg)
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    part = torch.full((4096,), float('nan'), device=dev)
    lr = torch.tensor([lr_t], dtype=torch.float32, device=dev) if dev_scalars else lr_t
    nparts = ops.adam_step_sq(pb, g, mb, vb, lr, part, grad_scale=0.5, l2=reg)
    assert nparts == min((n + 255) // 256, 4096)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb), 'adam_step_sq is not adam_step'
    sq_ref = float((p0.double() ** 2).sum())
    e_sq = abs(float(part[:nparts].double().sum()) - sq_ref) / sq_ref
    ce = torch.tensor(2.345, dtype=torch.float32, device=dev)
    ema = torch.tensor(1.25, dtype=torch.float32, device=dev)
    corr = 1.0 / (1 - 0.9 ** 7)
    out = ops.loss_bookkeeping(ce, part, nparts, 0.5 * reg, ema, torch.tensor([corr], dtype=torch.float32, device=dev) if dev_scalars else corr)
    loss_ref = 2.345 + 0.5 * reg * sq_ref
    ema_ref = 1.25 + 0.1 * (loss_ref - 1.25)
    e_ema = abs(float(ema) - ema_ref) / ema_ref
    e_out = abs(float(out) - ema_ref * corr) / (ema_ref * corr)
    record_measured('adam_sq_loss_bookkeeping[%d]' % n, sum_sq=e_sq, ema=e_ema, loss_average=e_out)
    assert e_sq <= 2e-6 and e_ema <= 1e-6 and e_out <= 1e-6, (e_sq, e_ema, e_out)


@pytest.mark.parametrize('n,ordered', [(300, False), (3000, False), (3000, True)])
def test_recurrence_fwd_t_is_the_forward_recurrence_on_the_transposed_operator(ops, dev, lib, n, ordered):
    """chebgcn_recurrence_fwd_t: T_k(L~^T) x -- on a NON-symmetric operator (a Laplacian is symmetric and would not tell L~ from
    its transpose), against float64, copy and in place; with it the gradient of a layer wrt its input is
    sum_k [T_k(L~^T) dy] W_k^T (TF autodiff of models_gcn.py:598-617), which the layer tests hold to the oracle."""
    import scipy.sparse as sp
    from gcn_fmri_decoding_amd import _lib, graph
    rs = np.random.RandomState(n)
    rows = np.repeat(np.arange(n), 6)
    cols = rs.randint(0, n, rows.size)
    A = sp.coo_matrix((rs.rand(rows.size).astype(np.float32) * 0.3, (rows, cols)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    if ordered:
        # rows AND columns sorted by descending length are what the ordered image needs: symmetric pattern, asymmetric values
        A = sp.csr_matrix(((A + A.T) != 0).astype(np.float32))
        A.data = (rs.rand(A.nnz).astype(np.float32) * 0.3)
    L = (A + sp.identity(n, dtype=np.float32, format='csr')).tocsr()          # rescale_L subtracts the identity again: L~ = A
    order = graph.length_order(L) if ordered else None
    g = ops.Graph(L, dev, order=order)
    assert g.ordered == ordered
    if ordered:
        L = graph.permute(L, order)
    indptr, indices, data = graph.rescaled_laplacian_csr(L)
    Lt = sp.csr_matrix((data.astype(np.float64), indices, indptr), shape=(n, n))
    assert abs(Lt - Lt.T).max() > 1e-3                                         # really not symmetric
    B, Fin, K, Mp = 3, 5, 4, g.Mp
    x = torch.full((B, Fin, Mp), float('nan'), device=dev)
    x[:, :, :n] = torch.as_tensor(rs.randn(B, Fin, n).astype(np.float32)).to(dev)
    stack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_recurrence_fwd_t(g.handle, P(x), P(stack), B, Fin, K, stream()), 'fwd_t')
    name = _lib.last_dispatch()
    X = x[:, :, :n].cpu().numpy().astype(np.float64).reshape(B * Fin, n).T
    T = [X, Lt.T @ X]
    for k in range(2, K):
        T.append(2 * (Lt.T @ T[-1]) - T[-2])
    ref = np.stack(T).transpose(0, 2, 1).reshape(K, B, Fin, n)
    e_t = np.abs(stack[..., :n].cpu().numpy() - ref).max() / np.abs(ref).max()
    wrong = Lt @ X                                                             # the untransposed operator must NOT match
    assert np.abs(stack[1, ..., :n].cpu().numpy().reshape(B * Fin, n).T - wrong).max() > 1e-3 * np.abs(wrong).max()
    s2 = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
    s2[0].copy_(x)
    _lib.check(lib.chebgcn_recurrence_fwd_t(g.handle, P(s2), P(s2), B, Fin, K, stream()), 'fwd_t in place')
    assert torch.equal(s2[..., :n], stack[..., :n])
    # the adjoint identity ties it to chebgcn_recurrence_fwd: <T(L~) u, v_k> summed over k  ==  <u, sum_k T_k(L~^T) v_k>
    record_measured('recurrence_fwd_t[%d,%s]' % (n, ordered), kernel=name, err=e_t)
    assert e_t <= REL, (name, e_t)
    assert name.endswith('false>') or 'false,' in name, name                     # a FORWARD kernel template, on the image of L~^T


def test_relu_grad_mean_writes_the_gated_gradient(ops, dev, lib):
    """chebgcn_relu_grad_mean: dy[b][o][m] = mask bit ? gmean[b][m] : 0 and the bias gradient in one pass (the last layer under
    the fused feature mean, models_gcn.py:673, when its input gradient is formed by the forward recurrence on dy)."""
    from gcn_fmri_decoding_amd import _lib
    B, M, F = 5, 1001, 7
    Mp = ops.plane_stride(M)
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    gm = torch.zeros((B, Mp), device=dev)
    gm[:, :M] = torch.randn((B, M), generator=gen, device=dev)
    mask = torch.randint(0, 16, (B, F, Mp // 4), dtype=torch.uint8, device=dev)
    bits = torch.stack([(mask >> r) & 1 for r in range(4)], -1).reshape(B, F, Mp).bool()
    for kind, shape in ((ops.BIAS_VERTEX, (F, Mp)), (ops.BIAS_FILTER, (F,))):
        dy = torch.full((B, F, Mp), float('nan'), device=dev)
        db = torch.full(shape, float('nan'), device=dev)
        n = lib.chebgcn_brelu_pool_bwd_workspace(B, M, F, 1, kind)
        ws = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        _lib.check(lib.chebgcn_relu_grad_mean(P(gm), P(mask), P(dy), P(db), kind, B, M, F, P(ws), n, stream()), 'relu_grad_mean')
        ref = torch.where(bits, gm[:, None, :].expand(B, F, Mp), torch.zeros((), device=dev))
        assert torch.equal(dy[..., :M], ref[..., :M])
        if kind == ops.BIAS_VERTEX:
            dbr = ref[..., :M].double().sum(0)
            assert float((db[:, :M].double() - dbr).abs().max()) <= 1e-6 * float(dbr.abs().max()) and float(db[:, M:].abs().sum()) == 0.0
        else:
            dbr = ref[..., :M].double().sum((0, 2))
            assert float((db.double() - dbr).abs().max()) <= 1e-6 * float(dbr.abs().max())


@pytest.mark.parametrize('Fin,K,Fout', [(32, 5, 32), (15, 5, 32), (64, 25, 64), (60, 5, 256), (3, 1, 7)])
def test_reindex_weights_is_the_index_map(ops, dev, lib, Fin, K, Fout):
    """chebgcn_reindex_weights: Wt[fo*K + k][fin] = W[fin*K + k][fo], bit for bit (the weights of the contraction that forms
    the input gradient from the stack of dy)."""
    import torch
    from gcn_fmri_decoding_amd import _lib
    W = torch.randn((Fin * K, Fout), device=dev)
    Wt = torch.full((Fout * K, Fin), float('nan'), device=dev)
    _lib.check(lib.chebgcn_reindex_weights(ops._p(W), ops._p(Wt), Fin, K, Fout, ops._stream()), 'reindex_weights')
    assert _lib.last_dispatch() == 'reindex_weights_kernel'
    want = W.view(Fin, K, Fout).permute(2, 1, 0).reshape(Fout * K, Fin)
    assert torch.equal(Wt, want)
    assert lib.chebgcn_reindex_weights(ops._p(W), ops._p(W), Fin, K, Fout, ops._stream()) != 0      # in place: refused
