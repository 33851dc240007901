"""PyTorch-ROCm face of libchebgcn.so: device graph handles, plane-layout helpers and the
``torch.autograd.Function`` wrappers around the HIP kernels.  PyTorch supplies device
memory, streams and autograd bookkeeping; every arithmetic op of the graph-convolution
path runs in the library (there is no torch / CPU fallback -- a missing library or a
failing call raises).

Plane layout: the reference's activation ``x[N, M, F]`` (models_gcn.py:588) is kept as a
contiguous *storage* tensor ``[N, F, Mp]`` (vertex axis fastest, ``Mp = plane_stride(M)``).
``plane_view(storage, M)`` exposes it with the reference's logical shape ``[N, M, F]``
without copying; ``plane_storage(x)`` goes back (zero-copy when ``x`` is such a view).
"""
import ctypes as C
import os
import weakref

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from . import graph as _graph
from ._lib import BIAS_FILTER, BIAS_NONE, BIAS_VERTEX, POOL_AVG, POOL_MAX, plane_stride


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class KernelTimers:
    """Optional per-launch HIP-event timing of the hot kernels (used by bench.py for the
    roofline line).  Events are recorded on the stream the kernel is launched on; nothing
    is synchronised until ``summary()``.  Disabled (``timers is None``) by default."""

    def __init__(self, every=1, by_dispatch=False):
        self.records = {}           # name -> list of (start, end, algorithmic bytes, flops)
        # by_dispatch: one entry per (op, kernel template chebgcn_last_dispatch() reported) instead of one per op -- a
        # network whose layers differ in size runs the same op on different kernels
        self.by_dispatch = bool(by_dispatch)
        # An event pair per launch costs ~5 % of a training step (the markers keep consecutive
        # kernels from overlapping): with ``every = n`` only every n-th step is instrumented.
        # The caller announces steps with ``next_step()``.
        self.every = max(1, int(every))
        self.steps = 0
        self.sampled_steps = 0
        self.active = True

    def next_step(self):
        self.active = self.steps % self.every == 0
        self.sampled_steps += int(self.active)
        self.steps += 1

    def launch(self, name, nbytes, flops, fn):
        if not self.active:
            return fn()
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record()
        rc = fn()
        end.record()
        if self.by_dispatch:
            name = '%s | %s' % (name, _lib.last_dispatch())
        self.records.setdefault(name, []).append((start, end, nbytes, flops))
        return rc

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _, _ in recs]
            out[name] = {'launches': len(recs), 'total_ms': float(sum(ms)), 'avg_ms': float(sum(ms) / len(ms)),
                         'bytes': float(sum(r[2] for r in recs)), 'flops': float(sum(r[3] for r in recs))}
        return out


timers = None
bf16_dy16 = True             # 'bf16' wide layers: dy handed to the two contraction gradients as bf16 (chebgcn_relu_grad_bf16); same results
fold_relu_grad = True        # pool == 1 layers: ReluGrad inside chebgcn_contract_bwd_*_relu (False: separate brelu_pool_bwd pass)
# atlas-sized graphs: recurrence + contraction (and the gradient wrt the input) as one on-chip launch per layer
fused_small = os.environ.get('CHEBGCN_FUSED_SMALL', '1') != '0' 
# contract_bwd_w on a second stream beside contract_bwd_x / recurrence_bwd (dW feeds neither).  'auto' (default since round 6):
# only for layers of more than 32 filters (both gradients are long matrix-bound kernels that share a CU well -- config-4 /
# config-5 layer 2 % faster with it in split bf16, 8 % in fp32).  For 32-filter layers the weight gradient cannot start
# beside the recurrence (one 160 KB workgroup per CU) and then shares HBM with the input gradient's contraction: measured in
# round 6 with the weight gradient on two workgroups per CU, same box -- configs[1] 3.14 ms with the second stream, 3.08 without;
# the captured atlas step (N = 360) 0.82 -> 0.76 ms, N = 1000 1.95 -> 1.88 ms (EXPERIMENTS 8.6).  True / False force it.
# Never on instrumented steps, whose per-kernel event times must not include a neighbour.
overlap_bwd_w = {'1': True, '0': False}.get(os.environ.get('CHEBGCN_OVERLAP_BWD_W', 'auto'), 'auto')
# d(loss)/dx of a layer with Fout <= Fin as  sum_k [T_k(L~^T) dy] W_k^T  -- the FORWARD recurrence on the planes of dy
# (chebgcn_recurrence_fwd_t, in place in slab 0 of the gradient stack), then the forward contraction kernel with the re-indexed
# weights -- instead of chebgcn_contract_bwd_x + chebgcn_recurrence_bwd (the same sum in Clenshaw form): as many bytes, on the two
# faster kernels (the stack of dy is written by a recurrence at 0.49 and read by a contraction at 0.64 of the HBM roofline
# instead of written by one at 0.55 and read by one at 0.43); False keeps the Clenshaw form everywhere
dx_by_forward = os.environ.get('CHEBGCN_DX_BY_FORWARD', '1') != '0'
bias_side_small = True       # fused atlas-size layers: the bias reduction on the second stream as well
_side_streams = {}


def _side_stream(dev):
    key = torch.device(dev).index
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


def _launch(name, nbytes, flops, fn):
    """Run one C-ABI call, optionally bracketed by HIP events."""
    if timers is None:
        return fn()
    return timers.launch(name, nbytes, flops, fn)


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.ChebgcnError('chebgcn ops need ROCm device tensors; got a %s tensor '
                                    '(there is no CPU path)' % t.device)


# ------------------------------------------------------------------------------------
# device graph
# ------------------------------------------------------------------------------------

class Graph:
    """Device image of one Laplacian: ``rescale_L(L, lmax=2)`` and its transpose, as the
    constant sparse operand of the recurrence (models_gcn.py:590-596)."""

    def __init__(self, L, device=None, planes=0, order=None):
        """``planes``: planes a recurrence workgroup carries on chip -- 0 = automatic, 2 or 4
        (chebgcn_graph_create_planes; a choice of speed, not of results).  ``order``: relabel the vertices first
        (``graph.permute(L, order)``; with ``graph.length_order(L)`` the library runs its ordered recurrence kernels,
        ``query(12) == 1``): every plane this graph is used with is then in that vertex order."""
        self.M = int(L.shape[0])
        self.Mp = plane_stride(self.M)
        if not planes:
            planes = int(os.environ.get('CHEBGCN_PLANES', '0'))      # A/B experiments (tools/ab_bench.sh): 2 or 4 for every graph
        if order is not None:
            L = _graph.permute(L, np.asarray(order, np.int64))
        indptr, indices, data = _graph.rescaled_laplacian_csr(L)
        self.nnz = int(len(data))
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = _lib.lib().chebgcn_graph_create_planes(self.M, self.nnz, indptr.ctypes.data_as(C.c_void_p),
                                                        indices.ctypes.data_as(C.c_void_p),
                                                        data.ctypes.data_as(C.c_void_p), int(planes), C.byref(handle))
        _lib.check(rc, 'graph_create')
        self.handle = handle
        self._ordered = None
        self._finalizer = weakref.finalize(self, _lib.lib().chebgcn_graph_destroy, handle)

    def query(self, what):
        v = C.c_int64()
        _lib.check(_lib.lib().chebgcn_graph_query(self.handle, what, C.byref(v)), 'graph_query')
        return v.value

    @property
    def on_chip(self):
        return bool(self.query(3))

    @property
    def ordered(self):
        """Rows sorted by descending length and a kernel shape that serves the graph: the recurrence runs on the ordered
        kernels (csrc/recurrence_ord_kernel.h)."""
        if self._ordered is None:
            self._ordered = bool(self.query(12))
        return self._ordered


_graph_cache = weakref.WeakValueDictionary()


def graph_for(L, device=None):
    """Cache of device graphs keyed by the identity of the SciPy matrix and the device."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = (id(L), dev)
    g = _graph_cache.get(key)
    if g is None or g.M != L.shape[0]:
        g = Graph(L, torch.device('cuda', dev))
        _graph_cache[key] = g
        # keep the graph alive as long as the matrix object is
        try:
            L._chebgcn_graphs = getattr(L, '_chebgcn_graphs', {})
            L._chebgcn_graphs[dev] = g
        except AttributeError:
            pass
    return g


# ------------------------------------------------------------------------------------
# plane layout helpers
# ------------------------------------------------------------------------------------

def plane_empty(B, F, M, device, zero=False):
    Mp = plane_stride(M)
    return (torch.zeros if zero else torch.empty)((B, F, Mp), dtype=torch.float32, device=device)


def plane_view(storage, M):
    """[B, F, Mp] storage -> logical [B, M, F] view (no copy)."""
    return storage[:, :, :M].permute(0, 2, 1)


def _is_plane_view(x):
    if x.dim() != 3 or x.dtype != torch.float32:
        return False
    B, M, F = x.shape
    Mp = plane_stride(M)
    if x.stride() != (F * Mp, 1, Mp):
        return False
    need = (x.storage_offset() + B * F * Mp) * 4
    return x.untyped_storage().nbytes() >= need


def plane_storage(x):
    """Logical [B, M, F] tensor -> contiguous [B, F, Mp] storage.  Zero-copy for tensors
    produced by ``plane_view``; otherwise one layout-change kernel (to_plane)."""
    _require_cuda(x)
    B, M, F = x.shape
    Mp = plane_stride(M)
    if _is_plane_view(x):
        return x.as_strided((B, F, Mp), (F * Mp, Mp, 1))
    return ToPlane.apply(x)


class ToPlane(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        B, M, F = x.shape
        out = plane_empty(B, F, M, x.device)
        _lib.check(_lib.lib().chebgcn_to_plane(_p(x), _p(out), B, M, F, _stream()), 'to_plane')
        ctx.shape = (B, M, F)
        return out

    @staticmethod
    def backward(ctx, g):
        B, M, F = ctx.shape
        g = g.contiguous()
        out = torch.empty((B, M, F), dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().chebgcn_from_plane(_p(g), _p(out), B, M, F, _stream()), 'from_plane')
        return out


def from_plane(storage, M):
    """[B, F, Mp] storage -> contiguous [B, M, F] tensor in the reference layout."""
    storage = storage.contiguous()
    B, F, Mp = storage.shape
    out = torch.empty((B, M, F), dtype=torch.float32, device=storage.device)
    _lib.check(_lib.lib().chebgcn_from_plane(_p(storage), _p(out), B, M, F, _stream()), 'from_plane')
    return out


def perm_data(x, perm, sample=None, out=None):
    """GPU ``coarsening.perm_data_3d`` (+ batch gather): x[S_total, N, F] (device, row
    layout), perm int32 [M] on device, optional sample indices -> storage [S, F, Mp]."""
    _require_cuda(x, perm, sample)
    x = x.contiguous()
    S_total, N, F = x.shape
    M = int(perm.numel())
    S = S_total if sample is None else int(sample.numel())
    if out is None:
        out = plane_empty(S, F, M, x.device)
    _lib.check(_lib.lib().chebgcn_perm_data(_p(x), _p(perm), _p(sample), _p(out), S, N, M, F, _stream()), 'perm_data')
    return out


# ------------------------------------------------------------------------------------
# the graph-convolution layer
# ------------------------------------------------------------------------------------

_workspaces = {}
# set by cgcnn._capture_step around a capture: scratch allocated while capturing comes from THAT graph's private pool and is
# never handed to another graph (or to eager code on a stream with the same handle)
capture_tag = None


def _cap():
    return capture_tag if (capture_tag is not None and torch.cuda.is_current_stream_capturing()) else None


def _workspace(nbytes, device, tag=''):
    """Scratch of a library call, kept per device, stream and use: launches on one stream are ordered, so the buffer of the
    previous call of the same kind is free by the time the next one runs."""
    key = (device.index, torch.cuda.current_stream().cuda_stream, tag, _cap())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


_mean_grad_buffers = {}


def _mean_grad_buffer(B, Mp, device):
    """[B, Mp] scratch of the fused last layer's backward; columns beyond M stay zero (only [:, :M] is ever written)."""
    key = (B, Mp, device.index, torch.cuda.current_stream().cuda_stream, _cap())
    buf = _mean_grad_buffers.get(key)
    if buf is None:
        buf = _mean_grad_buffers[key] = torch.zeros((B, Mp), dtype=torch.float32, device=device)
    return buf


def softmax_xent(logits, labels):
    """Mean softmax cross-entropy of ``logits [B, C]`` against ``labels [B]`` (int32 or int64) and its gradient wrt the
    logits, one launch (tf.nn.sparse_softmax_cross_entropy_with_logits + tf.reduce_mean, models_gcn.py:257-259).  Returns
    (loss: 0-d tensor, dlogits [B, C]); the caller seeds autograd with ``logits.backward(dlogits)``."""
    _require_cuda(logits, labels)
    if logits.dim() != 2 or logits.dtype != torch.float32 or labels.dim() != 1 or labels.numel() != logits.shape[0]:
        raise ValueError('softmax_xent: logits [B, C] float32 and labels [B]')
    if labels.dtype not in (torch.int32, torch.int64):
        labels = labels.long()
    z = logits.detach().contiguous()
    labels = labels.contiguous()
    B, C = z.shape
    loss = torch.empty((), dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z)
    _lib.check(_lib.lib().chebgcn_softmax_xent(_p(z), _p(labels), int(labels.dtype == torch.int64), _p(loss), _p(dz), B, C,
                                               _stream()), 'softmax_xent')
    return loss, dz


def cache_keys():
    """Keys of the per-stream scratch caches above (workspaces, the fused last layer's gradient buffer)."""
    return {('ws',) + k for k in _workspaces} | {('mg',) + k for k in _mean_grad_buffers}


def drop_cache_keys(keys):
    """Forget scratch buffers -- those a captured step allocated from its graph's private pool on its capture streams
    (cgcnn.enable_step_graph records them): kept here they would pin that pool after the graph is gone, and a later capture
    on the same stream would be handed a buffer of a destroyed graph."""
    for k in keys:
        (_workspaces if k[0] == 'ws' else _mean_grad_buffers).pop(tuple(k[1:]), None)


def _brelu_bwd_ws(B, M, F, pool, bias_kind, device):
    """Scratch of chebgcn_brelu_pool_bwd: the per-workgroup partials of a per-filter (b1relu) bias gradient."""
    n = _lib.lib().chebgcn_brelu_pool_bwd_workspace(B, M, F, pool, bias_kind)
    if n == 0:
        return None, 0
    ws = torch.empty(n, dtype=torch.uint8, device=device)
    return ws, n


def _check_grad_buffer(buf, shape, what):
    if tuple(buf.shape) != tuple(shape) or not buf.is_contiguous() or buf.dtype != torch.float32 or not buf.is_cuda:
        raise ValueError('%s buffer must be a contiguous float32 device tensor of shape %s' % (what, tuple(shape)))


PRECISIONS = {'f32': 0, 'bf16': 1, 'bf16x3': 3}     # contraction arithmetic -> passes of chebgcn_contract_fwd_bf16
FP32_MFMA_BALANCE = 19.6        # flop per HBM byte at which the fp32-input matrix cores (157 TFLOP/s) meet 8 TB/s


def resolve_precision(precision, Fin, K, Fout):
    """'auto' -> the arithmetic a layer of this shape computes in: 'f32' (fp32-input matrix instructions, exact products)
    where the contraction is HBM-bound anyway (what BASELINE configs[1] runs: 32 filters), 'bf16x3' where fp32 matrix
    work would bound it -- more than 32 filters and an arithmetic intensity 2*Fin*K*Fout / (4*(Fin*K + Fout)) above the
    machine balance of the fp32 matrix cores.  'bf16x3' splits each fp32 operand into two bf16 and accumulates
    hi*hi + hi*lo + lo*hi in fp32: 3e-6 ... 6e-6 of the tensor's scale from the fp32 layer (tests/test_gpu_dispatch.py holds
    every arm to 1e-5 against float64), at a third of the time.  Anything else is returned unchanged."""
    if precision != 'auto':
        return precision
    ai = float(Fin) * K * Fout / (2.0 * (Fin * K + Fout))
    return 'bf16x3' if (Fout > 32 and ai > FP32_MFMA_BALANCE) else 'f32'


def contract_fwd_into(stack, W, bias, bias_kind, out, argmax, B, M, Fin, K, Fout, pool, pool_kind, relu, precision='f32',
                      what='contract_fwd', gate=None):
    """Launches the forward contraction (models_gcn.py:611-648) into ``out``.

    precision 'f32' = chebgcn_contract_fwd (exact fp32 MFMA); 'bf16' / 'bf16x3' =
    chebgcn_contract_fwd_bf16 with 1 / 3 passes (wide layers, BASELINE config 5).
    ``gate``: a ReLU mask [B, Fout, Mp/4]; the result is stored gated by it (chebgcn_contract_fwd_gated: fp32, no bias / ReLU /
    pooling -- the input gradient in forward form with the ReluGrad of the layer below in its epilogue)."""
    lib = _lib.lib()
    if precision not in PRECISIONS:
        raise ValueError('precision must be one of %s' % sorted(PRECISIONS))
    Mo = M // pool
    nbytes, flops = 4.0 * B * (M * Fin * K + Mo * Fout), 2.0 * B * M * Fin * K * Fout
    if gate is not None:
        if precision != 'f32' or bias is not None or pool != 1 or relu or argmax is not None:
            raise ValueError('contract_fwd_into(gate=...): fp32, no bias, no ReLU, no pooling')
        _lib.check(_launch(what, nbytes + 0.25 * B * M * Fout, flops,
                           lambda: lib.chebgcn_contract_fwd_gated(_p(stack), _p(W), _p(gate), _p(out), B, M, Fin, K, Fout,
                                                                  _stream())), what)
        return
    if precision == 'f32':
        _lib.check(_launch(what, nbytes, flops,
                           lambda: lib.chebgcn_contract_fwd(_p(stack), _p(W), _p(bias), bias_kind, _p(out), _p(argmax), B, M,
                                                            Fin, K, Fout, pool, pool_kind, int(relu), _stream())),
                   what)
        return
    nws = lib.chebgcn_contract_fwd_bf16_workspace(Fin, K, Fout)
    ws = _workspace(nws, stack.device, 'fwd_bf16')
    _lib.check(_launch(what + '_' + precision, nbytes, flops,
                       lambda: lib.chebgcn_contract_fwd_bf16(_p(stack), _p(W), _p(bias), bias_kind, _p(out), _p(argmax), B, M,
                                                             Fin, K, Fout, pool, pool_kind, int(relu), PRECISIONS[precision],
                                                             _p(ws), nws, _stream())),
               what + '_bf16')


def _grad_mode(bufs):
    """Grad mode of the caller of ``cheb_conv`` (``Buffers.grad_mode``); a bare ``ChebConv.apply`` without buffers: on."""
    return True if bufs is None else bool(bufs.grad_mode)


class ChebConv(torch.autograd.Function):
    """y = pool(act(sum_k T_k(L~) x W_k + bias)) on plane storage tensors.

    forward : recurrence_fwd (models_gcn.py:598-610) + contract_fwd (:611-648)
    backward: brelu_pool_bwd, contract_bwd_w, contract_bwd_x, recurrence_bwd; for pool == 1 layers with
              ReLU the ReluGrad runs inside contract_bwd_w_relu / contract_bwd_x_relu on the bit mask the
              forward left, and brelu_pool_bwd only reduces the bias gradient
    ``bufs`` is a ``Buffers`` holder (kept out of autograd's sight): ``bufs.stack`` is an
    optional preallocated [K, B, Fin, Mp] buffer -- when ``x`` already is its slab 0 no copy
    of T_0 is made; ``bufs.out`` optionally receives the result, e.g. slab 0 of the next
    layer's stack.
    """

    @staticmethod
    def forward(ctx, x, W, bias, graph, K, pool, pool_kind, relu, bias_kind, bufs):
        _require_cuda(x, W, bias)
        stack, out = (bufs.stack, bufs.out) if bufs is not None else (None, None)
        lib = _lib.lib()
        x = x if x.is_contiguous() else x.contiguous()
        B, Fin, Mp = x.shape
        M = graph.M
        if Mp != graph.Mp:
            raise ValueError('activation plane stride %d does not match the graph (%d vertices)' % (Mp, M))
        FinK, Fout = W.shape
        if FinK != Fin * K:
            raise ValueError('weight rows %d != Fin*K = %d' % (FinK, Fin * K))
        Wc = W.detach().contiguous()
        Mo = M // pool
        mean = bool(bufs is not None and bufs.mean)
        precision = getattr(bufs, 'precision', 'f32') if bufs is not None else 'f32'
        # atlas-sized graphs (<= 384 vertices, Fin, Fout <= 32, no pooling): the whole layer in one on-chip launch
        # (csrc/fused_small.hip); the stack is written only when a weight gradient will read it
        if (fused_small and pool == 1 and precision == 'f32' and not mean
                and lib.chebgcn_fused_layer_supported(graph.handle, B, Fin, K, Fout)):
            return ChebConv._forward_fused(ctx, x, Wc, bias, graph, K, relu, bias_kind, bufs, stack, out)
        if stack is None:
            stack = torch.empty((K, B, Fin, Mp), dtype=torch.float32, device=x.device)
        # algorithmic bytes: recurrence 4*M*Fin*K per window; the contraction's
        # compulsory traffic 4*(M*Fin*K + M*Fout/pool) per window, flops 2*M*Fin*K*Fout
        _lib.check(_launch('recurrence_fwd', 4.0 * M * Fin * K * B, 0.0, lambda: lib.chebgcn_recurrence_fwd(
            graph.handle, _p(x), _p(stack), B, Fin, K, _stream())), 'recurrence_fwd')
        if mean:
            if not conv_mean_supported(B, M, Fin, K, Fout, pool, relu, getattr(bufs, 'precision', 'f32')):
                raise ValueError('cheb_conv(mean=True): layer not served (ops.conv_mean_supported)')
            wants_grad = _grad_mode(bufs) and any(ctx.needs_input_grad[:3])
            mask = torch.empty((B, Fout, Mp // 4), dtype=torch.uint8, device=x.device) if wants_grad else None
            y = torch.empty((B, Mp), dtype=torch.float32, device=x.device)
            b = bias.detach().contiguous() if bias is not None else None
            _lib.check(_launch('contract_fwd', 4.0 * B * M * (Fin * K + 1), 2.0 * B * M * Fin * K * Fout,
                               lambda: lib.chebgcn_contract_fwd_mean(_p(stack), _p(Wc), _p(b), bias_kind, _p(y), _p(mask), B, M,
                                                                     Fin, K, Fout, _stream())), 'contract_fwd_mean')
            ctx.save_for_backward(stack, Wc, None, mask)
            ctx.fold, ctx.mean, ctx.fused = True, True, False
            ctx.graph, ctx.cfg = graph, (B, M, Fin, K, Fout, pool, pool_kind, int(relu), bias_kind)
            ctx.bias_shape = None if bias is None else tuple(bias.shape)
            ctx.grad_bufs = (bufs.dW, bufs.dbias)
            ctx.done = bufs.done
            ctx.precision = 'f32'
            ctx.pool_maps = None
            ctx.Wt = bufs.Wt
            ctx.link_in, ctx.link_out = bufs.link_in, None
            return y[:, :M]                   # the logical [B, M] mean (row stride Mp): its gradient arrives dense
        if out is None:
            out = plane_empty(B, Fout, Mo, x.device)
        else:
            out = out.detach()                # fresh alias: an output, not an input, for autograd
            if tuple(out.shape) != (B, Fout, plane_stride(Mo)) or not out.is_contiguous():
                raise ValueError('out buffer has the wrong shape')
        argmax = None
        maps = bufs.pool_maps if (bufs is not None and pool > 1) else None
        wants_grad = _grad_mode(bufs) and any(ctx.needs_input_grad[:3])       # inference: no mask is written
        if maps is not None:
            # pooling between two vertex orders: the contraction (bias, ReLU) leaves the unpooled result in the source order,
            # one more pass gathers the clusters through LDS (csrc/pointwise.hip pool_gather_fwd_kernel)
            b = bias.detach() if bias is not None else None
            if b is not None and not b.is_contiguous():
                b = b.contiguous()
            y_full = plane_empty(B, Fout, M, x.device)
            contract_fwd_into(stack, Wc, b, bias_kind, y_full, None, B, M, Fin, K, Fout, 1, pool_kind, relu, precision)
            sel = torch.empty(out.shape, dtype=torch.uint8, device=x.device) if wants_grad else None
            _lib.check(_launch('pool_gather_fwd', B * Fout * (4.0 * (M + Mo) + Mo), 0.0, lambda: lib.chebgcn_pool_gather_fwd(
                _p(y_full), _p(maps[0]), _p(out), _p(sel), B, M, Fout, pool, pool_kind, int(relu), _stream())), 'pool_gather_fwd')
            ctx.save_for_backward(stack, Wc, None, sel)
            ctx.fold, ctx.mean, ctx.fused = False, False, False
            ctx.graph, ctx.cfg = graph, (B, M, Fin, K, Fout, pool, pool_kind, int(relu), bias_kind)
            ctx.bias_shape = None if bias is None else tuple(bias.shape)
            ctx.grad_bufs = (bufs.dW, bufs.dbias)
            ctx.done = bufs.done
            ctx.precision = precision
            ctx.pool_maps = maps
            ctx.Wt = bufs.Wt
            ctx.link_in, ctx.link_out = bufs.link_in, None
            return out
        if pool > 1 and (pool_kind == POOL_MAX or relu):
            argmax = torch.empty(out.shape, dtype=torch.uint8, device=x.device)
        # pool == 1 with ReLU: contract_fwd leaves a bit per vertex (the ReLU mask) and the gradients of the
        # contraction gate the incoming gradient themselves -- no dy tensor, no pass over `out` in backward
        fold = bool(fold_relu_grad and pool == 1 and relu and precision == 'f32')
        if pool == 1 and relu and wants_grad:
            # the mask also serves the separate ReluGrad pass (bf16 gradients): a byte per four vertices instead of `out`
            argmax = torch.empty((B, Fout, Mp // 4), dtype=torch.uint8, device=x.device)
        b = bias.detach() if bias is not None else None
        if b is not None and not b.is_contiguous():
            b = b.contiguous()
        contract_fwd_into(stack, Wc, b, bias_kind, out, argmax, B, M, Fin, K, Fout, pool, pool_kind, relu, precision)
        ctx.save_for_backward(stack, Wc, None if (pool == 1 and relu) else out, argmax)
        ctx.fold, ctx.mean, ctx.fused = fold, False, False
        ctx.graph, ctx.cfg = graph, (B, M, Fin, K, Fout, pool, pool_kind, int(relu), bias_kind)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        ctx.grad_bufs = (bufs.dW, bufs.dbias) if bufs is not None else (None, None)
        ctx.done = bufs.done if bufs is not None else None
        ctx.precision = precision
        ctx.pool_maps = None
        ctx.Wt = bufs.Wt if bufs is not None else None
        ctx.link_in = bufs.link_in if bufs is not None else None
        ctx.link_out = lo = bufs.link_out if bufs is not None else None
        if lo is not None:
            # this layer's backward takes its gated dy as slab 0 of the stack its recurrence_fwd_t fills (the by_fwd arm below)
            lo.gstack = None
            ok = bool(gate_links and pool == 1 and relu and wants_grad and argmax is not None and precision != 'bf16'
                      and dx_by_forward and ctx.needs_input_grad[0] and K > 1 and Fout <= Fin and graph.ordered)
            lo.mask, lo.shape = (argmax, (K, B, Fout, Mp)) if ok else (None, None)
        return out

    @staticmethod
    def _forward_fused(ctx, x, Wc, bias, graph, K, relu, bias_kind, bufs, stack, out):
        lib = _lib.lib()
        B, Fin, Mp = x.shape
        M, Fout = graph.M, Wc.shape[1]
        need_w = bool(_grad_mode(bufs) and ctx.needs_input_grad[1])
        wants_grad = _grad_mode(bufs) and any(ctx.needs_input_grad[:3])
        if not need_w:
            stack = None                          # nothing will read it (inference, frozen weights): it is never written
        elif stack is None:
            stack = torch.empty((K, B, Fin, Mp), dtype=torch.float32, device=x.device)
        if out is None:
            out = plane_empty(B, Fout, M, x.device)
        else:
            out = out.detach()
            if tuple(out.shape) != (B, Fout, Mp) or not out.is_contiguous():
                raise ValueError('out buffer has the wrong shape')
        mask = torch.empty((B, Fout, Mp // 4), dtype=torch.uint8, device=x.device) if (relu and wants_grad) else None
        b = bias.detach() if bias is not None else None
        if b is not None and not b.is_contiguous():
            b = b.contiguous()
        nws = lib.chebgcn_fused_layer_workspace(graph.handle, B, Fin, K, Fout)
        ws = _workspace(nws, x.device, 'fused_fwd') if nws else None
        nbytes = 4.0 * B * M * (Fin + Fout + (Fin * (K - 1) if stack is not None else 0))
        _lib.check(_launch('fused_layer_fwd', nbytes, 2.0 * B * M * Fin * K * Fout, lambda: lib.chebgcn_fused_layer_fwd(
            graph.handle, _p(x), _p(Wc), _p(b), bias_kind, _p(stack), _p(out), _p(mask), _p(ws), nws, B, Fin, K, Fout, int(relu),
            _stream())), 'fused_layer_fwd')
        ctx.save_for_backward(stack, Wc, None if relu else out, mask)
        ctx.fold, ctx.mean, ctx.fused = bool(relu), False, True
        ctx.graph, ctx.cfg = graph, (B, M, Fin, K, Fout, 1, POOL_MAX, int(relu), bias_kind)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        ctx.grad_bufs = (bufs.dW, bufs.dbias) if bufs is not None else (None, None)
        ctx.done = bufs.done if bufs is not None else None
        ctx.precision = 'f32'
        ctx.pool_maps = None
        ctx.Wt = None
        ctx.link_in = ctx.link_out = None
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.lib()
        stack, Wc, out, argmax = ctx.saved_tensors
        dW_buf, dbias_buf = ctx.grad_bufs
        B, M, Fin, K, Fout, pool, pool_kind, relu, bias_kind = ctx.cfg
        g = ctx.graph
        mean = ctx.mean
        if mean:
            # every filter's dy is gout / Fout, one plane [B][Mp] per window, zero in the padding: scaled into a buffer whose
            # padding was zeroed once (one kernel per step instead of autograd's zero-fill + copy of a sliced output)
            gm = _mean_grad_buffer(B, ctx.graph.Mp, gout.device)
            torch.mul(gout, 1.0 / Fout, out=gm[:, :M])
            gout = gm
        else:
            gout = gout.contiguous()
        dev = gout.device
        dbias = None
        bias_job = None
        if bias_kind != BIAS_NONE and ctx.needs_input_grad[2]:
            if dbias_buf is not None:
                _check_grad_buffer(dbias_buf, ctx.bias_shape, 'dbias')
                dbias = dbias_buf                 # overwritten, like dW: one use per step
            else:
                dbias = torch.zeros(ctx.bias_shape, dtype=torch.float32, device=dev)
        # gradient wrt the input by the forward recurrence on dy (see dx_by_forward): dy is then materialised (slab 0 of the stack
        # the recurrence fills), so the ReluGrad is not folded into the contraction gradients of this layer
        # On graphs in length order only: there the forward recurrence kernel is the faster of the two (0.49 against 0.43 of the
        # HBM roofline in the step).  In the caller's order the Clenshaw kernels are as fast or faster -- measured: the reference's
        # own shape at N = 1000 / 2000 (batch 128, K = 10) 1.93 / 3.32 ms this way against 1.89 / 3.16 ms; the level-0 layers of
        # the pooling network run their forward recurrence on two planes, the adjoint on four (common.h pick_ell)
        by_fwd = bool(dx_by_forward and ctx.needs_input_grad[0] and not ctx.fused and K > 1 and Fout <= Fin
                      and ctx.precision != 'bf16' and g.ordered
                      and (not mean or (ctx.fold and dbias is not None)))
        fold = ctx.fold and not by_fwd
        Mo = M // pool
        dy16 = bool(bf16_dy16 and not fold and ctx.precision == 'bf16' and pool == 1 and relu and argmax is not None
                    and lib.chebgcn_bf16_dy16_supported(B, M, Fin, K, Fout))
        gstack = None
        merge_bias = False
        if by_fwd and mean:
            # the last layer under the fused feature mean: every filter's gradient is the plane gout / Fout; one pass gates it
            # with the ReLU mask into slab 0 of the stack the recurrence fills and reduces the bias gradient
            gstack = torch.empty((K, B, Fout, g.Mp), dtype=torch.float32, device=dev)
            dy, mask = gstack[0], None
            bws, nbws = _brelu_bwd_ws(B, M, Fout, 1, bias_kind, dev)
            _lib.check(_launch('relu_grad_mean', B * M * (4.0 + Fout * 4.25), 0.0, lambda: lib.chebgcn_relu_grad_mean(
                _p(gout), _p(argmax), _p(dy), _p(dbias), bias_kind, B, M, Fout, _p(bws), nbws, _stream())), 'relu_grad_mean')
            mean = False                                  # from here on an ordinary layer with a materialised dy
        elif fold:
            # ReluGrad folded into the two contraction gradients (chebgcn_contract_bwd_*_relu read gout and the
            # mask); what is left of this pass is the bias reduction, which writes nothing but dbias
            dy, mask = gout, argmax
            # small launches (atlas-sized layers): the per-vertex bias gradient rides in the launch that adds the weight gradient's
            # partials (chebgcn_contract_bwd_w_relu_bias) -- one ~5 us launch less per layer
            merge_bias = bool(merge_bias_small and dbias is not None and bias_kind == BIAS_VERTEX and not mean
                              and ctx.needs_input_grad[1] and not PRECISIONS[ctx.precision]
                              and lib.chebgcn_contract_bwd_w_relu_bias_merged(B, M, Fin, K, Fout))
            if dbias is not None and not merge_bias:
                # feeds nothing in backward: enqueued BEHIND contract_bwd_x / recurrence_bwd (the chain the next layer waits
                # for) -- 3.88 against 3.93 ms per step at the bench shape; on the second stream it costs 4 %
                def bias_job():
                    bws, nbws = _brelu_bwd_ws(B, M, Fout, 1, bias_kind, dev)
                    if mean:
                        _lib.check(_launch('bias_grad', B * M * (4.0 + Fout * 0.25), 0.0, lambda: lib.chebgcn_bias_grad_relu_mean(
                            _p(gout), _p(mask), _p(dbias), bias_kind, B, M, Fout, _p(bws), nbws, _stream())), 'bias_grad_relu_mean')
                        return
                    _lib.check(_launch('bias_grad', B * Fout * M * (4.0 + 0.25), 0.0, lambda: lib.chebgcn_brelu_pool_bwd(
                        _p(gout), None, _p(mask), None, _p(dbias), bias_kind, B, M, Fout, 1, pool_kind, 1, _p(bws), nbws,
                        _stream())), 'brelu_pool_bwd')
        elif dy16:
            # one-pass bf16 gradients of a wide layer: the ReluGrad pass writes dy as bf16 -- what the matrix cores would round it
            # to anyway (bit-identical results), half the bytes of the largest operand of both gradients
            dy, mask = torch.empty((B, Fout, g.Mp), dtype=torch.bfloat16, device=dev), None
            bk = bias_kind if dbias is not None else BIAS_NONE
            bws, nbws = _brelu_bwd_ws(B, M, Fout, 1, bk, dev)
            _lib.check(_launch('relu_grad_bf16', B * Fout * M * (6.0 + 0.25), 0.0, lambda: lib.chebgcn_relu_grad_bf16(
                _p(gout), _p(argmax), _p(dy), _p(dbias), bk, B, M, Fout, _p(bws), nbws, _stream())), 'relu_grad_bf16')
        elif (by_fwd and ctx.link_out is not None and ctx.link_out.gstack is not None
              and ctx.link_out.gstack.data_ptr() == gout.data_ptr() and tuple(gout.shape) == (B, Fout, g.Mp)):
            # the layer above stored its input gradient gated by this layer's mask, straight into slab 0 of this stack
            # (GateLink): what is left of the ReluGrad pass is the bias reduction, a plain sum of gated values over the windows
            gstack, ctx.link_out.gstack = ctx.link_out.gstack, None
            dy, mask = gstack[0], None
            if dbias is not None:
                # at once, not behind the layer's other gradients like the bias reduction of the `fold` arm below: dy was written
                # by the kernel in front of this one (2.95 against 2.97-3.01 ms per step at the bench shape, same box)
                bws, nbws = _brelu_bwd_ws(B, M, Fout, 1, bias_kind, dev)
                _lib.check(_launch('bias_grad', B * Fout * M * 4.0, 0.0, lambda: lib.chebgcn_brelu_pool_bwd(
                    _p(dy), None, None, None, _p(dbias), bias_kind, B, M, Fout, 1, pool_kind, 0, _p(bws), nbws,
                    _stream())), 'brelu_pool_bwd')
        else:
            if ctx.link_out is not None:
                ctx.link_out.gstack = None
            if by_fwd:
                gstack = torch.empty((K, B, Fout, g.Mp), dtype=torch.float32, device=dev)
                dy, mask = gstack[0], None          # T_0 of the recurrence on dy: written in place
            else:
                dy, mask = torch.empty((B, Fout, g.Mp), dtype=torch.float32, device=dev), None
            bk = bias_kind if dbias is not None else BIAS_NONE
            if ctx.pool_maps is not None:
                # pooled between two vertex orders: the forward's selection bytes carry the ReLU of the maximum as well
                nbws = lib.chebgcn_pool_scatter_bwd_workspace(B, M, Fout, pool, bk)
                bws = torch.empty(nbws, dtype=torch.uint8, device=dev) if nbws else None
                _lib.check(_launch('pool_scatter_bwd', B * Fout * (4.0 * M + 5.0 * Mo), 0.0, lambda: lib.chebgcn_pool_scatter_bwd(
                    _p(gout), _p(argmax), _p(ctx.pool_maps[1]), _p(dy), _p(dbias), bk, B, M, Fout, pool, pool_kind, relu,
                    _p(bws), nbws, _stream())), 'pool_scatter_bwd')
            else:
                # with the ReLU mask of a pool == 1 layer `out` is not read (a byte per four vertices instead)
                nbytes = B * Fout * M * (8.0 + 0.25) if out is None else 4.0 * B * Fout * (2 * Mo + M)
                bws, nbws = _brelu_bwd_ws(B, M, Fout, pool, bk, dev)
                _lib.check(_launch('brelu_pool_bwd', nbytes, 0.0, lambda: lib.chebgcn_brelu_pool_bwd(
                    _p(gout), _p(out), _p(argmax), _p(dy), _p(dbias), bk, B, M, Fout,
                    pool, pool_kind, relu, _p(bws), nbws, _stream())), 'brelu_pool_bwd')
        dW = None
        if ctx.needs_input_grad[1]:
            passes = PRECISIONS[ctx.precision]
            if passes:
                nbytes = lib.chebgcn_contract_bwd_w_bf16_workspace(B, M, Fin, K, Fout)
            else:
                nbytes = lib.chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout)
            ws = _workspace(nbytes, dev)
            if dW_buf is not None:
                _check_grad_buffer(dW_buf, (Fin * K, Fout), 'dW')
                dW = dW_buf                       # written, not accumulated: one use per step
            else:
                dW = torch.empty((Fin * K, Fout), dtype=torch.float32, device=dev)

            def launch_bwd_w():
                if dy16:
                    call = lambda: lib.chebgcn_contract_bwd_w_bf16_dy16(_p(stack), _p(dy), _p(dW), _p(ws), ws.numel(), B, M, Fin,
                                                                        K, Fout, _stream())
                elif passes:
                    call = lambda: lib.chebgcn_contract_bwd_w_bf16(_p(stack), _p(dy), _p(dW), _p(ws), ws.numel(), B, M, Fin,
                                                                   K, Fout, passes, _stream())
                elif fold and mean:
                    call = lambda: lib.chebgcn_contract_bwd_w_relu_mean(_p(stack), _p(dy), _p(mask), _p(dW), _p(ws), ws.numel(),
                                                                        B, M, Fin, K, Fout, _stream())
                elif fold and merge_bias:
                    call = lambda: lib.chebgcn_contract_bwd_w_relu_bias(_p(stack), _p(dy), _p(mask), _p(dW), _p(dbias), _p(ws),
                                                                        ws.numel(), B, M, Fin, K, Fout, _stream())
                elif fold:
                    call = lambda: lib.chebgcn_contract_bwd_w_relu(_p(stack), _p(dy), _p(mask), _p(dW), _p(ws), ws.numel(), B,
                                                                   M, Fin, K, Fout, _stream())
                else:
                    call = lambda: lib.chebgcn_contract_bwd_w(_p(stack), _p(dy), _p(dW), _p(ws), ws.numel(), B, M, Fin, K,
                                                              Fout, _stream())
                what = 'contract_bwd_w' + ('_' + ctx.precision if passes else '')
                _lib.check(_launch(what, 4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout, call), what)

            want_side = (Fout > 32) if overlap_bwd_w == 'auto' else bool(overlap_bwd_w)
            side = _side_stream(dev) if (want_side and ctx.needs_input_grad[0] and (timers is None or not timers.active)) else None
            if side is not None:
                # dW does not feed dx: it runs beside contract_bwd_x / recurrence_bwd on a second stream
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    launch_bwd_w()
                    if bias_job is not None and ctx.fused and bias_side_small:
                        # atlas-sized layers: the main stream is the critical path of a chain of short launches and the second
                        # stream has slack -- the bias reduction goes there too (on the benchmark graph that costs 4 %: it stays
                        # behind contract_bwd_x / recurrence_bwd on the main stream)
                        bias_job()
                        bias_job = None
            else:
                launch_bwd_w()
        else:
            side = None
        dx = None
        if ctx.needs_input_grad[0] and ctx.fused:
            # G_j = dy W_j^T on the matrix cores feeding the adjoint recurrence on chip: no gradient stack in memory
            dx = torch.empty((B, Fin, g.Mp), dtype=torch.float32, device=dev)
            _lib.check(_launch('fused_layer_bwd_x', 4.0 * B * M * (Fin + Fout), 2.0 * B * M * Fin * K * Fout,
                               lambda: lib.chebgcn_fused_layer_bwd_x(g.handle, _p(dy), _p(mask), _p(Wc), _p(dx), B, Fin, K, Fout,
                                                                     _stream())), 'fused_layer_bwd_x')
        elif by_fwd:
            _lib.check(_launch('recurrence_fwd_t', 4.0 * B * M * Fout * K, 0.0, lambda: lib.chebgcn_recurrence_fwd_t(
                g.handle, _p(dy), _p(gstack), B, Fout, K, _stream())), 'recurrence_fwd_t')
            Wt = ctx.Wt                                                                        # W'[fo*K + k][fin] = W[fin*K + k][fo]
            if Wt is None or tuple(Wt.shape) != (Fout * K, Fin):
                Wt = torch.empty((Fout * K, Fin), dtype=torch.float32, device=dev)
                _lib.check(lib.chebgcn_reindex_weights(_p(Wc), _p(Wt), Fin, K, Fout, _stream()), 'reindex_weights')
            li, gate = ctx.link_in, None
            if (li is not None and li.mask is not None and ctx.precision == 'f32' and li.shape[1:] == (B, Fin, g.Mp)
                    and lib.chebgcn_contract_fwd_gated_supported(B, M, Fout, K, Fin)):
                # the layer below wants its dy as slab 0 of a gradient stack: this contraction stores it there, gated by that
                # layer's ReLU mask
                li.gstack = torch.empty(li.shape, dtype=torch.float32, device=dev)
                dx, gate = li.gstack[0], li.mask
            else:
                dx = torch.empty((B, Fin, g.Mp), dtype=torch.float32, device=dev)
            contract_fwd_into(gstack, Wt, None, BIAS_NONE, dx, None, B, M, Fout, K, Fin, 1, POOL_MAX, False, ctx.precision,
                              what='contract_bwd_x', gate=gate)
        elif ctx.needs_input_grad[0]:
            gstack = torch.empty((K, B, Fin, g.Mp), dtype=torch.float32, device=dev)
            passes = PRECISIONS[ctx.precision]
            if passes:
                nws = lib.chebgcn_contract_bwd_x_bf16_workspace(Fin, K, Fout)
                wsx = _workspace(nws, dev, 'bwd_x_bf16')                  # its own: bwd_w may be running beside it
                what = 'contract_bwd_x_' + ctx.precision
                if dy16:
                    call = lambda: lib.chebgcn_contract_bwd_x_bf16_dy16(_p(dy), _p(Wc), _p(gstack), B, M, Fin, K, Fout, _p(wsx), nws,
                                                                        _stream())
                else:
                    call = lambda: lib.chebgcn_contract_bwd_x_bf16(_p(dy), _p(Wc), _p(gstack), B, M, Fin, K, Fout, passes,
                                                                   _p(wsx), nws, _stream())
                _lib.check(_launch(what, 4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout, call), what)
            elif fold and mean:
                _lib.check(_launch('contract_bwd_x', 4.0 * B * M * (Fin * K + 1), 2.0 * B * M * Fin * K * Fout,
                                   lambda: lib.chebgcn_contract_bwd_x_relu_mean(_p(dy), _p(mask), _p(Wc), _p(gstack), B, M, Fin,
                                                                                K, Fout, _stream())), 'contract_bwd_x_relu_mean')
            elif fold:
                _lib.check(_launch('contract_bwd_x', 4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout,
                                   lambda: lib.chebgcn_contract_bwd_x_relu(_p(dy), _p(mask), _p(Wc), _p(gstack), B, M, Fin, K,
                                                                           Fout, _stream())), 'contract_bwd_x_relu')
            else:
                _lib.check(_launch('contract_bwd_x', 4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout,
                                   lambda: lib.chebgcn_contract_bwd_x(_p(dy), _p(Wc), _p(gstack), B, M, Fin, K, Fout,
                                                                      _stream())), 'contract_bwd_x')
            dx = torch.empty((B, Fin, g.Mp), dtype=torch.float32, device=dev)
            _lib.check(_launch('recurrence_bwd', 4.0 * B * M * Fin * (K + 1), 0.0, lambda: lib.chebgcn_recurrence_bwd(
                g.handle, _p(gstack), _p(dx), B, Fin, K, _stream())), 'recurrence_bwd')
        if bias_job is not None:
            bias_job()
        if side is not None:
            # joined before this layer's buffers (stack, dy, workspace) can be reused and before
            # anything consumes dW
            torch.cuda.current_stream(dev).wait_stream(side)
        if ctx.done is not None:
            ctx.done()                            # e.g. dist.DataParallel.layer_done: this layer's gradients are enqueued
        # gradients written into the caller's buffers are not handed to autograd a second time
        return (dx, None if dW_buf is not None else dW, None if dbias_buf is not None else dbias,
                None, None, None, None, None, None, None)


class Buffers:
    """Preallocated buffers for ``cheb_conv`` (plain object, not a tensor).  ``stack`` / ``out``:
    see ChebConv.  ``dW`` / ``dbias``: gradient buffers the backward pass WRITES (overwrites, it
    does not accumulate: one use of a variable per step) in place of returning the gradients to
    autograd -- the model hands over views of its flat gradient buffer and saves an add per
    variable and step.  ``done``: called at the end of the layer's backward, once its gradient
    kernels are enqueued (dist.DataParallel starts the layer's all-reduce from it).  ``mean``: the
    layer is followed by ``tf.reduce_mean(x, -1)`` (models_gcn.py:673) and returns that mean, storage
    [B, Mp], instead of its output (chebgcn_contract_fwd_mean; the gradients read one plane per window)."""
    __slots__ = ('stack', 'out', 'dW', 'dbias', 'precision', 'done', 'mean', 'grad_mode', 'pool_maps', 'Wt', 'link_in', 'link_out')

    def __init__(self, stack=None, out=None, dW=None, dbias=None, precision='f32', done=None, mean=False, pool_maps=None, Wt=None,
                 link_in=None, link_out=None):
        # ``GateLink`` objects shared with the layer below (link_in) / above (link_out) when this layer's input IS that layer's
        # output and nothing else reads it: the upper layer's input gradient is then stored gated by the lower layer's ReLU mask
        self.link_in, self.link_out = link_in, link_out
        # the layer's weights already re-indexed for the forward form of the input gradient (reindex_weights_batch: the model
        # does every layer in one launch in front of the backward pass); None: the backward pass re-indexes them itself
        self.Wt = Wt
        self.stack, self.out, self.dW, self.dbias, self.precision, self.done = stack, out, dW, dbias, precision, done
        # a pooled layer whose input and / or output vertices are not in the coarsening's tree order: (pmap, smap), int32 device
        # tensors of M entries each (``pool_maps``); the layer then pools through them (chebgcn_pool_gather_fwd / _scatter_bwd)
        self.pool_maps = pool_maps
        self.mean = mean          # the layer returns the mean over its filters, [B, Mp] (see conv_mean_supported)
        # the caller's grad mode: inside Function.forward grad mode is always off, and ``ctx.needs_input_grad`` is True for a
        # Parameter even under torch.no_grad() -- with this off nothing that only a backward pass would read is written
        # (the ReLU mask; in the fused atlas layer the whole K-slab stack)
        self.grad_mode = torch.is_grad_enabled()


class GateLink:
    """What two consecutive ``cheb_conv`` layers share so that the ReluGrad of the lower one runs in the epilogue of the upper
    one's input gradient (TF autodiff chains exactly these two ops: models_gcn.py:616 -> :625/:629 of the previous layer).

    The caller creates one per pair -- ``Buffers(link_out=l)`` for the lower layer, ``Buffers(link_in=l)`` for the upper -- and
    thereby states that the lower layer's output feeds the upper layer and NOTHING else.  The lower layer's forward fills
    ``mask`` / ``shape`` when its own backward will want its dy as slab 0 of a gradient stack; the upper layer's backward then
    allocates that stack, stores its gated input gradient there (``gstack``) and returns the slab to autograd; the lower layer's
    backward recognises the slab BY ADDRESS (anything else -- a copy, a sum with another gradient -- takes the ordinary pass, and
    gating a gated gradient again changes nothing) and only reduces its bias gradient."""
    __slots__ = ('mask', 'shape', 'gstack')

    def __init__(self):
        self.mask = self.shape = self.gstack = None


gate_links = os.environ.get('CHEBGCN_GATE_LINKS', '1') != '0'
merge_bias_small = os.environ.get('CHEBGCN_MERGE_BIAS_SMALL', '1') != '0'     # the bias gradient in the weight gradient's reduce launch (small launches)


def conv_mean_supported(B, M, Fin, K, Fout, pool, relu, precision='f32'):
    """Can ``cheb_conv(..., mean=True)`` serve this layer?  (pool 1, ReLU, fp32 contraction, a shape of the ring kernel.)"""
    precision = resolve_precision(precision, Fin, K, Fout)
    return bool(fold_relu_grad and pool == 1 and relu and precision == 'f32'
                and _lib.lib().chebgcn_contract_fwd_mean_supported(B, M, Fin, K, Fout))


def pool_maps(pool, src_order, dst_order, M, device):
    """Index maps of a pooled layer between two vertex orders (include/chebgcn.h, chebgcn_pool_gather_fwd): the reference pools
    the consecutive vertices ``pool*j .. pool*j + pool - 1`` of its (tree) order into vertex ``j`` (models_gcn.py:631-648).
    ``src_order[v']`` = reference vertex at position ``v'`` of the source level's internal order (None: identity), ``dst_order``
    likewise for the pooled level.  Returns (pmap, smap) int32 device tensors:
    ``pmap[j'*pool + i]`` = source position of member i of pooled position j'; ``smap[v']`` = ``j'*pool + i``."""
    M = int(M)
    Mo = M // pool
    src = np.arange(M, dtype=np.int64) if src_order is None else np.asarray(src_order, np.int64)
    dst = np.arange(Mo, dtype=np.int64) if dst_order is None else np.asarray(dst_order, np.int64)
    if src.shape != (M,) or dst.shape != (Mo,):
        raise ValueError('pool_maps: orders of %d / %d vertices expected' % (M, Mo))
    inv_src = np.empty(M, np.int64)
    inv_src[src] = np.arange(M)
    pmap = inv_src[(pool * dst[:, None] + np.arange(pool)[None, :]).reshape(-1)]
    smap = np.empty(M, np.int64)
    smap[pmap] = np.arange(M)
    dev = torch.device(device)
    return (torch.as_tensor(pmap.astype(np.int32)).to(dev), torch.as_tensor(smap.astype(np.int32)).to(dev))


def dx_by_forward_shape(graph, Fin, K, Fout, precision):
    """Does a layer of this shape form its input gradient by the forward recurrence on dy (``dx_by_forward``; the shape part of
    the rule in ``ChebConv.backward``)?  Such layers read the re-indexed weights W'[fo*K + k][fin] = W[fin*K + k][fo]."""
    return bool(dx_by_forward and K > 1 and Fout <= Fin and resolve_precision(precision, Fin, K, Fout) != 'bf16' and graph.ordered)


def reindex_weights_batch(Ws, shapes):
    """``[W'_l]`` for the layers ``Ws`` ([Fin*K, Fout] each, ``shapes`` = [(Fin, K, Fout)]) in ONE launch
    (chebgcn_reindex_weights_batch): W'[fo*K + k][fin] = W[fin*K + k][fo]."""
    n = len(Ws)
    if n == 0:
        return []
    if n > 16:
        return reindex_weights_batch(Ws[:16], shapes[:16]) + reindex_weights_batch(Ws[16:], shapes[16:])
    _require_cuda(*Ws)
    Ws = [w.detach() if w.is_contiguous() else w.detach().contiguous() for w in Ws]
    total = sum(fi * k * fo for fi, k, fo in shapes)
    flat = torch.empty(total, dtype=torch.float32, device=Ws[0].device)
    outs, at = [], 0
    for fi, k, fo in shapes:
        outs.append(flat[at:at + fi * k * fo].view(fo * k, fi))
        at += fi * k * fo
    arr_p = (C.c_void_p * n)
    arr_i = (C.c_int * n)
    _lib.check(_lib.lib().chebgcn_reindex_weights_batch(
        n, arr_p(*[w.data_ptr() for w in Ws]), arr_p(*[o.data_ptr() for o in outs]), arr_i(*[s[0] for s in shapes]),
        arr_i(*[s[1] for s in shapes]), arr_i(*[s[2] for s in shapes]), _stream()), 'reindex_weights_batch')
    return outs


def cheb_conv(x, W, bias, graph, K, pool=1, pool_kind=POOL_MAX, relu=False, bias_kind=BIAS_NONE, stack=None, out=None,
              dW=None, dbias=None, precision='f32', done=None, mean=False, pool_maps=None, Wt=None, link_in=None, link_out=None):
    """``precision``: arithmetic of the contraction and of its two gradients ('auto': resolve_precision; 'f32', 'bf16', 'bf16x3':
    chebgcn_contract_fwd_bf16 / _bwd_x_bf16 / _bwd_w_bf16 with 1 or 3 passes); storage, the recurrence, its adjoint
    and the bias / ReLU / pooling gradients stay fp32.  ``link_in`` / ``link_out``: ``GateLink``."""
    precision = resolve_precision(precision, x.shape[1], K, W.shape[1])
    bufs = Buffers(stack, out, dW, dbias, precision, done, mean, pool_maps if pool > 1 else None, Wt, link_in, link_out)
    return ChebConv.apply(x, W, bias, graph, K, pool, pool_kind, relu, bias_kind, bufs)


class BiasReluPool(torch.autograd.Function):
    """Standalone bias + ReLU + pooling (b1relu / b2relu / mpool1 / apool1 called on their
    own, models_gcn.py:619-648) on plane storage tensors."""

    @staticmethod
    def forward(ctx, x, bias, M, pool, pool_kind, relu, bias_kind):
        _require_cuda(x, bias)
        x = x if x.is_contiguous() else x.contiguous()
        B, F, Mp = x.shape
        out = plane_empty(B, F, M // pool, x.device)
        argmax = None
        if pool > 1 and (pool_kind == POOL_MAX or relu):
            argmax = torch.empty(out.shape, dtype=torch.uint8, device=x.device)
        b = bias.detach().contiguous() if bias is not None else None
        _lib.check(_lib.lib().chebgcn_brelu_pool_fwd(_p(x), _p(b), bias_kind, _p(out), _p(argmax), B, M, F, pool,
                                                     pool_kind, int(relu), _stream()), 'brelu_pool_fwd')
        ctx.save_for_backward(out, argmax)
        ctx.cfg = (B, M, F, pool, pool_kind, int(relu), bias_kind, Mp)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        out, argmax = ctx.saved_tensors
        B, M, F, pool, pool_kind, relu, bias_kind, Mp = ctx.cfg
        gout = gout.contiguous()
        dy = torch.empty((B, F, Mp), dtype=torch.float32, device=gout.device)
        dbias = None
        if bias_kind != BIAS_NONE and ctx.needs_input_grad[1]:
            dbias = torch.zeros(ctx.bias_shape, dtype=torch.float32, device=gout.device)
        bk = bias_kind if dbias is not None else BIAS_NONE
        bws, nbws = _brelu_bwd_ws(B, M, F, pool, bk, gout.device)
        _lib.check(_lib.lib().chebgcn_brelu_pool_bwd(_p(gout), _p(out), _p(argmax), _p(dy), _p(dbias), bk, B, M, F, pool,
                                                     pool_kind, relu, _p(bws), nbws, _stream()), 'brelu_pool_bwd')
        return dy, dbias, None, None, None, None, None


class FeatureMean(torch.autograd.Function):
    """tf.reduce_mean(x, -1) (models_gcn.py:673): storage [B, F, Mp] -> dense [B, M]."""

    @staticmethod
    def forward(ctx, x, M):
        _require_cuda(x)
        x = x if x.is_contiguous() else x.contiguous()
        B, F, Mp = x.shape
        y = torch.empty((B, M), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().chebgcn_feature_mean_fwd(_p(x), _p(y), B, M, F, _stream()), 'feature_mean_fwd')
        ctx.cfg = (B, M, F, Mp)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, M, F, Mp = ctx.cfg
        gy = gy.contiguous()
        dx = torch.empty((B, F, Mp), dtype=torch.float32, device=gy.device)
        _lib.check(_lib.lib().chebgcn_feature_mean_bwd(_p(gy), _p(dx), B, M, F, _stream()), 'feature_mean_bwd')
        return dx, None


FC_BWD_MAX_INNER = int(os.environ.get('CHEBGCN_FC_BWD_MAX_INNER', 4096))      # beyond: the gradients stay on the vendor GEMMs (tools/probes/fc_small_probe.py)
FC_FWD_MAX_INNER = int(os.environ.get('CHEBGCN_FC_FWD_MAX_INNER', 1 << 20))     # A/B knob for bench runs


def fc_forward(x, W, b, relu):
    """``act(x @ W + b)`` of the head's FC layers (models_gcn.py:650-656) by the library's small-product kernel, or None
    where the product is outside its range (large or odd inner size: the caller uses the vendor GEMM).  x may be a
    [B, M] view of a [B, Mp] buffer."""
    if not (x.is_cuda and x.dtype == torch.float32 and W.dtype == torch.float32 and x.dim() == 2 and W.is_contiguous()
            and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0):
        return None
    B, I = x.shape
    O = W.shape[1]
    L = _lib.lib()
    if I > FC_FWD_MAX_INNER or not L.chebgcn_fc_fwd_supported(B, I, O):
        return None
    y = torch.empty((B, O), dtype=torch.float32, device=x.device)
    nws = L.chebgcn_fc_fwd_workspace(B, I, O)
    ws = _workspace(nws, x.device, 'fc_fwd') if nws else None
    _lib.check(L.chebgcn_fc_fwd(_p(x), x.stride(0), _p(W), _p(b), _p(y), _p(ws), nws, B, I, O, 1 if relu else 0, _stream()),
               'fc_fwd')
    return y


def fc_backward(x, W, g, y, dW, db, need_dx):
    """The three gradients of ``act(x @ W + b)`` (ReluGrad on ``y`` where given) by the library kernels: dW and db are
    WRITTEN into the given buffers.  Returns ``(dx,)`` (``(None,)`` unless ``need_dx``), or None where fc_forward would
    decline and nothing was launched."""
    B, I = x.shape
    O = W.shape[1]
    L = _lib.lib()
    if not (x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1 and W.is_contiguous() and g.is_contiguous()
            and dW.is_contiguous() and db.is_contiguous() and (y is None or y.is_contiguous())
            and I <= FC_BWD_MAX_INNER and L.chebgcn_fc_fwd_supported(B, I, O)):
        return None
    dx = torch.empty((B, I), dtype=torch.float32, device=x.device) if need_dx else None
    _lib.check(L.chebgcn_fc_bwd(_p(x), x.stride(0), _p(W), _p(g), _p(y), _p(dW), _p(db), _p(dx), I, B, I, O, _stream()),
               'fc_bwd')
    return (dx,)


def adam_step_sq_all(p, g, m, v, n_reg, lr_t, sq_partials, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, l2=0.0):
    """``adam_step_sq`` over ALL variables in one launch: the first ``n_reg`` elements are regularised (L2 term in the gradient,
    counted in the partial sums of squares), the rest -- the biases -- take plain Adam.  Returns the number of partials."""
    _require_cuda(p, g, m, v, sq_partials)
    n = p.numel()
    if not (g.numel() == m.numel() == v.numel() == n) or n == 0 or not (0 <= n_reg <= n):
        raise ValueError('adam_step_sq_all: size mismatch')
    lib = _lib.lib()
    nparts = lib.chebgcn_adam_partials(n)
    if sq_partials.dtype != torch.float32 or sq_partials.numel() < nparts:
        raise ValueError('adam_step_sq_all: sq_partials needs %d float32' % nparts)
    dev_lr = isinstance(lr_t, torch.Tensor)
    if dev_lr and (lr_t.dtype != torch.float32 or lr_t.numel() != 1 or not lr_t.is_cuda):
        raise ValueError('adam_step_sq_all: a device lr_t must be one float32')
    _lib.check(lib.chebgcn_adam_step_sq_all(_p(p), _p(g), _p(m), _p(v), n, int(n_reg), 0.0 if dev_lr else float(lr_t),
                                            _p(lr_t) if dev_lr else None, float(beta1), float(beta2), float(eps), float(grad_scale),
                                            float(l2), _p(sq_partials), _stream()), 'adam_step_sq_all')
    return nparts


def adam_step_sq(p, g, m, v, lr_t, sq_partials, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, l2=0.0):
    """``adam_step`` that also leaves the per-workgroup partial sums of squares of the PRE-update ``p`` in ``sq_partials``
    (float32, at least ``adam_partials(n)`` elements): the L2 term of the loss without a pass of its own.  Returns the
    number of partials written."""
    _require_cuda(p, g, m, v, sq_partials)
    n = p.numel()
    if not (g.numel() == m.numel() == v.numel() == n) or n == 0:
        raise ValueError('adam_step_sq: size mismatch')
    lib = _lib.lib()
    nparts = lib.chebgcn_adam_partials(n)
    if sq_partials.dtype != torch.float32 or sq_partials.numel() < nparts:
        raise ValueError('adam_step_sq: sq_partials needs %d float32' % nparts)
    dev_lr = isinstance(lr_t, torch.Tensor)
    if dev_lr and (lr_t.dtype != torch.float32 or lr_t.numel() != 1 or not lr_t.is_cuda):
        raise ValueError('adam_step_sq: a device lr_t must be one float32')
    _lib.check(lib.chebgcn_adam_step_sq(_p(p), _p(g), _p(m), _p(v), n, 0.0 if dev_lr else float(lr_t), _p(lr_t) if dev_lr else None,
                                        float(beta1), float(beta2), float(eps), float(grad_scale), float(l2), _p(sq_partials),
                                        _stream()), 'adam_step_sq')
    return nparts


def loss_bookkeeping(cross_entropy, sq_partials, nparts, half_reg, ema, corr, decay=0.9):
    """loss = cross_entropy + half_reg * sum(sq_partials[:nparts]); ema <- ema + (1 - decay)(loss - ema) in place;
    returns loss_average = ema * corr (``corr``: float, or one-element device tensor read when the kernel runs)."""
    _require_cuda(cross_entropy, ema)
    out = torch.empty((), dtype=torch.float32, device=ema.device)
    dev_c = isinstance(corr, torch.Tensor)
    _lib.check(_lib.lib().chebgcn_loss_bookkeeping(_p(cross_entropy), _p(sq_partials) if nparts else None, int(nparts), float(half_reg),
                                                   _p(ema), float(decay), 0.0 if dev_c else float(corr), _p(corr) if dev_c else None,
                                                   None, _p(out), _stream()), 'loss_bookkeeping')
    return out


def adam_step(p, g, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, l2=0.0):
    """In-place TF-form Adam on flat fp32 buffers (models_gcn.py:296).  ``lr_t``: a Python float, or a one-element
    fp32 DEVICE tensor read when the kernel runs (chebgcn_adam_step_dev: the form a captured step graph replays)."""
    _require_cuda(p, g, m, v)
    n = p.numel()
    if not (g.numel() == m.numel() == v.numel() == n):
        raise ValueError('adam_step: size mismatch')
    if isinstance(lr_t, torch.Tensor):
        _require_cuda(lr_t)
        if lr_t.dtype != torch.float32 or lr_t.numel() != 1:
            raise ValueError('adam_step: a device lr_t must be one float32')
        _lib.check(_lib.lib().chebgcn_adam_step_dev(_p(p), _p(g), _p(m), _p(v), n, _p(lr_t), float(beta1), float(beta2),
                                                    float(eps), float(grad_scale), float(l2), _stream()), 'adam_step_dev')
        return
    _lib.check(_lib.lib().chebgcn_adam_step(_p(p), _p(g), _p(m), _p(v), n, float(lr_t), float(beta1), float(beta2),
                                            float(eps), float(grad_scale), float(l2), _stream()), 'adam_step')
