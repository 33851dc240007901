"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol that
include/chebgcn.h declares, argument validation works without a GPU, the product never
imports the oracle, and the host-side model logic (variable names / shapes / flat layout)
mirrors the reference."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT, csr_from, load_golden


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'chebgcn.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(chebgcn_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from gcn_fmri_decoding_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), 'libchebgcn.so does not export %s' % n
    # the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names
    # ... and the library exports nothing else under the chebgcn_ prefix (no undeclared hooks)
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-1].startswith('chebgcn_')})
    assert exported == names, sorted(set(exported) ^ set(names))
    assert _lib.lib().chebgcn_version() == 1
    for M in (1, 31, 32, 33, 10466):
        assert _lib.lib().chebgcn_plane_stride(M) == _lib.plane_stride(M) >= M
    assert _lib.plane_stride(10466) == 10496


def test_argument_validation_without_gpu():
    from gcn_fmri_decoding_amd import _lib
    lib = _lib.lib()
    out = ctypes.c_void_p()
    rc = lib.chebgcn_graph_create(0, 0, None, None, None, ctypes.byref(out))
    assert rc == -1 and b'graph_create' in lib.chebgcn_last_error()
    rp = np.array([0, 1, 3], np.int32)      # rowptr[M] != nnz
    ci = np.array([0, 1], np.int32)
    va = np.ones(2, np.float32)
    rc = lib.chebgcn_graph_create(2, 2, rp.ctypes.data_as(ctypes.c_void_p), ci.ctypes.data_as(ctypes.c_void_p),
                                  va.ctypes.data_as(ctypes.c_void_p), ctypes.byref(out))
    assert rc == -1
    assert lib.chebgcn_recurrence_fwd(None, None, None, 1, 1, 1, None) == -1
    assert lib.chebgcn_contract_fwd(None, None, None, 0, None, None, 1, 1, 1, 1, 1, 1, 0, 0, None) == -1
    assert lib.chebgcn_contract_bwd_w_workspace(0, 1, 1, 1, 1) == 0
    # bf16 contraction: workspace arithmetic and argument checks need no device
    assert lib.chebgcn_contract_fwd_bf16_workspace(0, 5, 256) == 0
    assert lib.chebgcn_contract_fwd_bf16_workspace(60, 5, 256) == 2 * 19 * 256 * 16 * 2      # hi + lo, 19 k-steps
    assert lib.chebgcn_contract_fwd_bf16(None, None, None, 0, None, None, 1, 1, 1, 1, 1, 1, 0, 0, 1, None, 0, None) == -1
    assert b'contract_fwd_bf16' in lib.chebgcn_last_error()
    # bf16 gradients of the contraction
    assert lib.chebgcn_contract_bwd_x_bf16_workspace(60, 5, 0) == 0
    assert lib.chebgcn_contract_bwd_x_bf16_workspace(60, 5, 256) == 2 * 16 * 320 * 16 * 2     # W^T: 16 k-steps x 300 -> 320 rows (five waves)
    assert lib.chebgcn_contract_bwd_x_bf16_workspace(32, 5, 64) == 2 * 4 * 256 * 16 * 2        # 160 rows: one group of 256
    assert lib.chebgcn_contract_bwd_x_bf16(None, None, None, 1, 1, 1, 1, 1, 1, None, 0, None) == -1
    assert b'contract_bwd_x_bf16' in lib.chebgcn_last_error()
    assert lib.chebgcn_contract_bwd_w_bf16_workspace(0, 1, 1, 1, 1) == 0
    assert lib.chebgcn_contract_bwd_w_bf16(None, None, None, None, 0, 1, 1, 1, 1, 1, 1, None) == -1
    assert b'contract_bwd_w_bf16' in lib.chebgcn_last_error()
    # the head's FC layer: range, split arithmetic and argument checks
    assert lib.chebgcn_fc_fwd_supported(128, 512, 256) == 1 and lib.chebgcn_fc_fwd_supported(64, 10466, 512) == 1
    assert lib.chebgcn_fc_fwd_supported(2 ** 11, 8, 2 ** 10) == 0 and lib.chebgcn_fc_fwd_supported(0, 8, 8) == 0
    assert lib.chebgcn_fc_fwd_workspace(128, 512, 256) == 0                      # short reduction: one launch, no partials
    assert lib.chebgcn_fc_fwd_workspace(64, 10466, 512) == 16 * 64 * 512 * 4     # 32 tiles -> 16 splits of the 10466 features
    assert lib.chebgcn_fc_fwd(None, 8, None, None, None, None, 0, 1, 8, 1, 0, None) == -1
    assert b'fc_fwd' in lib.chebgcn_last_error()
    assert lib.chebgcn_fc_bwd(None, 8, None, None, None, None, None, None, 8, 1, 8, 1, None) == -1
    assert b'fc_bwd' in lib.chebgcn_last_error()
    with pytest.raises(_lib.ChebgcnError):
        _lib.check(-1, 'x')


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'gcn_fmri_decoding_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f
                assert 'oracle/' not in text and 'oracle.' not in text.replace('# oracle.', ''), f


def test_cpu_tensor_is_rejected():
    import torch
    from gcn_fmri_decoding_amd import _lib, ops
    with pytest.raises(_lib.ChebgcnError):
        ops.plane_storage(torch.zeros(1, 4, 2))
    from gcn_fmri_decoding_amd import models_gcn
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            models_gcn.cgcnn(None, [sp.identity(8, format='csr')], [2], [2], [1], [3], verbose=False)


@pytest.mark.parametrize('name', ['inference_pool_n212', 'inference_flat_n212', 'inference_config1_n512', 'inference_pool6_n512'])
def test_variable_layout_matches_reference(name):
    """Shape-only build (device='meta'): variable names and TF shapes equal the reference's
    (golden 'param:*' entries come from the reference's own _inference run)."""
    from gcn_fmri_decoding_amd import models_gcn
    z = load_golden(name)
    Ls = [csr_from(z, 'L%d' % i) for i in range(int(z['nlevels']))]
    net = models_gcn.cgcnn('meta', Ls, z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist(),
                           channel=int(z['channel']), brelu=str(z['brelu']), batch_size=4, verbose=False,
                           regularization=5e-4)
    want = {k[len('param:'):]: z[k].shape for k in z.files if k.startswith('param:')}
    assert {k: tuple(net.variable(k).shape) for k in net.variables()} == want
    # regularised: conv weights and every fc weight and bias; not the conv biases
    assert set(net.regularizers) == {k for k in want if not (k.startswith('conv') and k.endswith('bias'))}
    # flat layout: [head | conv weights | conv biases], regularised part first
    groups = [s.group for s in net._spec_list]
    assert groups == sorted(groups, key=['head', 'convw', 'convb'].index)
    assert net._n_reg == sum(int(np.prod(s.shape)) for s in net._spec_list if s.regularized)
    assert net._n_total == net._flat.numel()
    # reported learning rate: staircase decay, optimizer itself fixed (models_gcn.py:283-296)
    net.decay_steps, net.decay_rate, net.learning_rate = 10, 0.9, 0.001
    net.global_step = 25
    assert abs(net.training(None, 0.001, 10, 0.9, 0.9) - 0.001 * 0.9 ** 2) < 1e-12


def test_constructor_checks_like_reference():
    from gcn_fmri_decoding_amd import models_gcn
    L = [sp.identity(16, format='csr', dtype=np.float32), sp.identity(8, format='csr', dtype=np.float32)]
    with pytest.raises(AssertionError):     # pooling size not a power of two
        models_gcn.cgcnn('meta', L, [2], [2], [3], [3], verbose=False)
    with pytest.raises(AssertionError):     # not enough coarsening levels for p=8
        models_gcn.cgcnn('meta', L, [2], [2], [8], [3], verbose=False)
    net = models_gcn.cgcnn('meta', L, [2, 2], [2, 3], [2, 1], [3], verbose=False, batch_size=2)
    assert [l.shape[0] for l in net.L] == [16, 8]


def test_best_checkpoint_policy_matches_checkmat(tmp_path):
    """checkmat.BestCheckpointSaver (checkmat.py:8-118): keep the 3 best by value, JSON index
    'best_checkpoints' keyed 'best.ckpt-<step>'; get_best_checkpoint (:121-138) returns the top one."""
    import json
    from gcn_fmri_decoding_amd import models_gcn as M

    class Stub(M.base_model):
        def __init__(self, root):
            self.dir_name, self._root = 'run', str(root)

        def _get_path(self, folder):
            return os.path.join(self._root, folder, self.dir_name)

        def state_dict(self):
            return {'k': 1}

    st, best = Stub(tmp_path), []
    for step, acc in [(10, 0.3), (20, 0.5), (30, 0.4), (40, 0.35), (50, 0.6), (60, 0.1), (70, 0.5)]:
        st._save_best(acc, step, best)
    path = os.path.join(str(tmp_path), 'checkpoints', 'run', 'model')
    index = json.load(open(os.path.join(path, 'best_checkpoints')))
    assert index == {'best.ckpt-50': 0.6, 'best.ckpt-20': 0.5, 'best.ckpt-70': 0.5}
    assert sorted(f for f in os.listdir(path) if f.endswith('.pt')) == ['best.ckpt-20.pt', 'best.ckpt-50.pt', 'best.ckpt-70.pt']
    assert M.get_best_checkpoint(path) == os.path.join(path, 'best.ckpt-50')
    # tf.train.Saver's state file as checkmat leaves it (checkmat.py:70-84): last save first, then the
    # survivors best-first followed by the new checkpoint; models_gcn.py:968-969 reads line 1
    assert open(os.path.join(path, 'checkpoint')).read().splitlines() == [
        'model_checkpoint_path: "best.ckpt-70"', 'all_model_checkpoint_paths: "best.ckpt-50"',
        'all_model_checkpoint_paths: "best.ckpt-20"', 'all_model_checkpoint_paths: "best.ckpt-70"']
    assert best == ['best.ckpt-50', 'best.ckpt-20', 'best.ckpt-70']


def test_resolve_precision_rule():
    """``cgcnn.contraction = 'auto'`` (ops.resolve_precision): exact fp32 products up to 32 filters -- every layer of BASELINE
    configs[1] and of the reference's training.py (models_gcn.py:611-617 computes in fp32) -- split bf16 where the fp32 matrix
    cores would bound the layer; explicit choices pass through."""
    from gcn_fmri_decoding_amd import ops
    auto = lambda fin, k, fout: ops.resolve_precision('auto', fin, k, fout)
    assert [auto(15, 5, 32), auto(32, 5, 32), auto(32, 25, 32), auto(15, 10, 32)] == ['f32'] * 4          # configs[1], training.py
    assert auto(64, 25, 64) == 'bf16x3' and auto(60, 5, 256) == 'bf16x3'                                  # BASELINE configs[3], [4]
    assert [auto(32, 10, 64), auto(64, 10, 64), auto(64, 5, 128), auto(128, 5, 128)] == ['bf16x3'] * 4    # the pooling ChebNet
    assert auto(2, 2, 64) == 'f32'                                  # 64 filters but 4 reduction rows: 1.9 flop/B, HBM-bound in fp32
    for p in ('f32', 'bf16', 'bf16x3'):
        assert ops.resolve_precision(p, 64, 25, 64) == p
    assert ops.FP32_MFMA_BALANCE == pytest.approx(157.3e12 / 8e12, rel=0.01)


def test_internal_planes_wrapper_keeps_the_order_explicit():
    """A batch in a model's internal vertex order is a wrapper object (models_gcn.InternalPlanes), not a tensor with a hidden
    attribute: window slices, clone and detach of the WRAPPER stay wrapped; nothing a tensor operation returns is wrapped;
    indexing another axis and refilling from a plain tensor are refused."""
    import torch
    from gcn_fmri_decoding_amd.models_gcn import InternalPlanes
    owner = object()
    x = torch.arange(2 * 3 * 8, dtype=torch.float32).reshape(2, 3, 8)
    w = InternalPlanes(x, owner)
    assert w.shape == x.shape and len(w) == 2 and w.planes is x and w.owner is owner
    for v in (w[:], w[0:1], w.clone(), w.detach(), w[torch.tensor([1, 0])]):
        assert isinstance(v, InternalPlanes) and v.owner is owner and v.planes.dim() == 3
    assert w[1].shape == (1, 3, 8) and torch.equal(w[1].planes[0], x[1])
    assert not isinstance(w.planes[:1], InternalPlanes) and not hasattr(w.planes.clone(), '_chebgcn_internal')
    with pytest.raises(IndexError):
        w[:, 0]
    with pytest.raises(TypeError):
        w.copy_(x)
    with pytest.raises(TypeError):
        w.copy_(InternalPlanes(x.clone(), object()))
    y = InternalPlanes(torch.zeros_like(x), owner)
    assert torch.equal(y.copy_(w).planes, x)
    assert InternalPlanes(w, owner).planes is x                      # wrapping a wrapper does not nest


def test_bank_order_is_a_length_sorted_permutation():
    """``graph.bank_order`` (chebgcn_bank_order, host only): a permutation that keeps the rows sorted by descending length, lowers
    the gather's fullest-bank count, is deterministic, and is the plain length order where no ordered kernel serves the graph.
    ``ops.pool_maps``: the two maps of a pooled layer between vertex orders are inverse to each other and list the members of a
    cluster in the reference's order (models_gcn.py:631-648, coarsening.py:168-215)."""
    import scipy.sparse as sp
    from gcn_fmri_decoding_amd import graph
    rs = np.random.RandomState(3)
    n = 3000
    rows = np.repeat(np.arange(n), 6)
    cols = rs.randint(0, n, rows.size)
    keep = rows != cols
    W = sp.coo_matrix((np.exp(-rs.rand(int(keep.sum()))).astype(np.float32), (rows[keep], cols[keep])), shape=(n, n)).tocsr()
    W = W.maximum(W.T)
    L = graph.laplacian(W.astype(np.float32), normalized=True)
    st, st2 = [], []
    order = graph.bank_order(L, sweeps=4, stats=st)
    assert sorted(order.tolist()) == list(range(n))
    lens = np.diff(graph.rescaled_laplacian_csr(graph.permute(L, order))[0])
    assert (np.diff(lens) <= 0).all()                              # still sorted by descending row length
    assert st[1] < st[0] and st[2] > 0, st
    assert np.array_equal(order, graph.bank_order(L, sweeps=4, stats=st2)) and st == st2
    assert np.array_equal(graph.bank_order(L, sweeps=0), graph.length_order(L))
    small = graph.laplacian(W[:500][:, :500].astype(np.float32), normalized=True)           # no ordered kernel at this size
    assert np.array_equal(graph.bank_order(small), graph.length_order(small))


def test_pool_maps_definition():
    torch = pytest.importorskip('torch')
    from gcn_fmri_decoding_amd import ops
    rs = np.random.RandomState(0)
    M, p = 48, 4
    src, dst = rs.permutation(M), rs.permutation(M // p)
    pm, sm = ops.pool_maps(p, src, dst, M, 'cpu')
    pm, sm = pm.numpy(), sm.numpy()
    # pooled position j' is reference vertex dst[j']: its members are the reference vertices p*dst[j'] + i, found at inv_src[...]
    for j in range(M // p):
        for i in range(p):
            assert src[pm[j * p + i]] == p * dst[j] + i
    assert np.array_equal(sm[pm], np.arange(M))
    ident = ops.pool_maps(p, None, None, M, 'cpu')
    assert np.array_equal(ident[0].numpy(), np.arange(M)) and np.array_equal(ident[1].numpy(), np.arange(M))
