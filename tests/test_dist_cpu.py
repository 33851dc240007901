"""Data-parallel helper (gcn_fmri_decoding_amd.dist) on CPU: world_size 2, gloo backend.

The HIP kernels cannot run here, so the helper is driven with a stand-in that has the same
flat-buffer layout as cgcnn ([head | conv weights | conv biases], per-variable views with
.grad pointing into one flat gradient buffer) and a tiny torch network.  Checked: parameters
are broadcast from rank 0, the head bucket is reduced from the post-accumulate hook, and the
averaged gradients equal the single-process gradients of the concatenated batch.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _Spec:
    def __init__(self, name, shape, group):
        self.name, self.shape, self.group = name, shape, group


class StandIn:
    """Same attributes DataParallel touches on a cgcnn model."""

    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        # flat order like cgcnn.build_graph: [head | conv weights | conv biases], conv variables in layer order
        self._spec_list = [_Spec('fc1/weights', (6, 4), 'head'), _Spec('fc1/bias', (4,), 'head'),
                           _Spec('conv1/weights', (3, 6), 'convw'), _Spec('conv2/weights', (6, 6), 'convw'),
                           _Spec('conv3/weights', (6, 6), 'convw'),
                           _Spec('conv1/bias', (6,), 'convb'), _Spec('conv2/bias', (6,), 'convb'),
                           _Spec('conv3/bias', (6,), 'convb')]
        sizes = [int(np.prod(s.shape)) for s in self._spec_list]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        n = int(offs[-1])
        self._flat = torch.randn(n, generator=g)
        self._grad = torch.zeros(n)
        self._adam_m = torch.zeros(n)
        self._adam_v = torch.zeros(n)
        self._params, self._slices = {}, {}
        for s, a, b in zip(self._spec_list, offs[:-1], offs[1:]):
            p = torch.nn.Parameter(self._flat[a:b].view(s.shape))
            p.grad = self._grad[a:b].view(s.shape)
            self._params[s.name] = p
            self._slices[s.name] = (int(a), int(b))
        self._n_head = sizes[0] + sizes[1]
        self._n_total = n
        self._dp = None

    def loss(self, x, y):
        p = self._params
        h = torch.relu(x @ p['conv1/weights'] + p['conv1/bias'])
        h = torch.relu(h @ p['conv2/weights'] + p['conv2/bias'])
        h = torch.relu(h @ p['conv3/weights'] + p['conv3/bias'])
        out = h @ p['fc1/weights'] + p['fc1/bias']
        return torch.nn.functional.cross_entropy(out, y)


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from gcn_fmri_decoding_amd import dist as gdist
        model = StandIn(seed=100 + rank)            # different initial values per rank
        dp = gdist.DataParallel(model)
        # conv buckets: layers [3] and [1, 2] (two runs of consecutive layers, last first), one
        # contiguous weight range and one bias range each
        sl = model._slices
        assert dp._buckets == [(3, [sl['conv3/weights'], sl['conv3/bias']]),
                               (1, [(sl['conv1/weights'][0], sl['conv2/weights'][1]), (sl['conv1/bias'][0], sl['conv2/bias'][1])])]
        # a model that re-draws its variables after wrapping (cgcnn.fit does: the reference re-runs
        # op_init there) must be re-synchronised: fit() calls broadcast_parameters() again
        with torch.no_grad():
            model._flat.copy_(torch.randn(model._flat.shape, generator=torch.Generator().manual_seed(200 + rank)))
        dp.broadcast_parameters()
        flat0 = model._flat.clone()
        g = torch.Generator().manual_seed(7)
        X = torch.randn(8, 3, generator=g)
        Y = torch.randint(0, 4, (8,), generator=g)
        xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
        model._grad.zero_()
        dp.begin_step()
        model.loss(xs, ys).backward()
        fired = not dp._pending and len(dp._work) == 1     # head bucket went out from the hook
        dp.layer_done(3)                                    # what ops.ChebConv.backward reports, layer by layer
        fired = fired and len(dp._work) == 3 and dp._sent == {0}
        dp.layer_done(2)                                    # not the first layer of its run: nothing goes out yet
        fired = fired and len(dp._work) == 3
        scale = dp.finish_step()                            # the run [1, 2] is sent here (layer 1 never reported)
        fired = fired and dp._sent == {0, 1}
        ret[rank] = (flat0.numpy(), (model._grad * scale).numpy().copy(), fired)
    finally:
        dist.destroy_process_group()


def test_data_parallel_gloo_world2():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    (f0, g0, fired0), (f1, g1, fired1) = ret[0], ret[1]
    assert np.array_equal(f0, f1)                      # broadcast from rank 0
    assert fired0 and fired1
    assert np.array_equal(g0, g1)
    # single-process reference on the whole batch, starting from rank 0's parameters
    ref = StandIn(seed=100)
    with torch.no_grad():
        ref._flat.copy_(torch.randn(ref._flat.shape, generator=torch.Generator().manual_seed(200)))
    assert np.array_equal(ref._flat.numpy(), f0)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 3, generator=g)
    Y = torch.randint(0, 4, (8,), generator=g)
    ref.loss(X, Y).backward()
    np.testing.assert_allclose(g0, ref._grad.numpy(), rtol=1e-5, atol=1e-7)
