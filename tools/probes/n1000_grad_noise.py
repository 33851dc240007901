#!/usr/bin/env python3
"""Gradient error against float64 of the N = 1000 reference-shape network (6 x [K = 10, F = 32], batch 128) with the SAME variables
in the caller's vertex order and in the length order (CHEBGCN_ORD_SMALL=0/1 in the environment), several seeds."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_fmri_decoding_amd import graph, models_gcn, ops
from oracle import layers_ref as R
dev = torch.device('cuda:0')
n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
Ls, perm, _ = graph.synthetic_graph(n_nodes, k=8, levels=1)
L = Ls[0]
M, B, C = L.shape[0], 128, 15
F, K, p, Mfc = [32] * 6, [10] * 6, [1] * 6, [512, 256, 22]
reg = 5e-4
for seed in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    net = models_gcn.cgcnn({'device': dev}, [L] * 6, F, K, p, Mfc, filter='chebyshev5', brelu='b2relu', pool='mpool1', initial='he',
                           channel=C, regularization=reg, dropout=1, batch_size=B, verbose=False)
    onet = R.Net([L] * 6, F, K, p, Mfc, channel=C, brelu='b2relu', regularization=reg)
    rs = np.random.RandomState(100 + seed)
    params = {}
    for k, s in onet.param_shapes().items():
        params[k] = ((0.05 * rs.randn(*s)) if k.endswith('bias') else rs.randn(*s) * np.sqrt(2.0 / s[0])).astype(np.float32)
        net.set_variable(k, params[k])
    x = np.zeros((B, M, C), np.float32)
    keep = np.asarray(perm) < n_nodes
    x[:, keep, :] = rs.randn(B, int(keep.sum()), C).astype(np.float32)
    labels = rs.randint(0, 21, B)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    onet64 = R.Net([L.astype(np.float64)] * 6, F, K, p, Mfc, channel=C, brelu='b2relu', regularization=reg)
    logits64, cache64 = onet64.forward(p64, x.astype(np.float64))
    loss64, dlogits64 = onet64.loss(p64, logits64, labels)
    grads64 = onet64.backward(p64, cache64, dlogits64)
    xs = ops.plane_storage(torch.as_tensor(x).to(dev)).contiguous()
    net.train_step(xs, torch.as_tensor(labels).to(dev))
    out = []
    for k in params:
        if not k.startswith('conv'):
            continue
        l2 = reg * p64[k] if onet.regularized(k) else 0
        ref = grads64[k] - l2
        e = np.abs(net.gradient(k).cpu().numpy().astype(np.float64) - ref) / np.abs(ref).max()
        out.append('%s %.1e/%.1e' % (k.replace('conv', 'c').replace('/weights', 'W').replace('/bias', 'b'), np.quantile(e, 0.99 if k.endswith('bias') else 0.999), e.max()))
    print('seed %d ordered=%s  ' % (seed, net.graphs[0].ordered) + '  '.join(out), flush=True)
    del net
