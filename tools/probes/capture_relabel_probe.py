#!/usr/bin/env python3
"""Debug probe: capture the training step of a pooling fixture with its levels forced into the length order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from conftest import load_golden, csr_from
from gcn_fmri_decoding_amd import models_gcn, ops, _lib
name, order, variant = sys.argv[1], sys.argv[2], sys.argv[3]
os.environ['CHEBGCN_VERTEX_ORDER'] = order
dev = torch.device('cuda:0')
z = load_golden(name)
Ls = [csr_from(z, 'L%d' % i) for i in range(int(z['nlevels']))]
F, K, p, M = z['F'].tolist(), z['K'].tolist(), z['p'].tolist(), z['M'].tolist()
x = torch.as_tensor(z['x']).to(dev); B = x.shape[0]
labels = torch.as_tensor(np.arange(B) % M[-1]).to(dev)
net = models_gcn.cgcnn({'device': dev}, Ls, F, K, p, M, channel=int(z['channel']), brelu=str(z['brelu']), batch_size=B, regularization=5e-4, dropout=1, verbose=False)
net.contraction = 'f32'
print(name, order, variant, 'p', p, 'relabelled', net._relabelled, [m is not None for m in net._pool_maps])
xs = ops.plane_storage(x)
if variant == 'nomaps':
    net._pool_maps = [None] * len(net._pool_maps)        # (wrong results, capture test only)
net.enable_step_graph(True)
_lib.dispatch_log = log = []
try:
    for i in range(4):
        log.clear()
        net.train_step(xs, labels)
        torch.cuda.synchronize()
        print('step', i, 'ok', 'captured' if net._sg is not None else 'eager')
except Exception as e:
    print('FAILED at', [w for w, _ in log][-6:], type(e).__name__, str(e)[:200])
