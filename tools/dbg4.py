#!/usr/bin/env python3
"""Localises differences between the two- and four-plane recurrence kernels (debug aid)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcn_fmri_decoding_amd import _lib, ops, graph

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
B, Fin, K = 3, 3, 5
dev = torch.device('cuda:0')
Ls, perm, _ = graph.synthetic_graph(nodes, k=8, levels=1)
L = Ls[0]; M = L.shape[0]
g2 = ops.Graph(L, dev, planes=2); g4 = ops.Graph(L, dev, planes=4)
print('M', M, 'planes', g2.query(6), g4.query(6), 'rows', g4.query(7))
torch.manual_seed(0)
x = torch.randn(B, Fin, g2.Mp, device=dev); x[:, :, M:] = 0
G = torch.randn(K, B, Fin, g2.Mp, device=dev); G[:, :, :, M:] = 0
lib = _lib.lib()
def run(g):
    st = torch.full((K, B, Fin, g.Mp), float('nan'), device=dev); dx = torch.full((B, Fin, g.Mp), float('nan'), device=dev)
    _lib.check(lib.chebgcn_recurrence_fwd(g.handle, ops._p(x), ops._p(st), B, Fin, K, ops._stream()), 'f')
    _lib.check(lib.chebgcn_recurrence_bwd(g.handle, ops._p(G), ops._p(dx), B, Fin, K, ops._stream()), 'b')
    torch.cuda.synchronize()
    return st, dx
s2, d2 = run(g2); s4, d4 = run(g4)
import scipy.sparse as sp
Lr = sp.csr_matrix(L); deg = np.diff(Lr.indptr)
rs = graph.rescaled_laplacian_csr(L)
rowlen = np.diff(rs[0]); 
collen = np.bincount(rs[1], minlength=M)
iso = (rowlen == 0) & (collen == 0)
print('isolated', iso.sum())
isot = torch.as_tensor(iso).to(dev)
for k in range(K):
    for b in range(B):
        for f in range(Fin):
            a, c = s4[k, b, f, :M], s2[k, b, f, :M]
            bad = ~torch.isclose(a, c, rtol=1e-4, atol=1e-5)
            if bad.any():
                idx = bad.nonzero().flatten()
                print('slab', k, 'plane', b * Fin + f, 'bad', int(bad.sum()), 'of which iso', int((bad & isot).sum()), 'nan', int(torch.isnan(a).sum()),
                      'first', idx[:6].tolist(), a[idx[:3]].tolist(), c[idx[:3]].tolist())
for b in range(B):
    for f in range(Fin):
        a, c = d4[b, f, :M], d2[b, f, :M]
        bad = ~torch.isclose(a, c, rtol=1e-4, atol=1e-5)
        if bad.any():
            idx = bad.nonzero().flatten()
            print('dx plane', b * Fin + f, 'bad', int(bad.sum()), 'iso', int((bad & isot).sum()), 'first', idx[:6].tolist(), a[idx[:3]].tolist(), c[idx[:3]].tolist())
print('done')
