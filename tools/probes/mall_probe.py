#!/usr/bin/env python3
"""Probe: does the stack of a layer survive in the 256 MB Infinity Cache between the recurrence that writes it and the contraction
that reads it, if a batch is processed in halves (214 MB per half at the bench shape)?  Times recurrence + contraction over 64
windows as one pair of launches and as two pairs over 32 windows each (same work, same kernels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from gcn_fmri_decoding_amd import _lib, ops, graph as G
dev = torch.device('cuda:0')
Ls, _ = bench.load_graph(10000, 1, 0, 1, None)
g = ops.Graph(Ls[0], dev, order=G.length_order(Ls[0]))
lib = _lib.lib(); P = ops._p; st = ops._stream()
B, F, K, Mp, M = 64, 32, 5, g.Mp, g.M
W = torch.randn(F * K, F, device=dev) * 0.1
bias = torch.randn(F, Mp, device=dev)
def run(parts, iters=30):
    Bp = B // parts
    stacks = [torch.randn(K, Bp, F, Mp, device=dev) for _ in range(parts)]
    outs = [torch.empty(Bp, F, Mp, device=dev) for _ in range(parts)]
    def once():
        for s, o in zip(stacks, outs):
            _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(s), P(s), Bp, F, K, st), 'rec')
            _lib.check(lib.chebgcn_contract_fwd(P(s), P(W), P(bias), 2, P(o), None, Bp, M, F, K, F, 1, 0, 1, st), 'con')
    for _ in range(3): once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): once()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for parts in (1, 2, 4, 1, 2, 4):
    print('batch 64 in %d part(s): %.4f ms per layer forward (recurrence + contraction)' % (parts, run(parts)), flush=True)
