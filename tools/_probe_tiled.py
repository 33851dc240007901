"""Probe: the contraction kernels on a stack whose (window, 512-vertex tile) rows are contiguous -- emulated with the existing
layout by M = 512, B' = B * tiles -- against the real shape (same bytes).  Premise check for a tiled stack layout."""
import sys, torch
sys.path.insert(0, '.')
from gcn_fmri_decoding_amd import _lib, ops
lib = _lib.lib(); dev = torch.device('cuda:0'); P = ops._p; st = ops._stream

def timeit(run, n=40):
    for _ in range(5): run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); run(); e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    return ms[n // 2], ms[0]

Fin, K, Fout = 32, 5, 32
for (B, M) in ((64, 10466), (64 * 41, 256), (64 * 21, 512), (64 * 11, 1024), (256, 10466), (256 * 21, 512)):
    Mp = ops.plane_stride(M)
    stack = torch.randn((K, B, Fin, Mp), device=dev); W = torch.randn((Fin * K, Fout), device=dev)
    bias = torch.randn((Fout, Mp), device=dev)
    out = torch.empty((B, Fout, Mp), device=dev); mask = torch.empty((B, Fout, Mp // 4), dtype=torch.uint8, device=dev)
    dy = torch.randn((B, Fout, Mp), device=dev); dW = torch.empty((Fin * K, Fout), device=dev)
    n = lib.chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout); ws = torch.empty(n, dtype=torch.uint8, device=dev)
    nbytes = 4.0 * B * M * (Fin * K + Fout)
    t = timeit(lambda: _lib.check(lib.chebgcn_contract_fwd(P(stack), P(W), P(bias), 2, P(out), P(mask), B, M, Fin, K, Fout, 1, 0, 1, st()), 'f'))
    name = _lib.last_dispatch()
    print('contract_fwd  B=%5d M=%5d  %.4f ms (min %.4f)  %.0f GB/s  %s' % (B, M, t[0], t[1], nbytes / t[0] / 1e6, name))
    t = timeit(lambda: _lib.check(lib.chebgcn_contract_bwd_w(P(stack), P(dy), P(dW), P(ws), n, B, M, Fin, K, Fout, st()), 'w'))
    name = _lib.last_dispatch()
    print('contract_bwd_w B=%5d M=%5d  %.4f ms (min %.4f)  %.0f GB/s  %s' % (B, M, t[0], t[1], nbytes / t[0] / 1e6, name))
