#!/usr/bin/env python3
"""chebgcn_contract_fwd against chebgcn_contract_fwd_gated at the bench launch (HIP events, median of 50)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_fmri_decoding_amd import _lib, ops
lib = _lib.lib()
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, M, Fin, K, Fout = 64, 10466, 32, 5, 32
Mp = ops.plane_stride(M)
stack = torch.randn((K, B, Fin, Mp), device=dev)
W = torch.randn((Fin * K, Fout), device=dev) * 0.1
gate = torch.randint(0, 16, (B, Fout, Mp // 4), device=dev, dtype=torch.uint8)
out = torch.empty((B, Fout, Mp), device=dev)
def t(fn, n=50):
    for _ in range(5): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
print('contract_fwd       %.4f ms' % t(lambda: lib.chebgcn_contract_fwd(P(stack), P(W), None, 0, P(out), None, B, M, Fin, K, Fout, 1, 0, 0, st())))
print('contract_fwd_gated %.4f ms  (CHEBGCN_GATED_LDS=%s)' % (t(lambda: lib.chebgcn_contract_fwd_gated(P(stack), P(W), P(gate), P(out), B, M, Fin, K, Fout, st())), os.environ.get('CHEBGCN_GATED_LDS', '0')))
