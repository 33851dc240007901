"""Timing of the first FC layer's three skinny GEMMs (64 x 10466 x 512, fp32) in the formulations torch / hipBLASLt offer."""
import torch, time
dev = torch.device('cuda:0')
import sys
B, M, O = (int(sys.argv[1]), int(sys.argv[2]), 512) if len(sys.argv) > 2 else (64, 10466, 512)
x = torch.randn(B, M, device=dev); W = torch.randn(M, O, device=dev) * 0.01; dy = torch.randn(B, O, device=dev)
Wt = W.t().contiguous(); xt = x.t().contiguous(); dyt = dy.t().contiguous()
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ev)
    return v[len(v) // 2] * 1e3
out = torch.empty(B, O, device=dev); dW = torch.empty(M, O, device=dev); dx = torch.empty(B, M, device=dev)
dWt = torch.empty(O, M, device=dev); dxt = torch.empty(M, B, device=dev); outt = torch.empty(O, B, device=dev)
print('fwd  x[64,M] @ W[M,512]                : %.1f us' % t(lambda: torch.mm(x, W, out=out)))
print('fwd  x[64,M] @ Wt[512,M].t()            : %.1f us' % t(lambda: torch.mm(x, Wt.t(), out=out)))
print('fwd  xt[M,64].t() @ W[M,512]            : %.1f us' % t(lambda: torch.mm(xt.t(), W, out=out)))
print('fwd  (Wt[512,M] @ xt[M,64]) -> [512,64] : %.1f us' % t(lambda: torch.mm(Wt, xt, out=outt)))
print('fwd  (W.t() @ xt)                       : %.1f us' % t(lambda: torch.mm(W.t(), xt, out=outt)))
print('dW   x.t()[M,64] @ dy[64,512]           : %.1f us' % t(lambda: torch.mm(x.t(), dy, out=dW)))
print('dW   xt[M,64] @ dy[64,512]              : %.1f us' % t(lambda: torch.mm(xt, dy, out=dW)))
print('dWt  dy.t()[512,64] @ x[64,M]           : %.1f us' % t(lambda: torch.mm(dy.t(), x, out=dWt)))
print('dWt  dyt[512,64] @ x[64,M]              : %.1f us' % t(lambda: torch.mm(dyt, x, out=dWt)))
print('dx   dy[64,512] @ W.t()[512,M]          : %.1f us' % t(lambda: torch.mm(dy, W.t(), out=dx)))
print('dx   dy[64,512] @ Wt[512,M]             : %.1f us' % t(lambda: torch.mm(dy, Wt, out=dx)))
print('dxt  W[M,512] @ dy.t()[512,64]          : %.1f us' % t(lambda: torch.mm(W, dy.t(), out=dxt)))
print('dxt  W[M,512] @ dyt[512,64]             : %.1f us' % t(lambda: torch.mm(W, dyt, out=dxt)))
print('copy 21.4 MB (reference point)          : %.1f us' % t(lambda: dW.copy_(W)))
