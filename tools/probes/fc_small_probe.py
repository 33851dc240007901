"""GPU time (HIP-graph replay of 50 calls) of the FC head's forward GEMMs in the formulations torch offers, hipBLASLt and rocBLAS."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_fmri_decoding_amd import ops
dev = torch.device('cuda:0')
def t(name, fn, n=50):
    try:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(n): fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        print('  %-40s %7.1f us' % (name, e0.elapsed_time(e1) * 1e3 / n))
    except Exception as e:
        print('  %-40s failed: %s' % (name, str(e)[:80]))
for lib in ('cublaslt',):
    torch.backends.cuda.preferred_blas_library(lib)
    for B, I, O in ((64, 10466, 512), (64, 512, 256), (64, 256, 22), (128, 360, 512), (128, 1000, 512), (128, 512, 256), (128, 256, 22)):
        print(lib, B, I, O)
        x = torch.randn(B, (I + 31) // 32 * 32, device=dev)[:, :I]; W = torch.randn(I, O, device=dev) * 0.01; b = torch.zeros(O, device=dev); g = torch.randn(B, O, device=dev)
        Wt = W.t().contiguous()
        t('fwd library kernel (bias + ReLU fused)', lambda: ops.fc_forward(x, W, b, True))
        if I > 4096:
            ops.FC_BWD_MAX_INNER = 1 << 20
        y = ops.fc_forward(x, W, b, True); dW = torch.empty_like(W); db = torch.empty_like(b)
        t('bwd library kernels (ReluGrad, dW, db, dx)', lambda: ops.fc_backward(x, W, g, y, dW, db, True))
        t('bwd library kernels (ReluGrad, dW, db)', lambda: ops.fc_backward(x, W, g, y, dW, db, False))
        t('bwd torch (threshold, mm, sum, mm)', lambda: (lambda gm: (torch.mm(x.t(), gm, out=dW), torch.sum(gm, 0, out=db), gm @ W.t()))(torch.ops.aten.threshold_backward(g, y, 0.0)))
        t('fwd addmm(b, x, W)', lambda: torch.addmm(b, x, W))
        t('fwd mm(x, W)', lambda: torch.mm(x, W))
        t('fwd linear(x, Wt, b)', lambda: torch.nn.functional.linear(x, Wt, b))
        t('fwd bmm', lambda: torch.bmm(x[None], W[None]))
        t('fwd (W.t() @ x.t())', lambda: torch.mm(W.t(), x.t()))
        t('dW mm(x.t(), g)', lambda: torch.mm(x.t(), g))
        t('dx mm(g, W.t())', lambda: torch.mm(g, W.t()))
