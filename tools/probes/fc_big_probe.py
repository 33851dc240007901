"""The first FC layer at the benchmark shape (64 x 10466 x 512) through the library: 20 eager calls (for rocprofv3 --kernel-trace
--stats) and the GPU time of a HIP-graph replay of 50 calls."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_fmri_decoding_amd import ops
dev = torch.device('cuda:0')
B, I, O = 64, 10466, 512
x = torch.randn(B, 10496, device=dev)[:, :I]; W = torch.randn(I, O, device=dev) * 0.01; b = torch.zeros(O, device=dev)
for _ in range(20):
    y = ops.fc_forward(x, W, b, True)
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(50): y = ops.fc_forward(x, W, b, True)
gr.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
print('fc_forward 64 x 10466 x 512: %.1f us per call' % (e0.elapsed_time(e1) * 1e3 / 50))
