"""The first FC layer at the benchmark shape (64 x 10466 x 512) through the library, 20 calls: for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_fmri_decoding_amd import ops
dev = torch.device('cuda:0')
B, I, O = 64, 10466, 512
x = torch.randn(B, 10496, device=dev)[:, :I]; W = torch.randn(I, O, device=dev) * 0.01; b = torch.zeros(O, device=dev)
for _ in range(20):
    y = ops.fc_forward(x, W, b, True)
torch.cuda.synchronize()
