#!/usr/bin/env python3
"""The reference's own training shape (atlas-sized graph, K = 10 x 6, batch 128; bench.refshape_leg) alone -- the program
`rocprofv3 --kernel-trace --stats -- python3 tools/refshape.py --nodes 360` profiles for profiles/r03_refshape_*."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument('--nodes', type=int, default=360)
ap.add_argument('--steps', type=int, default=50)
ap.add_argument('--warmup', type=int, default=10)
ap.add_argument('--bias-side', type=int, default=1, help='ops.bias_side_small (A/B)')
args = ap.parse_args()
import torch
import bench
from gcn_fmri_decoding_amd import ops
ops.bias_side_small = bool(args.bias_side)
print(json.dumps(bench.refshape_leg(torch.device('cuda:0'), args.nodes, args.steps, args.warmup)))
