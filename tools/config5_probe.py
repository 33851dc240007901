#!/usr/bin/env python3
"""BASELINE configs[4] / configs[3] (one wide layer, forward + backward in fp32 / bf16 / split bf16) alone: bench.layer_leg
without the rest of bench.py.  CHEBGCN_LIB selects a library variant (tools/bbuild.sh).  SHAPE="Fin K Fout" (default 60 5 256)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from gcn_fmri_decoding_amd import graph, ops
    dev = torch.device('cuda:0')
    Ls, _ = bench.load_graph(10000, 1, 0, 1, None)
    L = Ls[0]
    g = ops.Graph(graph.permute(L, graph.length_order(L)), dev)
    Fin, K, Fout = (int(v) for v in os.environ.get('SHAPE', '60 5 256').split())
    out = bench.layer_leg(g, 64, Fin, K, Fout, int(os.environ.get('STEPS', 10)), 'one wide layer, forward + backward')
    print('Fin = %d, K = %d, Fout = %d; precision auto -> %s' % (Fin, K, Fout, out['default_precision']))
    for p in ('f32', 'bf16', 'bf16x3'):
        leg = out[p]
        print('%-7s %.3f ms/step  %s' % (p, leg['ms_per_step'], json.dumps(leg.get('rel_err_vs_f32'))))
        for k, v in leg['kernels'].items():
            print('      %-24s %.4f ms  %6.0f GB/s' % (k, v['avg_ms'], v['GBps']))


if __name__ == '__main__':
    main()
