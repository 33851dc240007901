#!/usr/bin/env python3
"""Which recurrence kernel template each (graph size, planes) reaches: the table tests/test_gpu_recurrence_shapes.py asserts."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from gcn_fmri_decoding_amd import _lib, ops
    dev = torch.device('cuda:0')
    lib = _lib.lib()
    P = ops._p
    cases = [(n, 1) for n in (40, 100, 212, 360, 500, 900, 1000, 1500, 2000, 2600, 4000, 5000, 6000, 8000, 10000, 10500, 13000, 16000, 19000)]
    cases += [(5000, 2), (5000, 3), (4600, 3), (10000, 6), (9000, 3)]
    for n, lv in cases:
        Ls, _ = bench.load_graph(n, lv, 0, 1, None)
        L = Ls[0]
        for planes in (0, 2, 4):
            try:
                g = ops.Graph(L, dev, planes=planes)
            except _lib.ChebgcnError as e:
                print('n=%d l=%d planes=%d: %s' % (n, lv, planes, str(e)[:60]))
                continue
            for (B, Fin) in ((2, 3), (256, 32)):
                K = 3
                x = torch.zeros(B, Fin, g.Mp, device=dev)
                st = torch.zeros(K, B, Fin, g.Mp, device=dev)
                dx = torch.zeros(B, Fin, g.Mp, device=dev)
                _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(x), P(st), B, Fin, K, ops._stream()), 'f')
                nf = _lib.last_dispatch()
                _lib.check(lib.chebgcn_recurrence_bwd(g.handle, P(st), P(dx), B, Fin, K, ops._stream()), 'b')
                na = _lib.last_dispatch()
                print('n=%d l=%d M=%d rows=%d planes=%d(%d) nplanes=%d: %s | %s' % (n, lv, g.M, g.query(7), planes, g.query(6), B * Fin, nf, na), flush=True)
                del x, st, dx
            torch.cuda.synchronize()


if __name__ == '__main__':
    main()
