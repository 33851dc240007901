#!/usr/bin/env python3
"""Per-kernel micro-benchmark of libchebgcn.so on the benchmark graph (M = 10466).

    python tools/kbench.py [--B 64 256] [--fin 32] [--fout 32] [--K 5] [--iters 20]
                           [--kernels recurrence_fwd ...]

Times each C-ABI entry point with HIP events on the launch stream and prints achieved
algorithmic GB/s (SURVEY.md 8d byte counts) and the fraction of the 8 TB/s HBM roofline.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, nargs='+', default=[64, 256])
    ap.add_argument('--fin', type=int, default=32)
    ap.add_argument('--fout', type=int, default=32)
    ap.add_argument('--K', type=int, default=5)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--kernels', nargs='+', default=['recurrence_fwd', 'recurrence_fwd_inplace', 'recurrence_bwd', 'contract_fwd',
                                                     'contract_bwd_w_relu', 'contract_bwd_x_relu', 'bias_grad_relu', 'contract_bwd_w', 'contract_bwd_x',
                                                     'brelu_pool_bwd', 'recurrence_fwd_t', 'contract_fwd_dx', 'brelu_pool_bwd_mask', 'contract_fwd_dx_gated',
                                                     'bias_grad_sum'])
    ap.add_argument('--nodes', type=int, default=10000, help='points of the synthetic kNN graph (10000 -> M = 10466)')
    ap.add_argument('--levels', type=int, default=1, help='coarsening levels of the synthetic graph (1 -> fake vertices behind the real ones)')
    ap.add_argument('--json', default=None)
    ap.add_argument('--planes', type=int, default=0, help='planes per workgroup of the recurrence kernel (0 = automatic, 2, 4)')
    ap.add_argument('--order', default='length', choices=['bank', 'length', 'reference'],
                    help="vertex order of the graph: 'length' = relabelled by descending row length as cgcnn does for a network "
                         "without pooling (ordered recurrence kernels); 'reference' = the caller's numbering")
    ap.add_argument('--stamps', action='store_true', help='print the in-kernel phase stamps of a CG_X&64 build (tools/xbuild.sh 64)')
    ap.add_argument('--stagger', type=int, default=0, help='chebgcn_tune(4, x): start stagger override (x-1 eighths), 0 = automatic')
    ap.add_argument('--wide', type=int, default=0, help='chebgcn_tune(3, x): 1 = 1024-thread recurrence shape')
    args = ap.parse_args()

    import torch
    import bench
    from gcn_fmri_decoding_amd import _lib, ops
    dev = torch.device('cuda:0')
    Ls, perm = bench.load_graph(args.nodes, args.levels, 0, 1, None)
    lib = _lib.lib()
    import ctypes
    handle = ctypes.CDLL(_lib.LIB_PATH)
    tune = getattr(handle, 'chebgcn_tune', None)          # experiment builds only (tools/xbuild.sh, CHEBGCN_LIB=...)
    if tune is None and (args.wide or args.stagger or (args.stamps and not hasattr(handle, 'chebgcn_debug_stampsb'))):
        raise SystemExit('--wide / --stagger / --stamps need an experiment build (tools/xbuild.sh)')
    if tune is not None:
        tune(3, args.wide)
        tune(4, args.stagger)
    from gcn_fmri_decoding_amd import graph as G
    # 'bank': the length order refined inside its equal-length classes against LDS bank conflicts (graph.bank_order; experiment)
    order = None if (args.order == 'reference' or args.planes) else G.length_order(Ls[0]) if args.order == 'length' else G.bank_order(Ls[0])
    g = ops.Graph(Ls[0], dev, planes=args.planes, order=order)
    print('ordered recurrence kernels:', bool(g.query(12)), ' planes of the ordered image:', g.query(16), flush=True)
    print('planes per workgroup:', g.query(6), ' gather LDS cost (before, after placement, ideal):', g.query(9), g.query(10),
          g.query(11), ' ordered image:', g.query(13), g.query(14), g.query(15), flush=True)
    M, Mp = g.M, g.Mp
    print('M = %d, Mp = %d (plane stride %d bytes)' % (M, Mp, 4 * Mp), flush=True)
    results = []

    def timeit(fn, iters):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for s, e in evs:
            s.record()
            fn()
            e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)
        return ms[len(ms) // 2], ms[0]

    for B in args.B:
        Fin, Fout, K = args.fin, args.fout, args.K
        if Fin != Fout:
            args.kernels = [k for k in args.kernels if k not in ('recurrence_fwd_t', 'contract_fwd_dx', 'contract_fwd_dx_gated')]
        torch.manual_seed(0)
        x = torch.randn(B, Fin, Mp, device=dev)
        stack = torch.randn(K, B, Fin, Mp, device=dev)
        gstack = torch.randn(K, B, Fin, Mp, device=dev)
        dx = torch.empty(B, Fin, Mp, device=dev)
        W = torch.randn(Fin * K, Fout, device=dev) * 0.1
        Wt = torch.randn(Fout * K, Fin, device=dev) * 0.1       # (gstack below is [K, B, Fin, Mp]: the dx entries assume Fin == Fout)
        bias = torch.randn(Fout, Mp, device=dev)
        out = torch.empty(B, Fout, Mp, device=dev)
        dy = torch.randn(B, Fout, Mp, device=dev)
        dy16 = dy.to(torch.bfloat16)
        dbias = torch.empty(Fout, Mp, device=dev)
        dW = torch.empty(Fin * K, Fout, device=dev)
        ws = torch.empty(lib.chebgcn_contract_bwd_w_workspace(B, M, Fin, K, Fout), dtype=torch.uint8, device=dev)
        ws16 = torch.empty(lib.chebgcn_contract_fwd_bf16_workspace(Fin, K, Fout), dtype=torch.uint8, device=dev)
        wsx16 = torch.empty(lib.chebgcn_contract_bwd_x_bf16_workspace(Fin, K, Fout), dtype=torch.uint8, device=dev)
        wsw16 = torch.empty(lib.chebgcn_contract_bwd_w_bf16_workspace(B, M, Fin, K, Fout), dtype=torch.uint8, device=dev)
        mask = torch.randint(0, 16, (B, Fout, Mp // 4), dtype=torch.uint8, device=dev)     # ReLU bit mask, half the bits set
        st = ops._stream()
        P = ops._p
        calls = {
            'recurrence_fwd': (lambda: lib.chebgcn_recurrence_fwd(g.handle, P(x), P(stack), B, Fin, K, st),
                               4.0 * M * Fin * K * B, 0.0),
            # T_0 already in slab 0 of the stack (what the model does): no copy of the input
            'recurrence_fwd_inplace': (lambda: lib.chebgcn_recurrence_fwd(g.handle, P(stack), P(stack), B, Fin, K, st),
                                       4.0 * M * Fin * K * B, 0.0),
            'recurrence_bwd': (lambda: lib.chebgcn_recurrence_bwd(g.handle, P(gstack), P(dx), B, Fin, K, st),
                               4.0 * M * Fin * (K + 1) * B, 0.0),
            # the input gradient as the training step forms it (ops.dx_by_forward): the FORWARD kernel on the planes of dy with the
            # transposed operator, in place in slab 0 of the gradient stack, then the forward contraction on the re-indexed weights
            # (no bias, no ReLU, no mask)
            'recurrence_fwd_t': (lambda: lib.chebgcn_recurrence_fwd_t(g.handle, P(gstack), P(gstack), B, Fout, K, st),
                                 4.0 * M * Fout * K * B, 0.0),
            'contract_fwd_dx': (lambda: lib.chebgcn_contract_fwd(P(gstack), P(Wt), None, 0, P(dx), None, B, M, Fout, K, Fin,
                                                                 1, 0, 0, st),
                                4.0 * B * M * (Fout * K + Fin), 2.0 * B * M * Fin * K * Fout),
            # ... with the ReluGrad of the layer below in its epilogue (ops.GateLink: layers 3-6 of the bench network), and what is
            # left of that layer's own ReluGrad pass: the plain sum of the gated dy for the bias gradient
            'contract_fwd_dx_gated': (lambda: lib.chebgcn_contract_fwd_gated(P(gstack), P(Wt), P(mask), P(dx), B, M, Fout, K, Fin, st),
                                      B * M * (4.0 * (Fout * K + Fin) + Fin / 4.0), 2.0 * B * M * Fin * K * Fout),
            'bias_grad_sum': (lambda: lib.chebgcn_brelu_pool_bwd(P(dy), None, None, None, P(dbias), 2, B, M, Fout, 1, 0, 0, None, 0, st),
                              B * Fout * M * 4.0, 0.0),
            'contract_fwd': (lambda: lib.chebgcn_contract_fwd(P(stack), P(W), P(bias), 2, P(out), None, B, M, Fin, K, Fout,
                                                              1, 0, 1, st),
                             4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout),
            'contract_fwd_bf16': (lambda: lib.chebgcn_contract_fwd_bf16(P(stack), P(W), P(bias), 2, P(out), None, B, M, Fin, K,
                                                                        Fout, 1, 0, 1, 1, P(ws16), ws16.numel(), st),
                                  4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout),
            'contract_fwd_bf16x3': (lambda: lib.chebgcn_contract_fwd_bf16(P(stack), P(W), P(bias), 2, P(out), None, B, M, Fin,
                                                                          K, Fout, 1, 0, 1, 3, P(ws16), ws16.numel(), st),
                                    4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout),
            'contract_bwd_w': (lambda: lib.chebgcn_contract_bwd_w(P(stack), P(dy), P(dW), P(ws), ws.numel(), B, M, Fin, K,
                                                                  Fout, st),
                               4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout),
            'contract_bwd_x': (lambda: lib.chebgcn_contract_bwd_x(P(dy), P(W), P(gstack), B, M, Fin, K, Fout, st),
                               4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout),
            # the gradients of a pool == 1 ReLU layer as the model runs them: ReluGrad folded in (bit mask)
            'contract_bwd_w_relu': (lambda: lib.chebgcn_contract_bwd_w_relu(P(stack), P(dy), P(mask), P(dW), P(ws), ws.numel(), B, M,
                                                                            Fin, K, Fout, st),
                                    B * M * (4.0 * (Fin * K + Fout) + Fout / 4.0), 2.0 * B * M * Fin * K * Fout),
            'contract_bwd_x_relu': (lambda: lib.chebgcn_contract_bwd_x_relu(P(dy), P(mask), P(W), P(gstack), B, M, Fin, K, Fout, st),
                                    B * M * (4.0 * (Fin * K + Fout) + Fout / 4.0), 2.0 * B * M * Fin * K * Fout),
            'bias_grad_relu': (lambda: lib.chebgcn_brelu_pool_bwd(P(dy), None, P(mask), None, P(dbias), 2, B, M, Fout, 1, 0, 1, None, 0, st),
                               B * Fout * M * 4.25, 0.0),
            'contract_bwd_w_bf16': (lambda: lib.chebgcn_contract_bwd_w_bf16(P(stack), P(dy), P(dW), P(wsw16), wsw16.numel(), B, M, Fin,
                                                                            K, Fout, 1, st),
                                    4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout),
            'contract_bwd_w_bf16x3': (lambda: lib.chebgcn_contract_bwd_w_bf16(P(stack), P(dy), P(dW), P(wsw16), wsw16.numel(), B, M,
                                                                              Fin, K, Fout, 3, st),
                                      4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout),
            'contract_bwd_x_bf16': (lambda: lib.chebgcn_contract_bwd_x_bf16(P(dy), P(W), P(gstack), B, M, Fin, K, Fout, 1, P(wsx16),
                                                                            wsx16.numel(), st),
                                    4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout),
            'contract_bwd_x_bf16x3': (lambda: lib.chebgcn_contract_bwd_x_bf16(P(dy), P(W), P(gstack), B, M, Fin, K, Fout, 3, P(wsx16),
                                                                              wsx16.numel(), st),
                                      4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout),
            # wide bf16 layers: ReluGrad writes dy as bf16, the two gradients read it (same results as the fp32-dy entries above)
            'relu_grad_bf16': (lambda: lib.chebgcn_relu_grad_bf16(P(dy), P(mask), P(dy16), P(dbias), 2, B, M, Fout, None, 0, st),
                               B * Fout * M * 6.25, 0.0),
            'brelu_pool_bwd_mask': (lambda: lib.chebgcn_brelu_pool_bwd(P(dy), None, P(mask), P(out), P(dbias), 2, B, M, Fout, 1, 0, 1,
                                                                       None, 0, st), B * Fout * M * 8.25, 0.0),
            'contract_bwd_w_bf16_dy16': (lambda: lib.chebgcn_contract_bwd_w_bf16_dy16(P(stack), P(dy16), P(dW), P(wsw16), wsw16.numel(),
                                                                                      B, M, Fin, K, Fout, st),
                                         4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout),
            'contract_bwd_x_bf16_dy16': (lambda: lib.chebgcn_contract_bwd_x_bf16_dy16(P(dy16), P(W), P(gstack), B, M, Fin, K, Fout,
                                                                                      P(wsx16), wsx16.numel(), st),
                                         4.0 * B * M * (Fin * K + Fout), 2.0 * B * M * Fin * K * Fout),
            'brelu_pool_bwd': (lambda: lib.chebgcn_brelu_pool_bwd(P(dy), P(out), None, P(dx) if Fin == Fout else P(out),
                                                                  P(dbias), 2, B, M, Fout, 1, 0, 1, None, 0, st),
                               4.0 * B * Fout * 3 * M, 0.0),
        }
        for name in args.kernels:
            fn, nbytes, flops = calls[name]
            for abl in [0]:
                med, best = timeit(lambda: _lib.check(fn(), name), args.iters)
                r = {'kernel': name, 'B': B, 'Fin': Fin, 'Fout': Fout, 'K': K, 'ablate': abl, 'median_ms': med,
                     'min_ms': best, 'GBps': nbytes / med / 1e6, 'frac_hbm': nbytes / med / 1e6 / 8000.0,
                     'TFLOPs': flops / med / 1e9}
                results.append(r)
                if args.stamps and (name.startswith('recurrence') or (hasattr(handle, 'chebgcn_debug_stampsb') and 'bf16' in name)):
                    buf = (ctypes.c_longlong * (16 * 64))()
                    if not name.startswith('recurrence'):
                        assert handle.chebgcn_debug_stampsb(buf) == 0       # tools/bbuild.sh x64 "-DCG_EXPERIMENT=1 -DCG_X=64"
                    else:
                        assert (handle.chebgcn_debug_stampso if g.query(12) else handle.chebgcn_debug_stamps4 if g.query(6) == 4 and g.query(7) > 2048 else handle.chebgcn_debug_stamps)(buf) == 0
                    t = np.array(buf, dtype=np.int64).reshape(16, 64)
                    t0 = t[t > 0].min()
                    print('   stamps (cycle counter ticks since the first wave entered the group), one row per wave:')
                    ids = [i for i in range(64) if (t[:, i] > 0).any()]
                    print('   id   ' + ' '.join('%7d' % i for i in ids))
                    for w in range(16):
                        if (t[w] > 0).any():
                            print('   w%-3d ' % w + ' '.join('%7d' % (t[w, i] - t0 if t[w, i] > 0 else -1) for i in ids))
                print('%-16s B=%-4d abl=%-2d  %8.3f ms (min %7.3f)  %7.0f GB/s  %5.1f%% of 8 TB/s  %6.1f TFLOP/s   %s'
                      % (name, B, abl, med, best, r['GBps'], 100 * r['frac_hbm'], r['TFLOPs'], _lib.last_dispatch()), flush=True)
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(results, f, indent=1)


if __name__ == '__main__':
    main()
