#!/usr/bin/env python3
"""Ordered recurrence kernels (csrc/recurrence_ord.hip) against the CPU oracle and against the unordered kernels:
values of every plane of a short launch, then timings at the bench / north-star / config-4 shapes."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from gcn_fmri_decoding_amd import _lib, graph, ops
    from oracle import graph_ref as GR
    dev = torch.device('cuda:0')
    lib = _lib.lib()
    Ls, _ = bench.load_graph(10000, 1, 0, 1, None)
    L = Ls[0]
    M = L.shape[0]
    order = graph.length_order(L)
    Lp = graph.permute(L, order)
    g0 = ops.Graph(L, dev)
    g1 = ops.Graph(Lp, dev)
    print('ordered flags:', g0.query(12), g1.query(12), ' LDS cost old image (before, placed, ideal):', g0.query(9), g0.query(10), g0.query(11),
          ' ordered image:', g1.query(13), g1.query(14), g1.query(15), flush=True)
    assert g1.query(12) == 1
    Mp = g1.Mp
    P = ops._p
    st = ops._stream()
    Lr = GR.rescale_L(Lp, 2)
    LT = Lr.T.tocsr().astype(np.float64)
    for (B, Fin, K) in [(3, 5, 5), (2, 4, 2), (2, 7, 6)]:
        gen = torch.Generator(device=dev)
        gen.manual_seed(B + Fin)
        x = torch.randn((B, Fin, Mp), generator=gen, device=dev)
        x[..., M:] = float('nan')
        for inplace in (False, True):
            stack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
            if inplace:
                stack[0].copy_(x)
                _lib.check(lib.chebgcn_recurrence_fwd(g1.handle, P(stack), P(stack), B, Fin, K, st), 'fwd')
            else:
                _lib.check(lib.chebgcn_recurrence_fwd(g1.handle, P(x), P(stack), B, Fin, K, st), 'fwd')
            name = _lib.last_dispatch()
            worst = 0.0
            for b in range(B):
                xv = x[b, :, :M].cpu().numpy().T.astype(np.float32)
                T = [xv, (Lr @ xv).astype(np.float32)]
                for k in range(2, K):
                    T.append((2 * (Lr @ T[-1]) - T[-2]).astype(np.float32))
                ref = np.stack(T[:K]).transpose(0, 2, 1)
                got = stack[:, b, :, :M].cpu().numpy()
                worst = max(worst, np.abs(got - ref).max() / np.abs(ref).max())
            print('fwd B=%d Fin=%d K=%d inplace=%d: %s  rel err %.2e' % (B, Fin, K, inplace, name, worst), flush=True)
            assert worst < 1e-5
        G = torch.randn((K, B, Fin, Mp), generator=gen, device=dev)
        G[..., M:] = float('nan')
        dx = torch.full((B, Fin, Mp), float('nan'), device=dev)
        _lib.check(lib.chebgcn_recurrence_bwd(g1.handle, P(G), P(dx), B, Fin, K, st), 'bwd')
        name = _lib.last_dispatch()
        worst = 0.0
        for b in range(B):
            Gb = G[:, b, :, :M].cpu().numpy().transpose(0, 2, 1).astype(np.float64)
            c1, c2 = Gb[K - 1], np.zeros_like(Gb[0])
            for j in range(K - 2, 0, -1):
                c1, c2 = Gb[j] + 2 * (LT @ c1) - c2, c1
            dref = Gb[0] + LT @ c1 - c2 if K > 1 else Gb[0]
            worst = max(worst, np.abs(dx[b, :, :M].cpu().numpy().T - dref).max() / np.abs(dref).max())
        print('bwd B=%d Fin=%d K=%d: %s  rel err %.2e' % (B, Fin, K, name, worst), flush=True)
        assert worst < 2e-5

    def timeit(fn, iters=30):
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
        return ms[len(ms) // 2]

    handle = ctypes.CDLL(_lib.LIB_PATH)
    stamps = getattr(handle, 'chebgcn_debug_stampso', None)

    def show_stamps(tag):
        if stamps is None:
            return
        buf = (ctypes.c_longlong * (16 * 64))()
        assert stamps(buf) == 0
        t = np.array(buf, dtype=np.int64).reshape(16, 64)
        if not (t[:, 0] > 0).any():
            return
        t0 = t[:, 0][t[:, 0] > 0].min()
        ids = [i for i in range(64) if (t[:, i] > 0).any()]
        print('   stamps %s' % tag)
        print('   id   ' + ' '.join('%7d' % i for i in ids))
        for w in range(16):
            if t[w, 0] > 0:
                print('   w%-3d ' % w + ' '.join('%7d' % (t[w, i] - t0) for i in ids))

    shapes = [(64, 32, 5), (64, 15, 5), (256, 32, 5), (64, 64, 25)] if stamps is None else [(256, 32, 5)]
    for (B, Fin, K) in shapes:
        stack = torch.randn(K, B, Fin, Mp, device=dev)
        gstack = torch.randn(K, B, Fin, Mp, device=dev)
        dx = torch.empty(B, Fin, Mp, device=dev)
        for tag, g in (('old', g0), ('ord', g1)):
            tf = timeit(lambda: lib.chebgcn_recurrence_fwd(g.handle, P(stack), P(stack), B, Fin, K, st))
            tb = timeit(lambda: lib.chebgcn_recurrence_bwd(g.handle, P(gstack), P(dx), B, Fin, K, st))
            bf, bb = 4.0 * M * Fin * K * B, 4.0 * M * Fin * (K + 1) * B
            print('%s B=%d Fin=%d K=%d  fwd %.4f ms (%.3f of 8 TB/s)   bwd %.4f ms (%.3f)' % (
                tag, B, Fin, K, tf, bf / tf / 8e9, tb, bb / tb / 8e9), flush=True)
            if tag == 'ord':
                lib.chebgcn_recurrence_fwd(g.handle, P(stack), P(stack), B, Fin, K, st)
                torch.cuda.synchronize()
                show_stamps('fwd')
                lib.chebgcn_recurrence_bwd(g.handle, P(gstack), P(dx), B, Fin, K, st)
                torch.cuda.synchronize()
                show_stamps('bwd')


if __name__ == '__main__':
    main()
