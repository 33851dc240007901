#!/usr/bin/env python3
"""Fused atlas-size layer kernels (csrc/fused_small.hip) against the separate kernels of the library: forward output, ReLU
mask, stack; gradient wrt the input; timings at the reference's training shape."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from gcn_fmri_decoding_amd import _lib, graph, ops
    dev = torch.device('cuda:0')
    lib = _lib.lib()
    P, st = ops._p, ops._stream()
    for (N, B, Fin, K, Fout, bias_kind) in [(360, 128, 32, 10, 32, 2), (360, 128, 15, 10, 32, 2), (246, 5, 7, 3, 20, 1), (200, 3, 32, 1, 32, 0),
                                            (360, 4, 32, 2, 9, 2), (360, 300, 32, 10, 32, 2), (246, 200, 20, 4, 30, 1)]:
        Ls, perm, _ = graph.synthetic_graph(N, k=8, levels=1)
        g = ops.Graph(Ls[0], dev)
        M, Mp = g.M, g.Mp
        ok = lib.chebgcn_fused_layer_supported(g.handle, B, Fin, K, Fout)
        print('N=%d M=%d B=%d Fin=%d K=%d Fout=%d bias=%d supported=%d max row %d' % (N, M, B, Fin, K, Fout, bias_kind, ok, g.query(5)), flush=True)
        if not ok:
            continue
        gen = torch.Generator(device=dev)
        gen.manual_seed(N + Fin)
        x = torch.randn((B, Fin, Mp), generator=gen, device=dev)
        x[..., M:] = float('nan')
        W = torch.randn((Fin * K, Fout), generator=gen, device=dev) * (0.5 / np.sqrt(Fin * K))
        bias = None
        if bias_kind == 2:
            bias = torch.zeros((Fout, Mp), device=dev)
            bias[:, :M] = torch.randn((Fout, M), generator=gen, device=dev) * 0.3
        elif bias_kind == 1:
            bias = torch.randn((Fout,), generator=gen, device=dev) * 0.3
        relu = 1 if bias_kind else 0
        nws = lib.chebgcn_fused_layer_workspace(g.handle, B, Fin, K, Fout)
        ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
        # reference: the separate kernels
        stack_r = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
        _lib.check(lib.chebgcn_recurrence_fwd(g.handle, P(x), P(stack_r), B, Fin, K, st), 'rec')
        out_r = torch.full((B, Fout, Mp), float('nan'), device=dev)
        mask_r = torch.zeros((B, Fout, Mp // 4), dtype=torch.uint8, device=dev)
        _lib.check(lib.chebgcn_contract_fwd(P(stack_r), P(W), P(bias), bias_kind, P(out_r), P(mask_r) if relu else None, B, M, Fin, K, Fout, 1, 0, relu, st), 'con')
        for with_stack in (True, False):
            stack = torch.full((K, B, Fin, Mp), float('nan'), device=dev) if with_stack else None
            out = torch.full((B, Fout, Mp), float('nan'), device=dev)
            mask = torch.zeros((B, Fout, Mp // 4), dtype=torch.uint8, device=dev)
            _lib.check(lib.chebgcn_fused_layer_fwd(g.handle, P(x), P(W), P(bias), bias_kind, P(stack), P(out), P(mask) if relu else None, P(ws), nws,
                                                   B, Fin, K, Fout, relu, st), 'fused fwd')
            name = _lib.last_dispatch()
            e = float((out[..., :M] - out_r[..., :M]).abs().max() / out_r[..., :M].abs().max())
            print('   fwd %s stack=%d: out rel err %.2e' % (name, with_stack, e), end='')
            assert e < 1e-5
            if with_stack:
                es = float((stack[..., :M] - stack_r[..., :M]).abs().max() / stack_r[..., :M].abs().max())
                print('  stack %.2e' % es, end='')
                assert es < 1e-5
            if relu:
                bits = torch.stack([(mask >> r) & 1 for r in range(4)], -1).reshape(B, Fout, Mp)[..., :M].bool()
                assert torch.equal(bits, out[..., :M] > 0), 'mask'
                print('  mask ok', end='')
            print(flush=True)
        # gradient wrt the input
        dout = torch.randn((B, Fout, Mp), generator=gen, device=dev)
        dout[..., M:] = float('nan')
        gstack = torch.full((K, B, Fin, Mp), float('nan'), device=dev)
        if relu:
            _lib.check(lib.chebgcn_contract_bwd_x_relu(P(dout), P(mask_r), P(W), P(gstack), B, M, Fin, K, Fout, st), 'bwx')
        else:
            d0 = dout.clone(); d0[..., M:] = 0
            _lib.check(lib.chebgcn_contract_bwd_x(P(d0), P(W), P(gstack), B, M, Fin, K, Fout, st), 'bwx')
        dx_r = torch.full((B, Fin, Mp), float('nan'), device=dev)
        _lib.check(lib.chebgcn_recurrence_bwd(g.handle, P(gstack), P(dx_r), B, Fin, K, st), 'recb')
        dx = torch.full((B, Fin, Mp), float('nan'), device=dev)
        _lib.check(lib.chebgcn_fused_layer_bwd_x(g.handle, P(dout), P(mask_r) if relu else None, P(W), P(dx), B, Fin, K, Fout, st), 'fused bwd')
        e = float((dx[..., :M] - dx_r[..., :M]).abs().max() / dx_r[..., :M].abs().max())
        print('   bwd %s: dx rel err %.2e' % (_lib.last_dispatch(), e), flush=True)
        assert e < 2e-5

        def timeit(fn, iters=30):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
            for s, e in evs:
                s.record(); fn(); e.record()
            torch.cuda.synchronize()
            ms = sorted(s.elapsed_time(e) for s, e in evs)
            return 1e3 * ms[len(ms) // 2]
        if B >= 64:
            stack = torch.empty((K, B, Fin, Mp), device=dev)
            out = torch.empty((B, Fout, Mp), device=dev)
            mask = torch.zeros((B, Fout, Mp // 4), dtype=torch.uint8, device=dev)
            t_sep = timeit(lambda: (lib.chebgcn_recurrence_fwd(g.handle, P(x), P(stack), B, Fin, K, st),
                                    lib.chebgcn_contract_fwd(P(stack), P(W), P(bias), bias_kind, P(out), P(mask), B, M, Fin, K, Fout, 1, 0, relu, st)))
            t_f1 = timeit(lambda: lib.chebgcn_fused_layer_fwd(g.handle, P(x), P(W), P(bias), bias_kind, P(stack), P(out), P(mask), P(ws), nws, B, Fin, K, Fout, relu, st))
            t_f0 = timeit(lambda: lib.chebgcn_fused_layer_fwd(g.handle, P(x), P(W), P(bias), bias_kind, None, P(out), P(mask), P(ws), nws, B, Fin, K, Fout, relu, st))
            t_bs = timeit(lambda: (lib.chebgcn_contract_bwd_x_relu(P(dout), P(mask_r), P(W), P(gstack), B, M, Fin, K, Fout, st),
                                   lib.chebgcn_recurrence_bwd(g.handle, P(gstack), P(dx), B, Fin, K, st)))
            t_bf = timeit(lambda: lib.chebgcn_fused_layer_bwd_x(g.handle, P(dout), P(mask_r), P(W), P(dx), B, Fin, K, Fout, st))
            print('   us: forward separate %.1f, fused with stack %.1f, fused without %.1f;  backward-x separate %.1f, fused %.1f' % (
                t_sep, t_f1, t_f0, t_bs, t_bf), flush=True)
            import ctypes
            handle = ctypes.CDLL(_lib.LIB_PATH)
            if hasattr(handle, 'chebgcn_debug_stampsf'):                 # tools/fbuild.sh f64 "-DCG_EXPERIMENT=1 -DCG_X=64"
                for what, fn in (('forward with stack', lambda: lib.chebgcn_fused_layer_fwd(g.handle, P(x), P(W), P(bias), bias_kind, P(stack), P(out), P(mask), P(ws), nws, B, Fin, K, Fout, relu, st)),
                                 ('backward', lambda: lib.chebgcn_fused_layer_bwd_x(g.handle, P(dout), P(mask_r), P(W), P(dx), B, Fin, K, Fout, st))):
                    fn()
                    torch.cuda.synchronize()
                    buf = (ctypes.c_longlong * (16 * 64))()
                    assert handle.chebgcn_debug_stampsf(buf) == 0
                    t = np.array(buf, dtype=np.int64).reshape(16, 64)
                    t0 = t[t > 0].min()
                    ids = [i for i in range(64) if (t[:, i] > 0).any()]
                    print('   stamps, %s (0 start, 1 row, 2 W in LDS, 3 input, 4 image of T_0, 5.. end of step, 21.. after the gather of a step, 40 results out):' % what)
                    print('   id   ' + ' '.join('%6d' % i for i in ids))
                    for w in range(16):
                        if (t[w] > 0).any():
                            print('   w%-3d ' % w + ' '.join('%6d' % (t[w, i] - t0 if t[w, i] > 0 else -1) for i in ids))


if __name__ == '__main__':
    main()
