#!/usr/bin/env python3
"""Time the REFERENCE's own CPU implementation of the hot path, in the build container.

Test / measurement infrastructure (see oracle/__init__.py).  Runs only where ``/root/reference``
exists (it never travels to the GPU box); what it writes -- ``profiles/r06_reference_cpu.json`` --
is plain numbers that ``bench.py`` quotes, labelled static, as ``cpu_baseline.reference``.

    python oracle/time_reference.py [--ref /root/reference] [--out profiles/r06_reference_cpu.json]

What is timed ("B1"): TensorFlow is not installed here and the reference pins
no version, so its TF graph cannot run; the reference ships a second, NumPy/SciPy implementation of
the same filter -- ``lib_new/graph.py:155-172`` ``chebyshev(L, X, K)`` (the body of
``cgcnn.chebyshev2``, ``lib_new/models_gcn.py:558-585``) -- which IS importable.  One layer forward =

    L~  = graph.rescale_L(L, lmax=2)                       lib_new/graph.py:146-152   (imported, once, untimed)
    Xt  = graph.chebyshev(L~, x0[M, Fin*N], K)             lib_new/graph.py:155-172   (imported, timed)
    x   = Xt.reshape(K, M, Fin, N).transpose(3, 1, 2, 0)   lib_new/models_gcn.py:611-613 (the ops of chebyshev5,
          .reshape(N*M, Fin*K)                                                          in NumPy, timed)
    y   = np.matmul(x, W[Fin*K, Fout])                     lib_new/models_gcn.py:616    (timed)

Forward only: no autodiff exists on this path.  SciPy's CSR SpMM runs on ONE thread; the transpose copy is
NumPy (one thread); ``np.matmul`` uses the BLAS thread pool.  Shapes: BASELINE.json configs[1] (layer 1:
Fin = 15, layers 2-6: Fin = 32; K = 5, Fout = 32), configs[3] (K = 25, 64 -> 64), configs[4] (60 -> 256, K = 5),
batch 64 each, on the benchmark graph (N = 10000 -> M = 10466, built by the reference's functions: the product's
``graph.synthetic_graph`` is bit-identical to them, tests/test_host_golden.py).  1 warm-up + median of 5.
"""
import argparse
import json
import os
import platform
import sys
import time
import warnings

import numpy as np

warnings.filterwarnings('ignore')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r06_reference_cpu.json'))
    ap.add_argument('--repeats', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64)
    args = ap.parse_args()
    if not os.path.isdir(args.ref):
        raise SystemExit('%s not found: this script runs in the build container only' % args.ref)
    sys.path.insert(0, args.ref)
    sys.path.insert(0, ROOT)
    from lib_new import graph as ref_graph          # the reference's own module
    import scipy.sparse as sp
    import bench
    try:
        import threadpoolctl
        blas_threads = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1

    Ls, _ = bench.load_graph(10000, 1, 0, 1, None)
    L = sp.csr_matrix(Ls[0]).astype(np.float32)
    M = L.shape[0]
    Lt = ref_graph.rescale_L(L.copy(), lmax=2)       # graph.py:146-152
    Lt = sp.csr_matrix(Lt)
    N = args.batch
    rs = np.random.RandomState(0)

    def layer(Fin, K, Fout):
        x = rs.randn(N, M, Fin).astype(np.float32)
        W = (rs.randn(Fin * K, Fout) * np.sqrt(2.0 / (Fin * K))).astype(np.float32)
        t = {'recurrence_s': [], 'transpose_s': [], 'matmul_s': [], 'layer_s': []}
        for it in range(args.repeats + 1):
            t0 = time.perf_counter()
            x0 = np.ascontiguousarray(np.transpose(x, (1, 2, 0))).reshape(M, Fin * N)       # models_gcn.py:598-599
            Xt = ref_graph.chebyshev(Lt, x0, K)                                              # graph.py:155-172
            t1 = time.perf_counter()
            xs = np.ascontiguousarray(Xt.reshape(K, M, Fin, N).transpose(3, 1, 2, 0)).reshape(N * M, Fin * K)   # :611-613
            t2 = time.perf_counter()
            y = np.matmul(xs, W).reshape(N, M, Fout)                                         # :616-617
            t3 = time.perf_counter()
            if it:                                   # (the first pass is the warm-up: page faults of the stack)
                t['recurrence_s'].append(t1 - t0)
                t['transpose_s'].append(t2 - t1)
                t['matmul_s'].append(t3 - t2)
                t['layer_s'].append(t3 - t0)
            del x0, Xt, xs
        out = {k: float(np.median(v)) for k, v in t.items()}
        out['layer_s_all'] = [float(v) for v in t['layer_s']]
        out['shape'] = {'N': N, 'M': M, 'Fin': Fin, 'K': K, 'Fout': Fout}
        out['windows_per_s_this_layer_fwd'] = N / out['layer_s']
        out['checksum'] = float(np.abs(y).mean())
        print('Fin=%d K=%d Fout=%d: recurrence %.2f s, transpose %.2f s, matmul %.2f s, layer %.2f s' % (
            Fin, K, Fout, out['recurrence_s'], out['transpose_s'], out['matmul_s'], out['layer_s']), flush=True)
        return out

    res = {
        'what': "the reference's own NumPy/SciPy implementation of the Chebyshev filter, imported from /root/reference and "
                'timed in the build container: lib_new/graph.py:155-172 chebyshev() + the reshape/transpose of '
                'lib_new/models_gcn.py:611-613 + np.matmul (:616); forward only (no autodiff on this path); 1 warm-up + median of %d'
                % args.repeats,
        'host': {'cpus': os.cpu_count(), 'spmm_threads': 1, 'blas_threads': int(blas_threads), 'machine': platform.machine(),
                 'numpy': np.__version__, 'scipy': __import__('scipy').__version__,
                 'note': 'build container (8 vCPU), NOT the GPU box: the reference cannot travel there'},
        'graph': {'N': 10000, 'M': int(M), 'nnz': int(Lt.nnz)},
        'configs1_layer1': layer(15, 5, 32),
        'configs1_layers2to6': layer(32, 5, 32),
        'configs3': layer(64, 25, 64),
        'configs4': layer(60, 5, 256),
    }
    fwd = res['configs1_layer1']['layer_s'] + 5 * res['configs1_layers2to6']['layer_s']
    res['configs1_network_forward'] = {
        'conv_layers_s': fwd, 'windows_per_s': N / fwd,
        'what': 'six conv layers of BASELINE configs[1] forward (1 x Fin=15 + 5 x Fin=32), batch %d; no bias/ReLU, head, loss, backward '
                'or optimizer in it -- an upper bound of what this code path could deliver for the full step' % N}
    with open(args.out, 'w') as f:
        json.dump(res, f, indent=1)
    print('configs[1] forward, six conv layers: %.1f s -> %.2f windows/s' % (fwd, N / fwd))
    print('wrote', args.out)


if __name__ == '__main__':
    main()
