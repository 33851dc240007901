#!/usr/bin/env python3
"""Where the fixed cost of dist.DataParallel goes (world 1 on RCCL): the bench step plain / wrapped / wrapped with the
collectives skipped / wrapped and captured as a HIP graph."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import bench
    from gcn_fmri_decoding_amd import models_gcn, ops
    from gcn_fmri_decoding_amd import dist as gdist
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    nodes = int(os.environ.get('NODES', '10000'))
    B = int(os.environ.get('BATCH', '64'))
    K = int(os.environ.get('KORDER', '5'))
    Ls, perm = bench.load_graph(nodes, 1, 0, 1, None)
    torch.manual_seed(0)
    net = models_gcn.cgcnn({'device': dev}, [Ls[0]] * 6, [32] * 6, [K] * 6, [1] * 6, [512, 256, 22], filter='chebyshev5', brelu='b2relu',
                           pool='mpool1', initial='he', channel=15, regularization=5e-4, dropout=0.5, batch_size=B,
                           learning_rate=0.001, decay_rate=0.9, momentum=0.9, verbose=False)
    S = 4 * B
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    data = torch.randn((S, nodes, 15), generator=g, device=dev)
    labels = torch.randint(0, 21, (S,), generator=g, device=dev)
    perm_dev = net.compose_perm(perm)
    order = torch.stack([torch.randperm(S, generator=g, device=dev)[:B].to(torch.int32) for _ in range(64)])

    def step(i):
        idx = order[i % 64]
        x = net.as_internal(ops.perm_data(data, perm_dev, idx))
        return net.train_step(x, labels[idx.long()])

    def timed(tag, n=60):
        for i in range(8):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        print('%-40s %.4f ms/step' % (tag, ms), flush=True)
        return ms

    timed('plain')
    timed('plain again')
    net.enable_step_graph(True)
    timed('plain, captured')
    net.enable_step_graph(False)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % bench.free_port(), rank=0, world_size=1, device_id=dev)
    dp = gdist.DataParallel(net)
    timed('dp world 1')
    real = dp._reduce
    dp._reduce = lambda a, b: None
    timed('dp world 1, collectives skipped')
    dp._reduce = real
    timed('dp world 1 again')
    try:
        net.enable_step_graph(True)
        timed('dp world 1, captured')
    except Exception as e:                      # noqa: BLE001
        print('captured dp step failed:', repr(e)[:300])
    net.enable_step_graph(False)
    dp.remove()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
