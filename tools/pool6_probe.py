#!/usr/bin/env python3
"""SURVEY 8(f)4 alone: bench.pool6_leg (the monolith's pooling ChebNet at full size) without the rest of bench.py."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    out = bench.pool6_leg(torch.device('cuda:0'), int(os.environ.get('STEPS', 10)), 3, batch=int(os.environ.get('BATCH', 64)))
    print('%.3f ms/step  %.0f windows/s  step_roofline %.3f  kernel sum %.3f ms  precision %s' % (
        out['ms_per_step'], out['windows_per_s'], out['step_roofline']['frac'], out['kernel_ms_sum'], out['precision']))
    for k, v in sorted(out['kernels'].items(), key=lambda kv: -kv[1]['ms_per_step']):
        print('  %8.4f ms  x%-4.1f %6.0f GB/s %5.1f%%  %6.1f TF   %s' % (v['ms_per_step'], v['launches_per_step'], v['GBps'],
                                                                    100 * v['frac_hbm'], v['TFLOPs'], k))
    if os.environ.get('JSON'):
        json.dump(out, open(os.environ['JSON'], 'w'), indent=1)


if __name__ == '__main__':
    main()
