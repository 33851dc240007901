#!/usr/bin/env python3
"""Copies what tools/refresh_profiles.sh left under gpurun_out/refresh into profiles/ (tracked),
named per round, and derives profiles/traffic.json (HBM bytes per launch and kernel, read by bench.py).

    python tools/collect_profiles.py [round-tag, default r01]
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'refresh')
DST = os.path.join(ROOT, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'

for src, dst in [('bench_line.json', '%s_bench_line.json'), ('bench_kernel_stats.csv', '%s_bench_kernel_stats.csv'),
                 ('bench_kernel_stats_serial.csv', '%s_bench_kernel_stats_serial.csv'), ('bench_line_serial.json', '%s_bench_line_serial.json'),
                 ('kbench.txt', '%s_kbench.txt'), ('kbench.json', '%s_kbench.json'),
                 ('kbench_config4.txt', '%s_kbench_config4_K25_F64.txt'), ('kbench_config5.txt', '%s_kbench_config5_bf16.txt'),
                 ('hbm_stream_probe.txt', '%s_hbm_stream_probe.txt'), ('mfma_f32_probe.txt', '%s_mfma_f32_probe.txt'),
                 ('traffic_raw.json', '%s_traffic_raw.json'), ('kbench_two_planes.txt', '%s_kbench_two_plane_recurrence.txt'), ('kbench_reference_order.txt', '%s_kbench_reference_order_recurrence.txt'),
                 ('stampso.txt', '%s_recurrence_ord_phase_stamps.txt'), ('stampsb.txt', '%s_bf16_phase_stamps.txt'), ('stampsf.txt', '%s_fused_small_phase_stamps.txt'), ('config5_layer.txt', '%s_config5_layer.txt'), ('recurrence_ord_sq_counters.txt', '%s_recurrence_ord_sq_counters.txt'),
                 ('fused_small_check.txt', '%s_fused_small_check.txt'), ('parity_measured.jsonl', '%s_parity_measured.jsonl'),
                 ('mfma.txt', '%s_contraction_mfma_counters.txt'), ('stamps4.txt', '%s_recurrence4_phase_stamps.txt'), ('mfma_counters_available.txt', '%s_mfma_counters_available.txt'),
                 ('config4_kernel_stats.csv', '%s_config4_kernel_stats.csv'), ('config5_kernel_stats.csv', '%s_config5_kernel_stats.csv'),
                 ('northstar_recurrence_fwd_inplace_kernel_stats.csv', '%s_northstar_fwd_inplace_kernel_stats.csv'),
                 ('northstar_recurrence_fwd_kernel_stats.csv', '%s_northstar_fwd_copy_x_kernel_stats.csv'),
                 ('northstar_recurrence_bwd_kernel_stats.csv', '%s_northstar_bwd_kernel_stats.csv'),
                 ('config4_recurrence_fwd_inplace_kernel_stats.csv', '%s_config4_recurrence_fwd_kernel_stats.csv'),
                 ('config4_recurrence_bwd_kernel_stats.csv', '%s_config4_recurrence_bwd_kernel_stats.csv'), ('refshape_n360_kernel_stats.csv', '%s_refshape_n360_kernel_stats.csv'),
                 ('refshape_n360_line.json', '%s_refshape_n360_line.json'),
                 ('config4_layer.txt', '%s_config4_layer.txt'), ('ord_sizes.txt', '%s_ordered_recurrence_sizes.txt'),
                 ('wide_f32_vs_bf16x3.txt', '%s_wide_f32_vs_bf16x3.txt'), ('pool6.txt', '%s_pool6.txt'),
                 ('pool6_kernel_stats.csv', '%s_pool6_kernel_stats.csv'), ('kbench_pool6_level0.txt', '%s_kbench_pool6_level0.txt'),
                 ('pool6_level0_recurrence_kernel_stats.csv', '%s_pool6_level0_recurrence_kernel_stats.csv'),
                 ('events_under_rocprof_pool6_level0.txt', '%s_events_under_rocprof_pool6_level0.txt'),
                 ('ordered_n6000_kernel_stats.csv', '%s_ordered_n6000_kernel_stats.csv'),
                 ('ordered_n13000_kernel_stats.csv', '%s_ordered_n13000_kernel_stats.csv'),
                 ('events_under_rocprof_ordered_n6000.txt', '%s_events_under_rocprof_ordered_n6000.txt'),
                 ('events_under_rocprof_ordered_n13000.txt', '%s_events_under_rocprof_ordered_n13000.txt')] + [
        ('events_under_rocprof_%s_%s.txt' % (c, k), '%%s_events_under_rocprof_%s_%s.txt' % (c, k))
        for c in ('northstar', 'config4') for k in ('recurrence_fwd_inplace', 'recurrence_fwd', 'recurrence_bwd')]:
    p = os.path.join(SRC, src)
    if os.path.exists(p):
        if src.endswith('.txt') or src == 'bench_line.json':
            lines = [l for l in open(p) if 'amdgpu.ids' not in l and l.strip() not in ('1', '6 6 6')
                     and not re.match(r'[WEI]\d{8} ', l)]            # rocprofv3's own log lines
            if src == 'bench_line.json':
                lines = [l for l in lines if l.startswith('{"metric"')]
            open(os.path.join(DST, dst % tag), 'w').writelines(lines)
        else:
            shutil.copy(p, os.path.join(DST, dst % tag))

raw = json.load(open(os.path.join(SRC, 'traffic_raw.json')))
# rocprofv3's kernel name (substring) -> the name chebgcn_last_dispatch() reports for that template (what bench.py's
# `kernels_by_symbol` / `roofline.kernel` are keyed by); kbench launches each at the shape of the step (B=64, Fin=Fout=32, K=5)
names = {'cheb_ord_kernel<4, 10240, 6, 5, 512, false>': 'cheb_ord_kernel<10240,6,5,512,false>',
         'cheb_ord_kernel<4, 10240, 6, 5, 512, true>': 'cheb_ord_kernel<10240,6,5,512,true>',
         'contract_fwd_ring_kernel<true, false>': 'contract_fwd_ring_kernel', 'contract_fwd_ring_kernel<false, false>': 'contract_fwd_ring_kernel<pool>',
         'contract_fwd_ring_kernel<true, true>': 'contract_fwd_ring_kernel<gated>',
         'contract_bwd_w_kernel<5, true>': 'contract_bwd_w_kernel<5,true> + reduce_partials_wide',
         'contract_bwd_w_kernel<5, false>': 'contract_bwd_w_kernel<5,false> + reduce_partials_wide',
         'contract_bwd_x_lds_kernel<true>': 'contract_bwd_x_lds_kernel<true>', 'contract_bwd_x_lds_kernel<false>': 'contract_bwd_x_lds_kernel<false>',
         'bias_grad_relu_kernel<2, 4, false, true>': 'bias_grad_relu_kernel<CHEBGCN_BIAS_VERTEX,4>',
         'bias_grad_relu_kernel<2, 4, false, false>': 'bias_grad_sum_kernel<CHEBGCN_BIAS_VERTEX,4>', 'reduce_partials_wide': 'reduce_partials_wide'}
out = {'_note': 'HBM bytes per launch at the bench shape (B=64, Fin=Fout=32, K=5, M=10466), rocprofv3 --pmc FETCH_SIZE and '
                'WRITE_SIZE in separate passes with --kernel-trace only (tools/pmc_traffic.sh); bytes = (2*FETCH_SIZE + '
                'WRITE_SIZE) KiB -- FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; '
                'raw counters in %s_traffic_raw.json.  by_kernel: keyed by the template name chebgcn_last_dispatch() reports; a '
                'kernel that kbench launches on several operands (the forward recurrence on x and on dy, the forward contraction '
                'with and without bias) has the mean over those launches' % tag,
       'by_kernel': {}, 'by_rocprof_name': {}}
for k, v in raw.items():
    if 'FETCH_SIZE' not in v or 'WRITE_SIZE' not in v:
        continue
    nbytes = (2.0 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0
    out['by_rocprof_name'][k] = nbytes
    for pat, name in names.items():
        if pat in k and name not in out['by_kernel']:
            out['by_kernel'][name] = nbytes
json.dump(out, open(os.path.join(DST, 'traffic.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
