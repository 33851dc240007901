#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6b; rm -rf $out; mkdir -p $out
python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $out/t_round6.log 2>&1; echo "round6 rc=$?"; tail -3 $out/t_round6.log
python tools/pool6_probe.py > $out/pool6.txt 2>&1; echo "pool6 rc=$?"; head -30 $out/pool6.txt
python -m pytest tests -q -m gpu > $out/t_all.log 2>&1; echo "all rc=$?"; tail -15 $out/t_all.log
python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
l=json.loads([x for x in open('gpurun_out/r6b/bench_line.json') if x.startswith('{"metric"')][-1])
print(l['value'], l['ms_per_step'], l['ms_per_step_repeats']['all'], l['roofline'])
print('step_roofline', l['step_roofline']['frac'])
for k,v in l['kernels_by_symbol'].items(): print('  %-70s %.4f ms x%d  %.3f' % (k, v['avg_ms'], v['launches'], v['frac_hbm']))
print('pool6', l['pool6']['ms_per_step'], l['pool6']['step_roofline']['frac'])
print('refshape', {k:(v['eager']['ms_per_step'], v['hip_graph']['ms_per_step']) for k,v in l['refshape'].items()})
print('config5', {p:l['config5'][p]['ms_per_step'] for p in ('f32','bf16','bf16x3')}, 'config4.layer', {p:l['config4']['layer'][p]['ms_per_step'] for p in ('f32','bf16x3')})
print('cpu', l['cpu_baseline']['value'], l['cpu_baseline']['cores'])
PY
