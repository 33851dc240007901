#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6d; rm -rf $out; mkdir -p $out
for o in length bank; do
  echo "== order $o"; python tools/kbench.py --order $o --B 64 256 --iters 30 --kernels recurrence_fwd_inplace recurrence_bwd recurrence_fwd_t 2>&1 | grep -v Warning
done | tee $out/kbench_order.txt
for o in length bank; do
  echo "== N=6000 order $o"; python tools/kbench.py --nodes 6000 --order $o --B 256 --iters 20 --kernels recurrence_fwd_inplace recurrence_bwd 2>&1 | grep -v Warning
done | tee -a $out/kbench_order.txt
CHEBGCN_BANK_ORDER=0 python bench.py --kernel-legs 0 --cpu-windows 0 > $out/bench_length.json 2>/dev/null
python bench.py --kernel-legs 0 --cpu-windows 0 > $out/bench_bank.json 2>/dev/null
python - <<'PY'
import json
for t in ('length','bank'):
    l=json.loads([x for x in open('gpurun_out/r6d/bench_%s.json'%t) if x.startswith('{"metric"')][-1])
    print(t, l['value'], l['ms_per_step_repeats']['all'], l['roofline']['frac'], l['step_roofline']['frac'])
PY
