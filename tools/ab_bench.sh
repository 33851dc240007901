#!/bin/bash
# A/B of bench.py variants inside ONE gpurun call (box-to-box noise is +-5..10 %): alternates the
# argument sets given (one quoted string each) N times.  usage: tools/ab_bench.sh N "args A" "args B" ...
n=$1; shift
for i in $(seq 1 $n); do
  for a in "$@"; do
    python bench.py --cpu-windows 0 --kernel-legs 0 $a 2>/dev/null | python -c "
import json, sys
l = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s %9.1f windows/s  %.3f ms/step' % (sys.argv[1], l['value'], l['ms_per_step']))" "$a"
  done
done
