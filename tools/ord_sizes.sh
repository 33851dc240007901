#!/bin/bash
# The recurrence over graph sizes, ordered kernels against the round-3 kernels on the same box (batch 256, K = 5, Fin = 32).
#   usage (GPU box): bash tools/ord_sizes.sh [sizes...] > gpurun_out/ord_sizes.txt
cd "${GRAFT_REPO_ROOT:-.}"
sizes=${@:-"2600 6000 8000 10000 10242 13000 19000"}
for n in $sizes; do
  for order in length reference; do
    echo "== N=$n order=$order"
    timeout 600 python tools/kbench.py --nodes $n --order $order --B 64 256 --iters 20 --kernels recurrence_fwd_inplace recurrence_bwd 2>&1 | grep -v Warning
  done
done
