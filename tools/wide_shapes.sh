#!/bin/bash
# fp32 against split-bf16 ('bf16x3') contraction kernels at the wide shapes (M = 10466, batch 64): BASELINE configs 3 and 4, the
# F = 64 / 128 layers of the pooling ChebNet.   usage (GPU box): bash tools/wide_shapes.sh > gpurun_out/wide.txt
cd "${GRAFT_REPO_ROOT:-.}"
for shp in "64 64 25" "60 256 5" "128 128 5" "32 64 10" "64 64 10" "64 128 5"; do
  set -- $shp
  echo "== Fin=$1 Fout=$2 K=$3"
  timeout 300 python tools/kbench.py --B 64 --fin $1 --fout $2 --K $3 --iters 10 --kernels contract_fwd contract_fwd_bf16x3 contract_bwd_w contract_bwd_w_bf16x3 contract_bwd_x contract_bwd_x_bf16x3 2>&1 | grep "^contract"
done
