#!/bin/bash
# Runs on the GPU box (gpurun): refreshes everything profiles/ holds for the current kernels.
#   bench line, rocprofv3 --kernel-trace --stats of the same bench command, per-kernel micro-bench at the
#   bench / north-star shapes and at BASELINE configs 4 and 5, PMC traffic per kernel.
out=$GRAFT_REPO_ROOT/gpurun_out/refresh
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench_line.json 2> $out/bench.err
python3 tools/kbench.py --json $out/kbench.json > $out/kbench.txt 2>&1
python3 tools/kbench.py --B 64 256 --fin 64 --K 25 --kernels recurrence_fwd recurrence_bwd --iters 5 > $out/kbench_config4.txt 2>&1
python3 tools/kbench.py --B 64 --fin 60 --fout 256 --K 5 --kernels contract_fwd contract_fwd_bf16 contract_fwd_bf16x3 --iters 10 > $out/kbench_config5.txt 2>&1
bash tools/pmc_traffic.sh refresh > $out/traffic.log 2>&1
cp gpurun_out/traffic_refresh/traffic_raw.json $out/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-windows 0 > $out/prof.log 2>&1
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \;
rm -rf $out/prof/*/*kernel_trace.csv
ls -la $out
