#!/bin/bash
# Runs on the GPU box (gpurun): refreshes everything profiles/ holds for the current kernels.
#   bench line, rocprofv3 --kernel-trace --stats of the same bench command (and of the north-star launches and the
#   reference's own training shape), per-kernel micro-bench at the
#   bench / north-star shapes and at BASELINE configs 4 and 5 (with their own rocprofv3 kernel stats), PMC
#   traffic per kernel, matrix-core counters of the contraction kernels.
out=$GRAFT_REPO_ROOT/gpurun_out/refresh
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench_line.json 2> $out/bench.err
python3 tools/kbench.py --json $out/kbench.json > $out/kbench.txt 2>&1
python3 tools/kbench.py --planes 2 --B 64 256 --kernels recurrence_fwd recurrence_fwd_inplace recurrence_bwd > $out/kbench_two_planes.txt 2>&1
# the kernels of rounds 1-3 (the caller's vertex numbering: four planes at batch 256, two at batch 64) beside the ordered ones
python3 tools/kbench.py --order reference --B 64 256 --kernels recurrence_fwd recurrence_fwd_inplace recurrence_bwd > $out/kbench_reference_order.txt 2>&1
python3 tools/kbench.py --B 64 256 --fin 64 --K 25 --kernels recurrence_fwd_inplace recurrence_bwd --iters 30 > $out/kbench_config4.txt 2>&1
python3 tools/kbench.py --B 64 --fin 60 --fout 256 --K 5 --kernels contract_fwd contract_fwd_bf16 contract_fwd_bf16x3 contract_bwd_w contract_bwd_w_bf16 contract_bwd_w_bf16x3 contract_bwd_x contract_bwd_x_bf16 contract_bwd_x_bf16x3 brelu_pool_bwd_mask relu_grad_bf16 contract_bwd_w_bf16_dy16 contract_bwd_x_bf16_dy16 --iters 10 > $out/kbench_config5.txt 2>&1
bash tools/pmc_traffic.sh refresh > $out/traffic.log 2>&1
cp gpurun_out/traffic_refresh/traffic_raw.json $out/ 2>/dev/null
bash tools/pmc_mfma.sh refresh > $out/mfma.txt 2>&1
# the ordered recurrence kernels: SQ instruction / wait / LDS counters at the north-star shape and at the batch of the bench
# step, and the in-kernel phase stamps of one plane group (build_x/libchebgcn_x64.so: tools/xbuild.sh 64)
# (the experiment libraries are built HERE, not on the GPU box: a stale one lacks the symbols the ctypes table asks for -- skip it
#  rather than overwrite the stamp files with a traceback)
lib_ok() { [ -f "$1" ] && CHEBGCN_LIB=$1 python3 -c "from gcn_fmri_decoding_amd import _lib; _lib.lib()" 2>/dev/null; }
bash tools/pmc_sq.sh refresh_ord --B 256 --iters 3 --kernels recurrence_fwd_inplace recurrence_bwd > $out/recurrence_ord_sq_counters.txt 2>&1
if lib_ok $GRAFT_REPO_ROOT/build_x/libchebgcn_x64.so; then
  CHEBGCN_LIB=$GRAFT_REPO_ROOT/build_x/libchebgcn_x64.so python3 tools/kbench.py --B 256 --kernels recurrence_fwd_inplace recurrence_bwd --stamps > $out/stampso.txt 2>&1
fi
# the bf16 contraction kernels of config 5 (build_x/libchebgcn_b64.so: tools/bbuild.sh b64 "-DCG_EXPERIMENT=1 -DCG_X=64")
if lib_ok $GRAFT_REPO_ROOT/build_x/libchebgcn_b64.so; then
  CHEBGCN_LIB=$GRAFT_REPO_ROOT/build_x/libchebgcn_b64.so python3 tools/kbench.py --stamps --B 64 --fin 60 --fout 256 --K 5 --iters 10 --kernels contract_fwd_bf16 contract_bwd_x_bf16_dy16 > $out/stampsb.txt 2>&1
fi
python3 tools/config5_probe.py > $out/config5_layer.txt 2>&1
SHAPE="64 25 64" STEPS=5 python3 tools/config5_probe.py > $out/config4_layer.txt 2>&1
# round 5: the ordered recurrence over graph sizes beside the kernels of rounds 1-3 on the same box; fp32 against split-bf16
# contraction kernels at the wide shapes; the pooling ChebNet of SURVEY 8(f)4
bash tools/ord_sizes.sh 2600 6000 8000 10000 10242 13000 19000 > $out/ord_sizes.txt 2>&1
bash tools/wide_shapes.sh > $out/wide_f32_vs_bf16x3.txt 2>&1
python3 tools/pool6_probe.py > $out/pool6.txt 2>&1
python3 tools/fused_check.py > $out/fused_small_check.txt 2>&1
# ... and its phase stamps (build_x/libchebgcn_f64.so: tools/fbuild.sh f64 "-DCG_EXPERIMENT=1 -DCG_X=64")
if lib_ok $GRAFT_REPO_ROOT/build_x/libchebgcn_f64.so; then
  CHEBGCN_LIB=$GRAFT_REPO_ROOT/build_x/libchebgcn_f64.so python3 tools/fused_check.py 2>&1 | grep -A30 "N=360 M=376 B=128 Fin=32" | grep -B1 -A13 "stamps" > $out/stampsf.txt
fi
cp gpurun_out/pmcmfma_refresh/available.txt $out/mfma_counters_available.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-windows 0 --kernel-legs 0 > $out/prof.log 2>&1
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \;
# the same command with every kernel ALONE on the device (no second stream for contract_bwd_w): the AverageNs of a kernel symbol
# here is what `roofline` / `kernels_by_symbol` of the bench line (HIP events, instrumented pass) must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $out/profs -o benchs -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-windows 0 --kernel-legs 0 --overlap-bwd-w 0 > $out/profs.log 2>&1
find $out/profs -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats_serial.csv \;
grep '^{"metric"' $out/profs.log | tail -1 > $out/bench_line_serial.json
# configs[3] (K = 25, Fin = Fout = 64, batch 64): one program per recurrence entry, 30 launches each (the three warm-up launches of
# kbench are in the average: with 5 launches, as in round 2, they skewed it by 10-15 %)
for kern in recurrence_fwd_inplace recurrence_bwd; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof4_$kern -o c4 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --B 64 --fin 64 --fout 64 --K 25 --iters 30 --kernels $kern > $out/events_under_rocprof_config4_$kern.txt 2>&1
  find $out/prof4_$kern -name "*kernel_stats.csv" -exec cp {} $out/config4_${kern}_kernel_stats.csv \;
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof4 -o c4 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --B 64 --fin 64 --fout 64 --K 25 --iters 10 --kernels contract_fwd contract_bwd_w contract_bwd_x > $out/prof4.log 2>&1
find $out/prof4 -name "*kernel_stats.csv" -exec cp {} $out/config4_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof5 -o c5 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --B 64 --fin 60 --fout 256 --K 5 --iters 5 --kernels recurrence_fwd_inplace contract_fwd contract_fwd_bf16 contract_fwd_bf16x3 contract_bwd_w contract_bwd_w_bf16 contract_bwd_w_bf16x3 contract_bwd_x contract_bwd_x_bf16 contract_bwd_x_bf16x3 relu_grad_bf16 contract_bwd_w_bf16_dy16 contract_bwd_x_bf16_dy16 > $out/prof5.log 2>&1
find $out/prof5 -name "*kernel_stats.csv" -exec cp {} $out/config5_kernel_stats.csv \;
# the north-star shape (K = 5, Fin = 32, batch 256): one program per entry, so that the in-place forward, the forward with the
# copy of x and the adjoint each have their own average (the first two are the same kernel)
for kern in recurrence_fwd_inplace recurrence_fwd recurrence_bwd; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/profn_$kern -o ns -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --B 256 --iters 100 --kernels $kern > $out/events_under_rocprof_northstar_$kern.txt 2>&1
  find $out/profn_$kern -name "*kernel_stats.csv" -exec cp {} $out/northstar_${kern}_kernel_stats.csv \;
done
# round 5: the ordered kernels at N = 6000 (four planes, NG = 3) and N = 13000 (two planes), batch 256; the pooling ChebNet step
for n in 6000 13000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/profo_$n -o o -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --nodes $n --B 256 --iters 50 --kernels recurrence_fwd_inplace recurrence_bwd > $out/events_under_rocprof_ordered_n$n.txt 2>&1
  find $out/profo_$n -name "*kernel_stats.csv" -exec cp {} $out/ordered_n${n}_kernel_stats.csv \;
done
# round 6: the level-0 graph of the six-level pooling network (12672 vertices, 10000 active) in length order: the ordered kernel
# with NQ = NG + 1 + cheb_ord_tail_kernel, at the shapes of that network's first two layers' launches
python3 $GRAFT_REPO_ROOT/tools/kbench.py --levels 6 --B 64 --fin 32 --fout 32 --K 10 --iters 20 --kernels recurrence_fwd_inplace recurrence_bwd recurrence_fwd_t > $out/kbench_pool6_level0.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/profl0 -o l0 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --levels 6 --B 64 --fin 32 --fout 32 --K 10 --iters 30 --kernels recurrence_fwd_inplace recurrence_fwd_t > $out/events_under_rocprof_pool6_level0.txt 2>&1
find $out/profl0 -name "*kernel_stats.csv" -exec cp {} $out/pool6_level0_recurrence_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $out/profp -o p -- python3 $GRAFT_REPO_ROOT/tools/pool6_probe.py > $out/profp.log 2>&1
find $out/profp -name "*kernel_stats.csv" -exec cp {} $out/pool6_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $out/profr -o rs -- python3 $GRAFT_REPO_ROOT/tools/refshape.py --nodes 360 > $out/profr.log 2>&1
find $out/profr -name "*kernel_stats.csv" -exec cp {} $out/refshape_n360_kernel_stats.csv \;
grep "^{\"shape\"" $out/profr.log | tail -1 > $out/refshape_n360_line.json
# what the memory system and the fp32 matrix pipe give with nothing else going on (EXPERIMENTS.md 3b)
for b in 8 16 32 64; do $GRAFT_REPO_ROOT/tools/probes/hbm_stream_probe $b; done > $out/hbm_stream_probe.txt 2>&1
$GRAFT_REPO_ROOT/tools/probes/mfma_f32_probe > $out/mfma_f32_probe.txt 2>&1
rm -rf $out/prof $out/profs $out/profl0 $out/prof4 $out/prof4_* $out/prof5 $out/profn_* $out/profr $out/profo_* $out/profp
cp gpurun_out/parity_measured.jsonl $out/ 2>/dev/null
ls -la $out
