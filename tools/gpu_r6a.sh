#!/bin/bash
# round 6, first GPU pass: the new kernels' tests first, the pooling network's probe, then the whole GPU suite
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6a; rm -rf $out; mkdir -p $out
python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $out/t_round6.log 2>&1; echo "round6 rc=$?"; tail -5 $out/t_round6.log
python -m pytest tests/test_gpu_recurrence_shapes.py -x -q -m gpu -k "ordered" > $out/t_shapes.log 2>&1; echo "shapes rc=$?"; tail -5 $out/t_shapes.log
python -m pytest tests/test_gpu_bench_shapes.py -x -q -m gpu -k "pooling or brelu_pool" > $out/t_pool.log 2>&1; echo "pool rc=$?"; tail -5 $out/t_pool.log
python tools/pool6_probe.py > $out/pool6.txt 2>&1; echo "pool6 rc=$?"; tail -40 $out/pool6.txt
python -m pytest tests -q -m gpu -x > $out/t_all.log 2>&1; echo "all rc=$?"; tail -15 $out/t_all.log
