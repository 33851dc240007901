#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6i; rm -rf $out; mkdir -p $out
python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "relabelled" > $out/t.log 2>&1; echo "relabelled rc=$?"; tail -3 $out/t.log
python -m pytest tests -q -m gpu > $out/t_all.log 2>&1; echo "all rc=$?"; tail -6 $out/t_all.log
