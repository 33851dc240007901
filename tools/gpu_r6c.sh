#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6c; rm -rf $out; mkdir -p $out
python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py -x -q -m gpu > $out/t_round6.log 2>&1; echo "round6+3 rc=$?"; tail -3 $out/t_round6.log
python tools/pool6_probe.py > $out/pool6.txt 2>&1; echo "pool6 rc=$?"; head -30 $out/pool6.txt
