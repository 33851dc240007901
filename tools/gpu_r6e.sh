#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6e; rm -rf $out; mkdir -p $out
python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $out/t_round6.log 2>&1; echo "round6 rc=$?"; tail -3 $out/t_round6.log
python tools/pool6_probe.py > $out/pool6.txt 2>&1; echo "pool6 rc=$?"; head -30 $out/pool6.txt
python tools/config5_probe.py > $out/config5.txt 2>&1; tail -25 $out/config5.txt
python -m pytest tests -q -m gpu > $out/t_all.log 2>&1; echo "all rc=$?"; tail -8 $out/t_all.log
grep fit_tracks gpurun_out/parity_measured.jsonl | tail -2
