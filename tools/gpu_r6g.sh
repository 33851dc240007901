#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6g; rm -rf $out; mkdir -p $out
python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "relabelled" > $out/t.log 2>&1; echo "tests rc=$?"; tail -5 $out/t.log
# advisor (round 5): pick_ell on graphs beyond 10752 vertices in the caller's order -- four planes for every launch size?  small launches:
for pl in 0 2; do for B in 1 2 4 8; do
  echo "== planes $pl B $B"; python tools/kbench.py --levels 6 --order reference --planes $pl --B $B --fin 32 --K 10 --iters 30 --kernels recurrence_fwd_inplace recurrence_bwd 2>&1 | grep "recurrence_"
done; done | tee $out/pick_ell_small_launches.txt
python $GRAFT_REPO_ROOT/tools/kbench.py --levels 6 --B 64 --fin 32 --fout 32 --K 10 --iters 20 --kernels recurrence_fwd_inplace recurrence_bwd recurrence_fwd_t 2>&1 | grep -v amdgpu | tee $out/kbench_pool6_level0.txt
