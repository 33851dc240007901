cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round3.py tests/test_gpu_dispatch.py -q -m gpu -x 2>&1 | tail -3
for n in 1000 360 400 2000; do python tools/refshape.py --nodes $n | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($n, d['eager']['ms_per_step'], d['hip_graph']['ms_per_step'])"; done
python tools/kbench.py --nodes 1000 --B 128 --K 10 --iters 50 --kernels contract_fwd contract_fwd_dx contract_bwd_w contract_bwd_x 2>&1 | tail -5
