cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
for g in 1 0 1 0; do CHEBGCN_GATE_LINKS=0 CHEBGCN_RING_FLAT=$g python bench.py --cpu-windows 0 --kernel-legs 0 --instrumented-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flat=$g (links off)', d['value'], d['ms_per_step_repeats']['all'])"; done
python bench.py --cpu-windows 0 --kernel-legs 0 --instrumented-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step_repeats']['all'])"
