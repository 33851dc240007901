python tools/kbench.py --B 64 128 256 --kernels recurrence_fwd recurrence_bwd --iters 20 2>&1 | grep recurrence
python bench.py --cpu-windows 0 | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(l['value'], l['ms_per_step'], {k:round(v['avg_ms'],4) for k,v in l['kernels'].items()})"
