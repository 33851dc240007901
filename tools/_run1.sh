python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16" 2>&1 | tail -3
echo "=== config5 B=64"; python tools/kbench.py --B 64 --fin 60 --fout 256 --K 5 --kernels contract_fwd contract_fwd_bf16 contract_fwd_bf16x3 --iters 10 2>&1 | grep contract
