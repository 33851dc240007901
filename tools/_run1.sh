python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
python tools/kbench.py --B 64 256 --kernels contract_fwd --iters 20 2>&1 | grep contract
