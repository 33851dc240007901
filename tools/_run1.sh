echo "=== production"; python tools/kbench.py --B 64 256 --kernels recurrence_fwd recurrence_bwd --iters 10 2>&1 | grep recurrence
echo "=== x1024 (setprio)"; CHEBGCN_LIB=$PWD/build_x/libchebgcn_x1024.so python tools/kbench.py --B 64 256 --kernels recurrence_fwd recurrence_bwd --iters 10 2>&1 | grep recurrence
echo "=== parity"; python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for x in 1088 3136; do
  echo "=== stamps x$x"; CHEBGCN_LIB=$PWD/build_x/libchebgcn_x$x.so python tools/kbench.py --B 256 --kernels recurrence_fwd recurrence_bwd --stamps --iters 5 2>&1 | cut -c1-200 | grep -E "id  |w0 |w4 |w8 |recurr"
done
