cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for lib in old new; do
if [ $lib = old ]; then export CHEBGCN_LIB=$GRAFT_REPO_ROOT/build_x/libchebgcn_old.so; else unset CHEBGCN_LIB; fi
python bench.py --cpu-windows 0 --kernel-legs 0 --instrumented-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step_repeats']['all'])"
done; done
