#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6f; rm -rf $out; mkdir -p $out
python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "reindex or fit_tracks" > $out/t.log 2>&1; echo "tests rc=$?"; tail -3 $out/t.log
python -m pytest tests/test_gpu_dispatch.py -x -q -m gpu -k "config2_network or invisible" > $out/t2.log 2>&1; echo "tests2 rc=$?"; tail -3 $out/t2.log
python bench.py --kernel-legs 0 --cpu-windows 0 > $out/bench_a.json 2>/dev/null
CHEBGCN_FC_BWD_MAX_INNER=1048576 python bench.py --kernel-legs 0 --cpu-windows 0 > $out/bench_b.json 2>/dev/null
python bench.py --kernel-legs 0 --cpu-windows 0 --overlap-bwd-w 0 > $out/bench_c.json 2>/dev/null
python - <<'PY'
import json
for t in 'abc':
    l=json.loads([x for x in open('gpurun_out/r6f/bench_%s.json'%t) if x.startswith('{"metric"')][-1])
    print(t, round(l['value']), [round(v,4) for v in l['ms_per_step_repeats']['all']], round(l['roofline']['frac'],4), round(l['step_roofline']['frac'],4))
PY
