#!/bin/bash
# usage: tools/pmc_traffic.sh <tag> ; HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes,
# --kernel-trace only) of every libchebgcn kernel at the bench shape (B=64, Fin=Fout=32, K=5, M=10466)
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/traffic_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p$i -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --B 64 --iters 3 > $out/p$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + '/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'chebgcn' not in k: continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, d in acc.items():
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    print(k[:90], {c: round(x, 1) for c, x in res[k].items()})
json.dump(res, open(out + '/traffic_raw.json', 'w'), indent=1)
PY
