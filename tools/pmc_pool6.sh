#!/bin/bash
# usage: tools/pmc_pool6.sh ; HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, --kernel-trace only) of every libchebgcn
# kernel of the six-level pooling network's training step (tools/pool6_probe.py, batch 64), per kernel template and launch
out=$GRAFT_REPO_ROOT/gpurun_out/traffic_pool6
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  STEPS=3 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p$i -- python3 $GRAFT_REPO_ROOT/tools/pool6_probe.py > $out/p$i.log 2>&1
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
for k, d in sorted(acc.items()):
    if 'FETCH_SIZE' not in d or 'WRITE_SIZE' not in d: continue
    f, w = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']), sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
    res[k] = {'launches_seen': len(d['FETCH_SIZE']), 'FETCH_SIZE_KiB_mean': f, 'WRITE_SIZE_KiB_mean': w, 'bytes_mean': (2.0 * f + w) * 1024.0}
    print('%-110s n=%-4d %8.1f MB per launch (mean over its launches of different shapes)' % (k[:110], len(d['FETCH_SIZE']), res[k]['bytes_mean'] / 1e6))
json.dump(res, open(out + '/traffic_pool6.json', 'w'), indent=1)
PY
