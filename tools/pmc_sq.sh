#!/bin/bash
# usage: tools/pmc_sq.sh <tag> <kbench args...> ; instruction-mix / stall counters of the kernels kbench runs
# (each rocprofv3 --pmc pass in its own run, with --kernel-trace only; see MI355X_MICROARCH.md)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" \
           "SQC_ICACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p$i -- python3 $GRAFT_REPO_ROOT/tools/kbench.py "$@" > $out/p$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + '/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if 'cheb' not in k: continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        # dispatches alternate between ablation settings in kbench order; print the per-dispatch list tail
        print('   %-28s n=%d  first=%.4g  last=%.4g' % (c, len(v), v[0], v[-1]))
PY
