#!/bin/bash
# usage: tools/pmc_mfma.sh <tag> ; matrix-core utilisation counters of the contraction kernels (fp32 MFMA at the
# bench shape, bf16 MFMA at the config-5 shape).  Each rocprofv3 --pmc pass in its own run with --kernel-trace only.
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/pmcmfma_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "mfma\|GRBM_GUI_ACTIVE\|SQ_BUSY_CYCLES\|SQ_VALU_MFMA" | head -40 > $out/available.txt
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o a$i -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --B 64 --iters 3 --kernels contract_fwd contract_bwd_w_relu contract_bwd_x_relu > $out/a$i.log 2>&1
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o b$i -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --B 64 --fin 60 --fout 256 --K 5 --iters 3 --kernels contract_fwd contract_fwd_bf16 contract_fwd_bf16x3 contract_bwd_w_bf16 contract_bwd_w_bf16x3 contract_bwd_x_bf16 contract_bwd_x_bf16x3 > $out/b$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + '/**/*counter_collection.csv', recursive=True)):
    shape = 'config5' if '/b' in f.replace(out, '') else 'bench'
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'contract' not in k and 'reduce' not in k: continue
        acc[shape + ' ' + k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, d in sorted(acc.items()):
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    m = res[k]
    line = k + ' ' + ' '.join('%s=%.4g' % (c, x) for c, x in sorted(m.items()))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m and m['GRBM_GUI_ACTIVE'] > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES: matrix-pipe busy cycles summed over the chip's 1024 SIMDs (= 64 x the number of
        # v_mfma_f32_32x32x2_f32, 32 x the number of v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE: kernel cycles summed over the 8 XCDs
        line += '  -> MfmaUtil = %.3f' % (m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8.0 * 1024))
    print(line)
json.dump(res, open(out + '/mfma_raw.json', 'w'), indent=1)
PY
