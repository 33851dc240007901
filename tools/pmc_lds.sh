#!/bin/bash
# usage: tools/pmc_lds.sh <tag> <root>  -- LDS counters of the recurrence forward kernel (gather-only and full)
tag=$1; root=$2
out=$GRAFT_REPO_ROOT/gpurun_out/pmclds_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o p -- python3 $root/tools/kbench.py --kernels recurrence_fwd --B 256 --iters 3 --ablate 17 0 > $out/p.log 2>&1
grep recurrence $out/p.log
