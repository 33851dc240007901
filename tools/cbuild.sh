#!/bin/bash
# Builds variants of the library with other -D flags for contract.hip into build_x/libchebgcn_<name>.so
# (select with CHEBGCN_LIB=...).  usage: tools/cbuild.sh name "-DCG_LB_FWD=4 ..." [name flags ...]
set -e
cd "$(dirname "$0")/../gcn_fmri_decoding_amd/csrc"
make -s
mkdir -p ../../build_x
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on $flags -Rpass-analysis=kernel-resource-usage -c contract.hip -o ../../build_x/contract_$name.o 2>&1 \
    | grep -i "Function Name\|VGPRs:\|AGPRs:\|Occupancy\|Scratch" | sed 's/.*remark: *//' | paste - - - - - | sed 's/\[-Rpass[^]]*\]//g' \
    | grep "fwd_kernelILi1\|fwd_ring\|bwd_x_lds\|bwd_x_kernelILb1ELb1ELb0\|bwd_w_kernelILi5ELb1\|bwd_w_db_kernelILi5ELb1" | sed "s/^/$name: /" | cut -c1-200
  /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 graph.o recurrence.o recurrence4.o recurrence_ord.o recurrence_ord2.o recurrence_ord2a.o recurrence_ord_small.o ../../build_x/contract_$name.o contract_bf16.o pointwise.o head.o fused_small.o coarsen_host.o -o ../../build_x/libchebgcn_$name.so
done
