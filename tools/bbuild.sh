#!/bin/bash
# Builds variants of the library with other -D flags for contract_bf16.hip into build_x/libchebgcn_<name>.so
# (select with CHEBGCN_LIB=...).  usage: tools/bbuild.sh name "-DCG_BF16_WGPC=2 ..." [name flags ...]
set -e
cd "$(dirname "$0")/../gcn_fmri_decoding_amd/csrc"
make -s
mkdir -p ../../build_x
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on $flags -Rpass-analysis=kernel-resource-usage -c contract_bf16.hip -o ../../build_x/contract_bf16_$name.o 2>&1 \
    | grep -i "Function Name\|VGPRs:\|AGPRs:\|Occupancy\|Scratch" | sed 's/.*remark: *//' | paste - - - - - | sed 's/\[-Rpass[^]]*\]//g' \
    | grep "fwd_bf16_kernelILi1" | sed "s/^/$name: /" | cut -c1-220
  /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 graph.o recurrence.o recurrence4.o recurrence_ord.o recurrence_ord2.o recurrence_ord2a.o recurrence_ord_small.o contract.o ../../build_x/contract_bf16_$name.o pointwise.o head.o fused_small.o coarsen_host.o -o ../../build_x/libchebgcn_$name.so
done
