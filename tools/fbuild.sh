#!/bin/bash
# Builds variants of the library with other -D flags for fused_small.hip into build_x/libchebgcn_<name>.so
# (select with CHEBGCN_LIB=...).  usage: tools/fbuild.sh name "-DCG_EXPERIMENT=1 -DCG_X=64" [name flags ...]
set -e
cd "$(dirname "$0")/../gcn_fmri_decoding_amd/csrc"
make -s
mkdir -p ../../build_x
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on $flags -Rpass-analysis=kernel-resource-usage -c fused_small.hip -o ../../build_x/fused_small_$name.o 2>&1 \
    | grep -i "Function Name\|VGPRs:\|AGPRs:\|Occupancy\|Scratch" | sed 's/.*remark: *//' | paste - - - - - | sed 's/\[-Rpass[^]]*\]//g' \
    | grep "fused_layer_kernelILi12ELi8" | sed "s/^/$name: /" | cut -c1-220
  /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 graph.o recurrence.o recurrence4.o recurrence_ord.o recurrence_ord2.o recurrence_ord2a.o recurrence_ord_small.o contract.o contract_bf16.o pointwise.o head.o ../../build_x/fused_small_$name.o coarsen_host.o -o ../../build_x/libchebgcn_$name.so
done
