#!/bin/bash
# Builds variants of the library with other -D flags for the recurrence kernels into build_x/libchebgcn_<name>.so
# (select one with CHEBGCN_LIB=... python tools/kbench.py; A/B inside ONE gpurun call).
#   usage: tools/vbuild.sh name "-DCG_GATHER_ASM=0 ..." [name flags ...]      (-DCG_EXPERIMENT=1 -DCG_X=64: phase stamps)
set -e
cd "$(dirname "$0")/../gcn_fmri_decoding_amd/csrc"
make -s -j8
mkdir -p ../../build_x
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  for f in recurrence recurrence4 recurrence_ord recurrence_ord2 recurrence_ord2a recurrence_ord_small; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on $flags -c $f.hip -o ../../build_x/${f}_$name.o &
  done
  wait
  /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 graph.o ../../build_x/recurrence_$name.o ../../build_x/recurrence4_$name.o ../../build_x/recurrence_ord_$name.o ../../build_x/recurrence_ord2_$name.o ../../build_x/recurrence_ord2a_$name.o ../../build_x/recurrence_ord_small_$name.o contract.o contract_bf16.o pointwise.o head.o fused_small.o coarsen_host.o -o ../../build_x/libchebgcn_$name.so
done
