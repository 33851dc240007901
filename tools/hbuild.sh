#!/bin/bash
# Builds variants of the library with other -D flags for head.hip into build_x/libchebgcn_<name>.so (select with CHEBGCN_LIB=...).
#   usage: tools/hbuild.sh name "-DFC_X=1 ..." [name flags ...]
set -e
cd "$(dirname "$0")/../gcn_fmri_decoding_amd/csrc"
make -s
mkdir -p ../../build_x
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on $flags -c head.hip -o ../../build_x/head_$name.o
  /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 graph.o recurrence.o recurrence4.o contract.o contract_bf16.o pointwise.o ../../build_x/head_$name.o coarsen_host.o -o ../../build_x/libchebgcn_$name.so
done
