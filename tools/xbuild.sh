#!/bin/bash
# Builds experiment variants of the library (recurrence*.hip with -DCG_EXPERIMENT -DCG_X=<bits>; 64 = phase stamps)
# into build_x/libchebgcn_x<bits>.so; select one with CHEBGCN_LIB=... python tools/kbench.py.
set -e
cd "$(dirname "$0")/../gcn_fmri_decoding_amd/csrc"
make -s
mkdir -p ../../build_x
for x in "$@"; do
  for f in recurrence recurrence4 recurrence_ord recurrence_ord2 recurrence_ord2a recurrence_ord_small; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -DCG_EXPERIMENT=1 -DCG_X=$x -c $f.hip -o ../../build_x/${f}_x$x.o
  done
  /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 graph.o ../../build_x/recurrence_x$x.o ../../build_x/recurrence4_x$x.o ../../build_x/recurrence_ord_x$x.o ../../build_x/recurrence_ord2_x$x.o ../../build_x/recurrence_ord2a_x$x.o ../../build_x/recurrence_ord_small_x$x.o contract.o contract_bf16.o pointwise.o head.o fused_small.o coarsen_host.o -o ../../build_x/libchebgcn_x$x.so
done
