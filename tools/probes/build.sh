#!/bin/bash
# Builds the stand-alone probes (binaries are git-ignored; they travel to the GPU box with the gpurun snapshot).
cd "$(dirname "$0")"
for p in ta_lds_probe hbm_stream_probe mfma_f32_probe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o $p $p.hip || exit 1
done
