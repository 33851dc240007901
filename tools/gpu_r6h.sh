#!/bin/bash
cd $GRAFT_REPO_ROOT
for args in "inference_pool_n212 reference x" "inference_pool_n212 length! x" "inference_pool_n212 length! nomaps" "inference_pool6_n512 length! x"; do
  python tools/probes/capture_relabel_probe.py $args 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -8
done
