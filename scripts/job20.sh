#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r02u; mkdir -p $OUT
PROMONET_HIP_LIB=$GRAFT_REPO_ROOT/promonet_amd/lib/libpromonet_hip_tune.so timeout 300 python scripts/timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline_pair.txt
