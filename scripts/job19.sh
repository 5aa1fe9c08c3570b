#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r02t; mkdir -p $OUT
for v in tunebase tune; do
  echo "== $v" | tee -a $OUT/timeline_pair.txt
  PROMONET_HIP_LIB=$GRAFT_REPO_ROOT/promonet_amd/lib/libpromonet_hip_$v.so timeout 300 python scripts/timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline_pair.txt
done
timeout 900 bash scripts/ab.sh "" _2wg 2>&1 | tee $OUT/ab_2wg.txt
