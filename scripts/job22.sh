#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r02w; mkdir -p $OUT
export PROMONET_HIP_LIB=$GRAFT_REPO_ROOT/promonet_amd/lib/libpromonet_hip_tune.so
for st in 0 2; do
echo "== PM_STAGGER=$st" | tee -a $OUT/timeline_pair.txt
PM_STAGGER=$st timeout 300 python scripts/timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline_pair.txt
done
AB_FILTER=pair timeout 1200 bash scripts/ab_env.sh PM_STAGGER=0 PM_STAGGER=1 PM_STAGGER=2 PM_STAGGER=3 2>&1 | tee $OUT/ab_stagger.txt
