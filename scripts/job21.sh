#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r02v; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
PROMONET_HIP_LIB=$GRAFT_REPO_ROOT/promonet_amd/lib/libpromonet_hip_tune.so timeout 300 python scripts/timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/timeline_pair.txt
timeout 900 bash scripts/ab.sh _base "" 2>&1 | tee $OUT/ab_persistent2.txt
