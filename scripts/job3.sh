cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 600 python -m pytest tests/test_gpu_edit.py tests/test_gpu_fargan.py -m gpu -q -s -k "selective or ragged" > gpurun_out/r02c/pytest.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r02c/pytest.log
export PROMONET_HIP_LIB=$GRAFT_REPO_ROOT/promonet_amd/lib/libpromonet_hip_tuning.so
timeout 300 python scripts/timeline_fargan.py 2>&1 | tee gpurun_out/r02c/timeline_fargan.txt
timeout 300 python scripts/timeline_block.py 2>&1 | tee gpurun_out/r02c/timeline_block.txt
timeout 300 python scripts/timeline.py 2>&1 | tee gpurun_out/r02c/timeline_pair.txt
