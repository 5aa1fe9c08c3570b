cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "block_iteration or golden or full_size_f16" > gpurun_out/r02e/pytest2.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02e/pytest2.log
AB_FILTER=pair_c128 bash scripts/ab.sh "" _nodma 2>&1 | tee gpurun_out/r02e/ab2.log
PROMONET_HIP_LIB=$GRAFT_REPO_ROOT/promonet_amd/lib/libpromonet_hip_tuning.so timeout 300 python scripts/timeline.py 2>&1 | grep -v "chunk0" | tee gpurun_out/r02e/timeline_pair_dma.txt
