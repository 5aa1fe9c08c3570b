cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02o
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "block_iteration or golden or ragged or windows or batched" > gpurun_out/r02o/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02o/pytest.log
AB_FILTER=pair_c128 bash scripts/ab.sh "" _nopersist 2>&1 | tee gpurun_out/r02o/ab_persist.log
