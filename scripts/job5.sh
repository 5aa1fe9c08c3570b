cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "block_iteration or whole_mrf or input_conv or golden or full_size or ragged or batched" > gpurun_out/r02e/pytest.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r02e/pytest.log
AB_FILTER=pair_c128 bash scripts/ab.sh "" _nodma 2>&1 | tee gpurun_out/r02e/ab.log
