cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -s -k "every_sample or saturate" 2>&1 | grep -E "full size|passed|failed|Error" | tee gpurun_out/r02j/every_sample.log
bash scripts/build_variant.sh nosat -DPM_NO_F16_SATURATE > /dev/null 2>&1
bash scripts/ab.sh "" _nosat 2>&1 | tee gpurun_out/r02j/ab_saturate_upper.log
