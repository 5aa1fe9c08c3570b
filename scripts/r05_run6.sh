#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r05g; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -k "transpose or golden or full_size_every or ragged" > $OUT/pytest_up.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_up.log; tail -3 $OUT/pytest_up.log
AB_FILTER=convT scripts/ab.sh "" _base 2>&1 | tee $OUT/ab_wide_upsampler.txt
