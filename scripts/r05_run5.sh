#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r05e; mkdir -p $OUT; cd $ROOT
PM_RECORD_ERRORS=1 timeout 1500 python -m pytest tests/test_gpu_preprocess_full.py tests/test_gpu_model.py -q -x -k "preprocess or spectrogram or loudness or mel or walk" > $OUT/pytest_fft.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_fft.log
tail -5 $OUT/pytest_fft.log
scripts/ab_preprocess.sh "" _base 2>&1 | tee $OUT/ab_fft_packed.txt
