#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r05i; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_preprocess_full.py tests/test_gpu_model.py -q -x -k "preprocess or spectrogram or loudness or mel or walk" > $OUT/pytest_fft.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_fft.log
tail -3 $OUT/pytest_fft.log
timeout 600 python bench.py --model fargan --steps 3 --warmup 1 > $OUT/bench_fargan.json 2> $OUT/bench_fargan.err; echo "bench fargan rc $?"
scripts/ab_preprocess.sh "" _base 2>&1 | tee $OUT/ab_mel_quads.txt
