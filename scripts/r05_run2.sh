#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r05b; mkdir -p $OUT; cd $ROOT
timeout 600 python scripts/loudness_bins_diag.py > $OUT/loudness_bins_diag.txt 2>&1; tail -9 $OUT/loudness_bins_diag.txt
PM_RECORD_ERRORS=1 timeout 1500 python -m pytest tests/test_gpu_preprocess_full.py tests/test_gpu_fargan.py -q -s > $OUT/pytest_new.log 2>&1; echo "pytest new rc $?" | tee -a $OUT/pytest_new.log
grep -n "per bin\|passed\|failed" $OUT/pytest_new.log | tail
for round in 1 2 3; do
  for v in "" _nox3f32; do
    PROMONET_HIP_LIB=$ROOT/promonet_amd/lib/libpromonet_hip$v.so timeout 300 python bench.py --dtype fp32 --batch 8 --seconds 5 --steps 6 --warmup 2 --sustain 0 --no-cpu-baseline --no-traffic --no-secondary 2>$OUT/ab_err$v.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernels']
print('fp32 config2 variant[$v] round $round: %.2f ms | ' % r['ms_per_step'] + ' '.join('%s %.3f' % (n, v['ms_per_step']) for n, v in sorted(k.items()) if 'c32' in n or 'c64' in n))" | tee -a $OUT/ab_x3skew_f32.txt
  done
done
