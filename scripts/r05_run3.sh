#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r05c; mkdir -p $OUT; cd $ROOT
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc $?"
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r05c/bench_default.json').read().strip().splitlines()[-1])
print('%.2f ms/step' % r['ms_per_step'], 'roofline', r['roofline']['kernel'], '%.3f' % r['roofline']['frac'], 'sustained peak', r['roofline'].get('sustained_peak'))
for k, v in r.get('secondary', {}).items():
    if isinstance(v, dict):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, str)) or a in ('error', 'skipped')})
        for a, b in v.items():
            if isinstance(b, dict):
                print('   ', a, {x: (round(y, 4) if isinstance(y, float) else y) for x, y in b.items() if not isinstance(y, (dict, str))})
    else:
        print(k, v)
PY
PM_RECORD_ERRORS=1 timeout 1500 python -m pytest tests/test_gpu_preprocess_full.py tests/test_gpu_distributed.py -q -x > $OUT/pytest_new.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_new.log
tail -4 $OUT/pytest_new.log
for round in 1 2 3; do
  for v in "" _nox3f32; do
    PROMONET_HIP_LIB=$ROOT/promonet_amd/lib/libpromonet_hip$v.so timeout 300 python bench.py --dtype fp32 --batch 8 --seconds 5 --steps 6 --warmup 2 --sustain 0 --no-cpu-baseline --no-traffic --no-secondary 2>$OUT/ab_err$v.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernels']
print('fp32 config2 variant[$v] round $round: %.2f ms | ' % r['ms_per_step'] + ' '.join('%s %.3f' % (n, v['ms_per_step']) for n, v in sorted(k.items()) if 'c32' in n or 'c64' in n))" | tee -a $OUT/ab_x3skew_f32.txt
  done
done
