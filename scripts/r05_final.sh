#!/bin/bash
# Round 5 final measurement set (GPU box): full GPU suite, smoke, the driver's
# bench command, the 2-rank folded run, FARGAN line, rocprofv3 stats + PMC.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r05}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
PM_RECORD_ERRORS=1 timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" | tee -a $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; echo "bench default rc $?"
timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-traffic --no-secondary > $OUT/bench_bf16_2ranks.json 2> $OUT/bench_bf16_2ranks.err; echo "bench 2 ranks rc $?"; grep -h WARNING $OUT/bench_bf16_2ranks.err | head -2
timeout 600 python bench.py --model fargan --steps 3 --warmup 1 > $OUT/bench_fargan.json 2> $OUT/bench_fargan.err; echo "bench fargan rc $?"
python - $OUT <<'PY'
import json, sys
out = sys.argv[1]
for name in ('bench_bf16', 'bench_bf16_2ranks', 'bench_fargan'):
    try:
        r = json.loads(open(f'{out}/{name}.json').read().strip().splitlines()[-1])
        print(name, '%.2f ms/step' % r['ms_per_step'], 'n_gpus', r['n_gpus'], 'roofline', (r.get('roofline') or {}).get('kernel'), '%.3f' % (r.get('roofline') or {}).get('frac', 0))
    except Exception as e:
        print(name, 'unreadable:', e)
PY
scripts/profile_gpu.sh $TAG > /dev/null 2>&1; tail -3 gpurun_out/prof_$TAG/summary.txt
find gpurun_out -name "*.csv" -size +2M -delete
