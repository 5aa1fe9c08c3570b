#!/bin/bash
# Round 6 final measurement set (GPU box): full GPU suite, smoke, the driver's bench command,
# the other operand modes, FARGAN, the precision sweeps of the default mode, rocprofv3 stats + PMC.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r06}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
PM_RECORD_ERRORS=1 timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log; cp gpurun_out/measured_errors.json $OUT/ 2>/dev/null
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" | tee -a $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; echo "bench default rc $?"
timeout 600 python bench.py --dtype checkpoint --no-cpu-baseline --no-secondary > $OUT/bench_checkpoint.json 2> $OUT/bench_checkpoint.err; echo "bench checkpoint rc $?"
timeout 600 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --no-traffic > $OUT/bench_f16.json 2> $OUT/bench_f16.err
timeout 600 python bench.py --dtype fp32 --batch 8 --seconds 5 --no-cpu-baseline --no-secondary --no-traffic > $OUT/bench_fp32_config2.json 2> $OUT/bench_fp32_config2.err
timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-traffic --no-secondary > $OUT/bench_bf16_2ranks.json 2> $OUT/bench_bf16_2ranks.err; echo "bench 2 ranks rc $?"
timeout 600 python bench.py --model fargan --steps 3 --warmup 1 --no-secondary > $OUT/bench_fargan.json 2> $OUT/bench_fargan.err; echo "bench fargan rc $?"
timeout 900 python scripts/checkpoint_schedule.py > $OUT/checkpoint_schedule.txt 2>&1
timeout 900 python scripts/precision_seeds.py 5 > $OUT/precision_seeds.txt 2>&1
python - $OUT <<'PY'
import json, sys
out = sys.argv[1]
for name in ('bench_bf16', 'bench_checkpoint', 'bench_f16', 'bench_fp32_config2', 'bench_bf16_2ranks', 'bench_fargan'):
    try:
        r = json.loads(open(f'{out}/{name}.json').read().strip().splitlines()[-1])
        rf = r.get('roofline') or {}
        print(name, '%.2f ms/step' % r['ms_per_step'], 'n_gpus', r['n_gpus'], 'roofline', rf.get('kernel'), '%.3f' % rf.get('frac', 0), 'sustained', rf.get('sustained_peak'))
    except Exception as e:
        print(name, 'unreadable:', e)
PY
scripts/profile_gpu.sh $TAG > /dev/null 2>&1; tail -3 gpurun_out/prof_$TAG/summary.txt
find gpurun_out -name "*.csv" -size +2M -delete
