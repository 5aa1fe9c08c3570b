#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the round's standard measurement set.
# usage: scripts/gpu_round.sh <tag> [tests|bench|all]
set -u
TAG=${1:-r02}; WHAT=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [[ $WHAT == all || $WHAT == tests ]]; then
  timeout 2400 python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1
  echo "pytest rc $?" | tee -a $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
if [[ $WHAT == all || $WHAT == bench ]]; then
  timeout 600 python bench.py > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; echo "bench bf16 (default) rc $?"
  timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-traffic --no-secondary > $OUT/bench_bf16_2ranks.json 2> $OUT/bench_bf16_2ranks.err; echo "bench 2 ranks rc $?"
  timeout 600 python bench.py --dtype f16 --no-cpu-baseline --no-traffic --no-secondary > $OUT/bench_f16.json 2> $OUT/bench_f16.err; echo "bench f16 rc $?"
  timeout 600 python bench.py --dtype fp32 --batch 8 --seconds 5 --no-cpu-baseline --no-traffic --no-secondary > $OUT/bench_fp32_config2.json 2> $OUT/bench_fp32.err; echo "bench fp32 rc $?"
  timeout 600 python bench.py --model fargan --dtype fp32 --steps 5 --warmup 1 > $OUT/bench_fargan.json 2> $OUT/bench_fargan.err; echo "bench fargan rc $?"
  for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], '%.2f ms/step' % r['ms_per_step'], 'sustained %.2f' % r.get('sustained_ms_per_step', 0), 'n_gpus', r['n_gpus'], 'roofline', r['roofline']['kernel'], '%.3f' % r['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], 'unreadable:', e)
PY
  done
fi
