#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ak
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02ak/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02ak/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/r02ak/bench_bf16.json 2> gpurun_out/r02ak/bench.err; echo "bench rc $?"
timeout 600 python bench.py --model fargan --dtype fp32 --steps 5 --warmup 1 > gpurun_out/r02ak/bench_fargan.json 2>/dev/null; echo "fargan rc $?"
python - <<'PY'
import json
for f in ('bench_bf16','bench_fargan'):
    r=json.loads(open('gpurun_out/r02ak/%s.json'%f).read().strip().splitlines()[-1])
    print(f, r['ms_per_step'], r['value'], r['roofline'].get('frac'))
PY
