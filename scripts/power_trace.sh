#!/bin/bash
# On the GPU box: socket power / shader clock sampled beside a sustained bench
# loop (is the conv path running at the power cap?).
# usage: scripts/power_trace.sh <outfile> [bench args...]
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
rocm-smi --showpower --showclocks --showmaxpower --showperflevel > $OUT.static 2>&1
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > $OUT.samples &
SAMPLER=$!
python $ROOT/bench.py --steps 400 --warmup 20 --sustain 0 --no-cpu-baseline --no-traffic --no-secondary "$@" > $OUT.bench.json 2>/dev/null
kill $SAMPLER
python - "$OUT" <<'PY'
import json, re, sys
out = sys.argv[1]
power, sclk = [], []
for line in open(out + '.samples'):
    try:
        card = json.loads(line)['card0']
    except Exception:
        continue
    for key, value in card.items():
        if 'Power' in key and 'W' in key:
            try: power.append(float(value))
            except ValueError: pass
        if key.startswith('sclk'):
            m = re.search(r'(\d+)Mhz', value)
            if m: sclk.append(int(m.group(1)))
def stats(v):
    v = sorted(v)
    return 'n=%d min %.0f median %.0f max %.0f' % (len(v), v[0], v[len(v) // 2], v[-1]) if v else 'none'
print('power (W):', stats(power))
print('sclk (MHz):', stats(sclk))
r = json.loads(open(out + '.bench.json').read().strip().splitlines()[-1])
print('bench: %.2f ms/step over %d steps' % (r['ms_per_step'], r['steps']))
PY
