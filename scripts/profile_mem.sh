#!/bin/bash
# Memory-path PMC passes (run ON THE GPU BOX): usage scripts/profile_mem.sh <tag>
set -u
TAG=${1:-mem}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-traffic --no-secondary --sustain 0 --steps 1 --warmup 0"
run() {  # run <name> <counters...>: one PMC pass, bounded (a bad counter set aborts and can hang)
    local name=$1; shift
    timeout 150 rocprofv3 --output-format csv --pmc "$@" --kernel-trace -d $OUT/pmc_$name -o $name -- $BENCH > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed"
}
run ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum
run tcc2 TCC_REQ_sum TCC_BUSY_avr
run lvl SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
run fifo SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
cd $ROOT
python - <<PY
import csv, re, json
from collections import defaultdict
from pathlib import Path
root = Path("$OUT")
def short(name):
    m = re.match(r'void (\w+)<(.*)>\(', name)
    return f"{m.group(1)}<{m.group(2).replace('Elem','').replace(' ','')}>" if m else name.split('(')[0][:50]
out = {}
for f in root.rglob('*counter_collection.csv'):
    s = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if 'conv_' not in k: continue
        s[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    for k in s:
        out.setdefault(k, {}).update({c: s[k][c] / n[k][c] for c in s[k]})
(root / 'summary_mem.json').write_text(json.dumps(out, indent=1))
for k in sorted(out):
    v = out[k]; print(k); print('   ' + '  '.join(f'{c}={x:.4g}' for c, x in sorted(v.items())))
PY
find $OUT -name "*.csv" -size +2M -delete
