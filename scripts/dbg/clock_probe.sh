#!/bin/bash
# ON THE GPU BOX: effective clock (GRBM_GUI_ACTIVE / 8 / duration) + MFMA-busy of the kernels
# matching $2, for library variant $1. usage: scripts/dbg/clock_probe.sh _noduo mrf
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1; PAT=$2
OUT=$ROOT/gpurun_out/clock$V
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-traffic --no-secondary --sustain 0 --steps 4 --warmup 1"
PROMONET_HIP_LIB=$ROOT/promonet_amd/lib/libpromonet_hip$V.so timeout 300 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU \
   --kernel-trace -d $OUT -o p -- $BENCH > $OUT/log 2>&1
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
f = glob.glob('$OUT/**/*counter_collection.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if '$PAT' in r['Kernel_Name']:
        rows[(r['Kernel_Name'][:60], r['Dispatch_Id'])][r['Counter_Name']].append(float(r['Counter_Value']))
        rows[(r['Kernel_Name'][:60], r['Dispatch_Id'])]['t'] = [float(r['End_Timestamp']) - float(r['Start_Timestamp'])]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (k, d), c in rows.items():
    for n, v in c.items(): agg[k][n].append(sum(v))
for k, c in agg.items():
    t = sum(c['t']) / len(c['t']); g = sum(c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE'])
    m = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES'])
    print('$V', k, 'n=%d dur %.1f us clock %.3f GHz mfma_busy %.3e -> pipe busy %.1f %%' % (len(c['t']), t / 1e3, g / 8 / t, m, 100 * m / 1024 / (g / 8)))
PY
