#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats + separate PMC
# passes of the FFT kernels (scripts/stft_only.py). usage: scripts/profile_stft.sh <tag>
set -u
TAG=${1:-stft}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/scripts/stft_only.py"
timeout 200 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- $CMD 22 > $OUT/stats.log 2>&1
timeout 200 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq -o sq -- $CMD 2 > $OUT/pmc_sq.log 2>&1
timeout 200 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- $CMD 2 > $OUT/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- $CMD 2 > $OUT/pmc_write.log 2>&1
cd $ROOT
python - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'fft' in r['Name'] or 'mel' in r['Name']:
            print('stats', r['Name'][:60], 'calls', r['Calls'], 'avg ns', r['AverageNs'])
for tag in ('sq', 'fetch', 'write'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + '/pmc_%s/**/*counter_collection.csv' % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'fft' in r['Kernel_Name']:
                acc[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        print('pmc', k, {c: sum(v) / len(v) for c, v in cs.items()}, 'dispatches', len(next(iter(cs.values()))))
PY
find $OUT -name "*.csv" -size +2M -delete
