#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + separate
# PMC passes of the bench command; summaries land in gpurun_out/prof_<tag>/.
# usage: scripts/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-traffic --no-secondary --sustain 0 $*"
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH --steps 20 --warmup 2 > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS \
    --kernel-trace -d $OUT/pmc_sq -o sq -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE \
    --kernel-trace -d $OUT/pmc_lds -o lds -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_lds.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_write.log 2>&1
cd $ROOT
python scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
# keep the merged output small: drop the raw per-dispatch traces
find $OUT -name "*.csv" -size +2M -delete
tail -60 $OUT/summary.txt
