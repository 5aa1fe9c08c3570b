#!/bin/bash
# Run ON THE GPU BOX: the round's bench lines, rocprofv3 kernel stats (20
# profiled steps) + PMC passes for both 16-bit operand types, the preprocessing
# bench and its kernel stats. usage: scripts/final_round.sh <tag>
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
scripts/gpu_round.sh $TAG bench
scripts/profile_gpu.sh $TAG > /dev/null 2>&1; tail -3 gpurun_out/prof_$TAG/summary.txt
scripts/profile_gpu.sh ${TAG}_f16 --dtype f16 > /dev/null 2>&1; tail -3 gpurun_out/prof_${TAG}_f16/summary.txt
mkdir -p gpurun_out/$TAG/preprocess
python scripts/bench_preprocess.py 2>/dev/null | tail -1 > gpurun_out/$TAG/preprocess/bench_preprocess.json
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $ROOT/gpurun_out/$TAG/preprocess/stats -o stats -- python $ROOT/scripts/bench_preprocess.py > $ROOT/gpurun_out/$TAG/preprocess/stats.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $ROOT/gpurun_out/$TAG/fargan_stats -o stats -- python $ROOT/bench.py --model fargan --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-secondary --sustain 0 > $ROOT/gpurun_out/$TAG/fargan_stats.log 2>&1)
find gpurun_out/$TAG -name "*.csv" -size +2M -delete
find gpurun_out/$TAG -name "*kernel_stats.csv" | head
