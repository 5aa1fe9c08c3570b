#!/bin/bash
# ON THE GPU BOX: FARGAN (BASELINE.json config 5) profile of the round -> gpurun_out/fargan/
# kernel-trace stats, the PMC-measured HBM traffic in the bench line (both storage modes), the
# phase timeline of a sub-frame step (tuning build, if present).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fargan
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for storage in fp32 mixed; do
  timeout 600 python $ROOT/bench.py --model fargan --dtype $storage --steps 5 --warmup 2 --sustain 0 --no-cpu-baseline --no-secondary > $OUT/bench_$storage.json 2> $OUT/bench_$storage.err
  timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_$storage -o stats -- python $ROOT/bench.py --model fargan --dtype $storage --steps 5 --warmup 2 --sustain 0 --no-cpu-baseline --no-secondary --no-traffic > $OUT/stats_$storage.log 2>&1
  f=$(find $OUT/stats_$storage -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$storage.csv
  rm -rf $OUT/stats_$storage
done
if [ -f $ROOT/promonet_amd/lib/libpromonet_hip_tuning.so ]; then
  for storage in fp32 mixed; do
    FARGAN_WEIGHT_DTYPE=$storage PROMONET_HIP_LIB=$ROOT/promonet_amd/lib/libpromonet_hip_tuning.so timeout 300 python $ROOT/scripts/timeline_fargan.py > $OUT/timeline_$storage.txt 2>&1
  done
fi
cd $ROOT; head -c 1500 $OUT/bench_mixed.json; echo; cat $OUT/kernel_stats_mixed.csv | head -8
