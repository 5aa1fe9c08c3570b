#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r02ab; mkdir -p $OUT
timeout 300 bash scripts/power_trace.sh $OUT/power_bf16 2>&1 | tee $OUT/power_bf16.txt
cat $OUT/power_bf16.static | head -40
head -c 600 $OUT/power_bf16.samples
