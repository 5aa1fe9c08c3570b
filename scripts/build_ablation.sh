#!/bin/bash
# Timing-experiment builds of the f16 kernels (WRONG numerics by design):
#   libpromonet_hip_noA.so  no weight (A) stream from L2 in the MFMA loop
#   libpromonet_hip_noB.so  no activation (B) stream from LDS
#   libpromonet_hip_noAB.so neither: MFMA + epilogues + staging only
set -e
cd $(dirname $0)/..
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-value -fno-honor-nans"
for v in A B AB; do
  D="-DPM_TUNING"; [[ $v == *A* ]] && D="$D -DPM_ABLATE_A"; [[ $v == *B* ]] && D="$D -DPM_ABLATE_B"
  /opt/rocm/bin/hipcc $F $D -c promonet_amd/csrc/pm_conv_f16.hip -o build/obj/pm_conv_f16_no$v.o &
done
wait
for v in A B AB; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj/pm_api.o build/obj/pm_conv_f16_no$v.o build/obj/pm_conv_bf16.o build/obj/pm_conv_f32.o build/obj/pm_conv_f16x3.o build/obj/pm_conv_f16_mrf.o build/obj/pm_conv_bf16_mrf.o -o promonet_amd/lib/libpromonet_hip_no$v.so
done
ls -la promonet_amd/lib
