#!/bin/bash
# Build an A/B variant of the library: scripts/build_variant.sh <suffix> <-DFLAGS...>
# -> promonet_amd/lib/libpromonet_hip_<suffix>.so (select with PROMONET_HIP_LIB)
set -e
cd $(dirname $0)/..
SUF=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-value -fno-honor-nans $*"
mkdir -p build/obj_$SUF
for f in pm_api pm_conv_f16 pm_conv_bf16 pm_conv_f32 pm_conv_f16x3 pm_conv_f16a2; do
  /opt/rocm/bin/hipcc $F -c promonet_amd/csrc/$f.hip -o build/obj_$SUF/$f.o &
done
for f in pm_conv_f16_mrf pm_conv_bf16_mrf; do   # (Makefile: MRF_FLAGS)
  /opt/rocm/bin/hipcc $F ${MRF_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp} -c promonet_amd/csrc/$f.hip -o build/obj_$SUF/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj_$SUF/*.o -o promonet_amd/lib/libpromonet_hip_$SUF.so
