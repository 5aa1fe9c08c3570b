# Build libpromonet_hip.so (gfx950) and nothing else. `make -j4`.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
# TUNING=1: A/B scaffolding (PM_FUSION / PM_NO_NARROW / PM_FARGAN environment
# switches, phase-timeline stamps, ablation defines). Never in the shipped .so.
TUNING ?= 0
CXXFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-value -fno-honor-nans
ifeq ($(TUNING),1)
CXXFLAGS += -DPM_TUNING
endif
SRC = promonet_amd/csrc
OBJ = build/obj
LIB = promonet_amd/lib/libpromonet_hip.so
OBJS = $(OBJ)/pm_api.o $(OBJ)/pm_conv_f16.o $(OBJ)/pm_conv_bf16.o $(OBJ)/pm_conv_f32.o \
       $(OBJ)/pm_conv_f16x3.o $(OBJ)/pm_conv_f16a2.o \
       $(OBJ)/pm_conv_f16_mrf.o $(OBJ)/pm_conv_bf16_mrf.o
# the whole-MRF kernels: see pm_conv_bf16_mrf.hip
MRF_FLAGS = -mllvm -amdgpu-sched-strategy=max-ilp
HDRS = $(wildcard $(SRC)/*.h) include/promonet_hip.h

all: $(LIB)

$(OBJ)/%_mrf.o: $(SRC)/%_mrf.hip $(HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(CXXFLAGS) $(MRF_FLAGS) -c $< -o $@

$(OBJ)/%.o: $(SRC)/%.hip $(HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p promonet_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@
	@python3 scripts/check_spills.py $(OBJS) || true

# fails when any kernel spills VGPRs or uses scratch memory
check: $(LIB)
	python3 scripts/check_spills.py $(OBJS)

clean:
	rm -rf build $(LIB)

# Micro-benchmarks behind DESIGN.md section 6 (run the binaries on the GPU box)
MICRO = mfma_peak mfma_shapes mma_loop phase_overlap overlap2 lds_dma xcd_exchange
micro: $(addprefix promonet_amd/lib/,$(MICRO))
promonet_amd/lib/%: scripts/micro/%.hip promonet_amd/csrc/pm_conv.h
	$(HIPCC) $(CXXFLAGS) -Wno-unused-result $< -o $@
