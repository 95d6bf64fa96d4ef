# Builds the gfx950 C-ABI library in-tree (hipcc cross-compiles without a GPU).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := vllm_omni_draft_amd/csrc
SRCS  := $(CSRC)/sd_runtime.hip $(CSRC)/sd_sampling.hip $(CSRC)/sd_linear.hip $(CSRC)/sd_linear_wide.hip $(CSRC)/sd_attention.hip $(CSRC)/sd_ipc.hip $(CSRC)/sd_chain.hip $(CSRC)/sd_draft_layer.hip
OBJS  := $(SRCS:.hip=.o)
LIB   := vllm_omni_draft_amd/libsd_amd.so
EXTRA ?=
# -ffp-contract=off: whether a*b+c becomes one fma is otherwise the compiler's choice per
# template instantiation, and two instantiations of a kernel (one or two token tiles, ring
# depths, ...) must give a row the same bits.
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $(EXTRA)

# The CPU build of the same C ABI (C++ / OpenMP): its own library, loaded only when the host
# code is told to (SD_AMD_BACKEND=cpu); the GPU library never falls back to it.
CPU_LIB := vllm_omni_draft_amd/libsd_cpu.so
CXX ?= g++

all: $(LIB) $(CPU_LIB)

$(CPU_LIB): vllm_omni_draft_amd/csrc_cpu/sd_cpu.cpp include/sd_amd.h
	$(CXX) -O3 -std=c++17 -fPIC -shared -fopenmp -mavx2 -mfma -ffp-contract=off -Wall -o $@ $<

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/sd_common.h include/sd_amd.h
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib

# Diagnostic build with in-kernel timestamps (experiments/*_stamps.py load it through SD_AMD_LIB);
# never loaded by the package on its own.
STAMP_DIR := build/stamps
STAMP_LIB := $(STAMP_DIR)/libsd_amd_stamps.so
STAMP_OBJS := $(patsubst $(CSRC)/%.hip,$(STAMP_DIR)/%.o,$(SRCS))
$(STAMP_DIR)/%.o: $(CSRC)/%.hip $(CSRC)/sd_common.h include/sd_amd.h
	@mkdir -p $(STAMP_DIR)
	$(HIPCC) $(FLAGS) -DSD_ATTN_STAMPS -DSD_LIN_STAMPS -DSD_CHAIN_STAMPS -DSD_DRAFT_STAMPS -DSD_WIDE_STAMPS -c $< -o $@
$(STAMP_LIB): $(STAMP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(STAMP_OBJS) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
stamps: $(STAMP_LIB)

# The library with a run-time switch between the two many-row linear kernels
# (experiments/bench_wide_shapes.py, experiments/bench_batch.py WIDE_FOUR=0): same sources,
# sd_linear_wide.hip compiled with -DSD_WIDE_SWITCH.
SWITCH_LIB := build/exp/libsd_amd_switch.so
$(SWITCH_LIB): $(OBJS)
	@mkdir -p build/exp
	$(HIPCC) $(FLAGS) -DSD_WIDE_SWITCH -c $(CSRC)/sd_linear_wide.hip -o build/exp/sd_linear_wide_switch.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(filter-out $(CSRC)/sd_linear_wide.o,$(OBJS)) \
		build/exp/sd_linear_wide_switch.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
wide-switch: $(SWITCH_LIB)

clean:
	rm -f $(OBJS) $(LIB) $(CPU_LIB)
	rm -rf $(STAMP_DIR) build/exp/*.o build/exp/*.so

.PHONY: all clean stamps wide-switch
