# Builds the gfx950 C-ABI library in-tree (hipcc cross-compiles without a GPU).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := vllm_omni_draft_amd/csrc
SRCS  := $(CSRC)/sd_runtime.hip $(CSRC)/sd_sampling.hip $(CSRC)/sd_linear.hip $(CSRC)/sd_linear_wide.hip $(CSRC)/sd_attention.hip
OBJS  := $(SRCS:.hip=.o)
LIB   := vllm_omni_draft_amd/libsd_amd.so
EXTRA ?=
# -ffp-contract=off: whether a*b+c becomes one fma is otherwise the compiler's choice per
# template instantiation, and two instantiations of a kernel (one or two token tiles, ring
# depths, ...) must give a row the same bits.
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $(EXTRA)

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/sd_common.h include/sd_amd.h
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean
