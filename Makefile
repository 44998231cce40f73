# Build of the MI355X-native DMRG hot path.  gfx950 only; no CUDA/dual paths.
#   make            -> dmrg.x_amd/libdmrgx_hip.so (product) + oracle/liboracle_kron.so (test infrastructure)
HIPCC      ?= hipcc
ARCH       ?= gfx950
PKG        := dmrg.x_amd
CSRC       := $(PKG)/csrc
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function
HIP_SRCS   := $(CSRC)/lib.hip $(CSRC)/ggemm.hip $(CSRC)/kron_plan.hip $(CSRC)/eigs.hip $(CSRC)/rdm.hip $(CSRC)/rotate.hip
HIP_OBJS   := $(HIP_SRCS:.hip=.o)

all: $(PKG)/libdmrgx_hip.so oracle/liboracle_kron.so

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/ggemm.h include/dmrgx.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(PKG)/libdmrgx_hip.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(HIP_OBJS) -o $@

# x86-64-v3 (AVX2/FMA), not -march=native: the .so is built here and travels to the GPU host
oracle/liboracle_kron.so: oracle/kron_ref.c
	gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC $< -o $@

clean:
	rm -f $(HIP_OBJS) $(PKG)/libdmrgx_hip.so oracle/liboracle_kron.so

.PHONY: all clean
