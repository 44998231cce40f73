# Build of the MI355X-native DMRG hot path.  gfx950 only; no CUDA/dual paths.
#   make            -> dmrg.x_amd/libdmrgx_hip.so (product) + oracle/liboracle_kron.so (test infrastructure)
HIPCC      ?= hipcc
ARCH       ?= gfx950
PKG        := dmrg.x_amd
CSRC       := $(PKG)/csrc
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function -Wno-pass-failed -Wno-inline-asm
HIP_SRCS   := $(CSRC)/lib.hip $(CSRC)/pool.hip $(CSRC)/ggemm.hip $(CSRC)/kron_plan.hip $(CSRC)/eigs.hip $(CSRC)/rdm.hip $(CSRC)/hqr.hip $(CSRC)/symeig.hip $(CSRC)/rotate.hip $(CSRC)/comm.hip
HIP_OBJS   := $(HIP_SRCS:.hip=.o)

HOST       := $(PKG)/host
HOST_HDRS  := $(wildcard $(HOST)/*.hpp) include/dmrgx.h

all: $(PKG)/libdmrgx_hip.so $(PKG)/dmrgx-square-lattice $(PKG)/dmrgx-host-tool oracle/liboracle_kron.so $(CSRC)/ggemm.isa.ok

# the grouped GEMM keeps operands in flight in registers the compiler must never touch (csrc/ggemm.hip, "DEEP staging registers"):
# the generated ISA is scanned on every build
$(CSRC)/ggemm.isa.ok: $(CSRC)/ggemm.hip $(CSRC)/ggemm.h $(CSRC)/common.h tools/check_staging_regs.py
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only $< -o $(CSRC)/ggemm.isa.s
	python3 tools/check_staging_regs.py $(CSRC)/ggemm.isa.s
	touch $@

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/ggemm.h $(CSRC)/hqr.h $(CSRC)/symeig.h include/dmrgx.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(PKG)/libdmrgx_hip.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(HIP_OBJS) -ldl -lpthread -lrt -o $@

# host sweep engine (plain C++17 over the C ABI: no HIP headers needed)
$(PKG)/dmrgx-square-lattice: $(HOST)/DMRG-SquareLattice.cpp $(HOST_HDRS) $(PKG)/libdmrgx_hip.so
	g++ -std=c++17 -O2 -pthread -Wall -Wno-unused-variable -Iinclude -I$(HOST) $< -L$(PKG) -ldmrgx_hip -Wl,-rpath,'$$ORIGIN' -o $@

# host-logic test harness (runs without a GPU)
$(PKG)/dmrgx-host-tool: $(HOST)/host_tool.cpp $(HOST_HDRS) $(PKG)/libdmrgx_hip.so
	g++ -std=c++17 -O1 -pthread -Wall -Wno-unused-variable -Iinclude -I$(HOST) $< -L$(PKG) -ldmrgx_hip -Wl,-rpath,'$$ORIGIN' -o $@

# measurement probes (developer tools; not part of `all`): the library kernel on synthetic products, the barrier-free
# one-wave-per-tile prototype it is compared with, the bare-MFMA ceiling
probes: tools/ggemm_probe tools/wgemm_probe tools/mfma_f64_probe
tools/ggemm_probe: tools/ggemm_probe.hip $(PKG)/libdmrgx_hip.so
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Iinclude -I$(CSRC) $< -L$(PKG) -ldmrgx_hip -Wl,-rpath,'$$ORIGIN/../$(PKG)' -o $@
tools/wgemm_probe: tools/wgemm_probe.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 $< -o $@
tools/mfma_f64_probe: tools/mfma_f64_probe.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 $< -o $@

# the reference's own driver source must compile against these headers (only where the reference tree is present)
dropin-check:
	g++ -std=c++17 -fsyntax-only -Iinclude -I$(HOST) /root/reference/src/DMRG-SquareLattice.cpp && echo "drop-in OK: reference src/DMRG-SquareLattice.cpp compiles against dmrg.x_amd/host headers"

# x86-64-v3 (AVX2/FMA), not -march=native: the .so is built here and travels to the GPU host
oracle/liboracle_kron.so: oracle/kron_ref.c
	gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC $< -o $@

clean:
	rm -f $(HIP_OBJS) $(CSRC)/ggemm.isa.ok $(CSRC)/ggemm.isa.s $(PKG)/libdmrgx_hip.so $(PKG)/dmrgx-square-lattice $(PKG)/dmrgx-host-tool oracle/liboracle_kron.so

.PHONY: all clean dropin-check probes
