# Top-level build: HIP library (gfx950), host C library, oracle (test infrastructure).
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) --offload-compress -O3 -std=c++17 -fPIC -Iinclude -Imulticlust_amd/csrc -Wall -Wno-unused-function $(EXTRA)
CFLAGS   = -std=gnu11 -O2 -fPIC -Wall -Wextra -Iinclude

OBJ  = build/obj
LIB  = multiclust_amd/lib
KS   = $(shell seq 1 64)
KOBJ = $(foreach k,$(KS),$(OBJ)/mchip_k$(k).o)

BIN  = multiclust_amd/bin

all: $(LIB)/libmulticlust_hip.so $(LIB)/libmulticlust_host.so $(BIN)/multiclust oracle

# -amdgpu-spill-vgpr-to-agpr=false: ROCm 7.2's hipcc loses a lane when it reloads a multi-dword VGPR tuple whose spill was split
# between AGPRs and scratch (fewer free AGPRs than lanes).  Shown in profiles/r03_k52_spill.md on the one-lane K = 52 instance of
# k_individual_sparse<4,true,true,false>: gn.g0 = v[0:3] goes to a253, a254, a255 + scratch; a[128:131] gets three lanes back,
# a129 <- a255 is never emitted, and every block after a chunk's first gathers rows from garbage allele indices (-inf / NaN).
# With the flag tuples spill to scratch whole.  Only kernels that use the whole register file spill (K > 20); applied to every
# translation unit all the same.  Reproducer: `make exp-k52`, scripts/diag/k52_spill.sh.
KFLAGS = -mllvm -amdgpu-spill-vgpr-to-agpr=false
$(OBJ)/mchip_k%.o: multiclust_amd/csrc/mchip_kernels_k.hip multiclust_amd/csrc/mchip_internal.h multiclust_amd/csrc/mchip_finalize.h include/multiclust_hip.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) -DMCHIP_K=$* -c $< -o $@

$(OBJ)/mchip.o: multiclust_amd/csrc/mchip.hip multiclust_amd/csrc/mchip_internal.h multiclust_amd/csrc/mchip_finalize.h multiclust_amd/csrc/mchip_progress.h include/multiclust_hip.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) -c $< -o $@

$(OBJ)/mchip_comm.o: multiclust_amd/csrc/mchip_comm.hip multiclust_amd/csrc/mchip_progress.h include/multiclust_hip.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) -c $< -o $@

$(LIB)/libmulticlust_hip.so: $(OBJ)/mchip.o $(OBJ)/mchip_comm.o $(KOBJ)
	@mkdir -p $(LIB)
	$(HIPCC) --offload-arch=$(ARCH) --offload-compress -shared -fPIC -o $@ $^ -ldl

HOST_SRC = $(filter-out multiclust_amd/host/mc_main.c,$(wildcard multiclust_amd/host/*.c))
$(LIB)/libmulticlust_host.so: $(HOST_SRC) $(wildcard multiclust_amd/host/*.h) include/multiclust_hip.h $(LIB)/libmulticlust_hip.so
	@mkdir -p $(LIB)
	$(CC) $(CFLAGS) -Imulticlust_amd/host -shared -o $@ $(HOST_SRC) -L$(LIB) -lmulticlust_hip -Wl,-rpath,'$$ORIGIN' -lm -lpthread

# the drop-in command line (same flags, reader and output files as the reference's `multiclust`)
$(BIN)/multiclust: multiclust_amd/host/mc_main.c $(LIB)/libmulticlust_host.so
	@mkdir -p $(BIN)
	$(CC) $(CFLAGS) -Imulticlust_amd/host -o $@ $< -L$(LIB) -lmulticlust_host -lmulticlust_hip -Wl,-rpath,'$$ORIGIN/../lib' -lm -lpthread

oracle: $(LIB)/libmulticlust_host.so
	$(MAKE) -C oracle all

# microbenchmarks behind profiles/r01_fp64_microbench.txt and r01_op_cost_microbench.txt (run on the GPU box; binaries are not tracked)
micro: scripts/micro/fp64_micro scripts/micro/op_cost scripts/micro/lds_valu scripts/micro/vgpr_banks scripts/micro/mfma_sustained
scripts/micro/%: scripts/micro/%.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -o $@ $<

# experiment behind profiles/r02_scatter_experiment.txt: the K = 8 kernels with MCHIP_EXP_SCATTER, linked with
# the product's other objects into a library of its own; scripts/diag/scatter_exp.sh loads it through MCHIP_LIB_PATH
exp-scatter: $(LIB)/libmulticlust_hip.so
	@mkdir -p build/exp scripts/exp
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) -DMCHIP_EXP_SCATTER -DMCHIP_K=8 -c multiclust_amd/csrc/mchip_kernels_k.hip -o build/exp/mchip_k8.o
	$(HIPCC) --offload-arch=$(ARCH) --offload-compress -shared -fPIC -o scripts/exp/libmulticlust_hip_scatter.so \
		$(filter-out $(OBJ)/mchip_k8.o,$(OBJ)/mchip.o $(OBJ)/mchip_comm.o $(KOBJ)) build/exp/mchip_k8.o -ldl

# diagnosis behind profiles/r03_k52_spill.md: the K = 52 kernels with one lane per individual (MCHIP_FORCE_SPLIT1: the instance
# that returned -inf / NaN before the lane split) built with hipcc's default VGPR->AGPR spilling and with KFLAGS, each linked
# with a matching mchip.o into a library of its own; scripts/diag/k52_spill.sh runs both through MCHIP_LIB_PATH
exp-k52: $(LIB)/libmulticlust_hip.so
	@mkdir -p build/exp scripts/exp
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) -DMCHIP_FORCE_SPLIT1 -c multiclust_amd/csrc/mchip.hip -o build/exp/mchip_split1.o
	$(HIPCC) $(HIPFLAGS) -DMCHIP_FORCE_SPLIT1 -DMCHIP_K=52 -c multiclust_amd/csrc/mchip_kernels_k.hip -o build/exp/mchip_k52_agpr.o
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) -DMCHIP_FORCE_SPLIT1 -DMCHIP_K=52 -c multiclust_amd/csrc/mchip_kernels_k.hip -o build/exp/mchip_k52_scratch.o
	for v in agpr scratch; do $(HIPCC) --offload-arch=$(ARCH) --offload-compress -shared -fPIC -o scripts/exp/libmulticlust_hip_k52$$v.so \
		build/exp/mchip_split1.o $(OBJ)/mchip_comm.o $(filter-out $(OBJ)/mchip_k52.o,$(KOBJ)) build/exp/mchip_k52_$$v.o -ldl; done

# A/B builds of the K = 8 kernels: `make exp-k8 EXPNAME=w5 EXPFLAGS=-DMCHIP_SPARSE_WAVES=5` -> scripts/exp/libmulticlust_hip_w5.so
# (the K = 8 object rebuilt with EXPFLAGS, everything else the product's); scripts/diag/ab.sh times variants against the shipped
# library in alternation on one box
EXPNAME ?= exp
exp-k8: $(LIB)/libmulticlust_hip.so
	@mkdir -p build/exp scripts/exp
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) $(EXPFLAGS) -DMCHIP_K=8 -c multiclust_amd/csrc/mchip_kernels_k.hip -o build/exp/mchip_k8_$(EXPNAME).o
	$(HIPCC) --offload-arch=$(ARCH) --offload-compress -shared -fPIC -o scripts/exp/libmulticlust_hip_$(EXPNAME).so \
		$(filter-out $(OBJ)/mchip_k8.o,$(OBJ)/mchip.o $(OBJ)/mchip_comm.o $(KOBJ)) build/exp/mchip_k8_$(EXPNAME).o -ldl

# the same for any K: `make exp-k EXPK=40 EXPNAME=k40s4 EXPFLAGS=-DMCHIP_COL_SPLIT4_ABOVE=24`
EXPK ?= 8
exp-k: $(LIB)/libmulticlust_hip.so
	@mkdir -p build/exp scripts/exp
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) $(EXPFLAGS) -DMCHIP_K=$(EXPK) -c multiclust_amd/csrc/mchip_kernels_k.hip -o build/exp/mchip_k$(EXPK)_$(EXPNAME).o
	$(HIPCC) --offload-arch=$(ARCH) --offload-compress -shared -fPIC -o scripts/exp/libmulticlust_hip_$(EXPNAME).so \
		$(filter-out $(OBJ)/mchip_k$(EXPK).o,$(OBJ)/mchip.o $(OBJ)/mchip_comm.o $(KOBJ)) build/exp/mchip_k$(EXPK)_$(EXPNAME).o -ldl

clean:
	rm -rf build $(LIB)/*.so $(BIN) scripts/micro/fp64_micro scripts/micro/op_cost scripts/micro/lds_valu scripts/micro/vgpr_banks scripts/micro/mfma_sustained scripts/exp/*.so
	$(MAKE) -C oracle clean

.PHONY: all oracle micro exp-scatter exp-k52 exp-k8 exp-k clean
