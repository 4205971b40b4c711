# Builds the product library (HIP, gfx950), the host helper library and the CPU oracle.
ROCM_PATH ?= /opt/rocm
HIPCC ?= $(ROCM_PATH)/bin/hipcc
ARCH  ?= gfx950
CXX   ?= g++
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
CSRC := dqmc_amd/csrc
OBJS := $(CSRC)/gemm.o $(CSRC)/elementwise.o $(CSRC)/checkerboard.o $(CSRC)/update.o $(CSRC)/update_sm.o $(CSRC)/qr.o $(CSRC)/qr_colown.o $(CSRC)/qr_coop.o $(CSRC)/qr_panel.o $(CSRC)/lu.o $(CSRC)/lu_blocked.o $(CSRC)/lu_gj.o $(CSRC)/tri_solve.o $(CSRC)/engine.o $(CSRC)/replica.o

all: dqmc_amd/libdqmc_hip.so dqmc_amd/libdqmc_host.so dqmc_amd/dqmc_driver oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/wave.h include/dqmc_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/qr_colown.o: $(CSRC)/qr_colown_regs.inc
$(CSRC)/update.o: $(CSRC)/walk_bodies.inc
$(CSRC)/walk_bodies.inc: scripts/gen_walk_bodies.py
	python3 scripts/gen_walk_bodies.py
$(CSRC)/qr_colown_regs.inc: scripts/gen_qr_colown_regs.py
	python3 scripts/gen_qr_colown_regs.py

dqmc_amd/libdqmc_hip.so: $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -L$(ROCM_PATH)/lib -lrccl -Wl,-rpath,$(ROCM_PATH)/lib

dqmc_amd/libdqmc_host.so: dqmc_amd/host/host_capi.cpp dqmc_amd/host/dqmc_host.hpp dqmc_amd/host/results_h5.hpp include/dqmc_hip.h dqmc_amd/libdqmc_hip.so
	$(CXX) -O2 -std=c++17 -fPIC -shared -pthread -Iinclude -Idqmc_amd/host -o $@ dqmc_amd/host/host_capi.cpp -Ldqmc_amd -ldqmc_hip -ldl -Wl,-rpath,'$$ORIGIN'

dqmc_amd/dqmc_driver: dqmc_amd/host/main.cpp dqmc_amd/host/dqmc_host.hpp dqmc_amd/host/results_h5.hpp include/dqmc_hip.h dqmc_amd/libdqmc_hip.so
	$(CXX) -O2 -std=c++17 -pthread -Iinclude -Idqmc_amd/host -o $@ dqmc_amd/host/main.cpp -Ldqmc_amd -ldqmc_hip -ldl -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(CSRC)/*.o dqmc_amd/*.so dqmc_amd/dqmc_driver
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
