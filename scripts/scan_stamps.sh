#!/bin/bash
# diagnostic builds with s_memtime stamps (never used for timing results): libraries under scripts/stamp_build/, read by
# scripts/sm_stamps.py (update kernels) and scripts/qr_stamps.py (column-owner QRCP)
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/stamp_build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
L="-L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib"
/opt/rocm/bin/hipcc $F -DDQ_SCAN_STAMPS -c dqmc_amd/csrc/update.hip -o scripts/stamp_build/update.o
/opt/rocm/bin/hipcc $F -DDQ_SM_STAMPS -c dqmc_amd/csrc/update_sm.hip -o scripts/stamp_build/update_sm.o
/opt/rocm/bin/hipcc $F -DDQ_QR_STAMPS -c dqmc_amd/csrc/qr_colown.hip -o scripts/stamp_build/qr_colown.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/stamp_build/libdqmc_hip_scan.so $(ls dqmc_amd/csrc/*.o | grep -v "csrc/update.o") scripts/stamp_build/update.o $L
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/stamp_build/libdqmc_hip_sm.so $(ls dqmc_amd/csrc/*.o | grep -v "csrc/update_sm.o") scripts/stamp_build/update_sm.o $L
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/stamp_build/libdqmc_hip_qr.so $(ls dqmc_amd/csrc/*.o | grep -v "csrc/qr_colown.o") scripts/stamp_build/qr_colown.o $L
