#!/bin/bash
# diagnostic build of the update kernels with s_memtime stamps (never used for timing results)
set -e
cd "$(dirname "$0")/.."
root=$PWD
mkdir -p /tmp/dqstamp && cp -r dqmc_amd /tmp/dqstamp/ && cp -r include /tmp/dqstamp/
cd /tmp/dqstamp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDQ_SCAN_STAMPS -c dqmc_amd/csrc/update.hip -o dqmc_amd/csrc/update.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDQ_QR_STAMPS -c dqmc_amd/csrc/qr_onchip.hip -o dqmc_amd/csrc/qr_onchip.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDQ_QR_STAMPS -c dqmc_amd/csrc/qr_colown.hip -o dqmc_amd/csrc/qr_colown.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o dqmc_amd/libdqmc_hip.so dqmc_amd/csrc/*.o
mkdir -p $root/scripts/stamp_build && cp dqmc_amd/libdqmc_hip.so $root/scripts/stamp_build/libdqmc_hip_stamps.so
