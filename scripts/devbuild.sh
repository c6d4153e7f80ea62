#!/bin/bash
# Development build: only the [[144,12,12]] kernel variant (SWD_DEV_VARIANTS=-DSWD_BB288_ONLY: the [[288,12,18]] ones), into
# ${SWD_DEV_OUT:-libswd_hip_dev.so} (extra flags: $@); select it with SWD_LIB=<that file>
set -e
cd "$(dirname "$0")/../slidingwindowdecoder_amd/csrc"
mkdir -p build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-sched-strategy=${SWD_SCHED:-iterative-ilp} -mllvm -amdgpu-atomic-optimizer-strategy=None -I../../include -I. ${SWD_DEV_VARIANTS:--DSWD_HEADLINE_ONLY}"
for f in swd_osdw swd_kernels_k0 swd_kernels_k1 swd_kernels_k2 swd_kernels_k3 swd_kernels_k7; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o build/${f}_dev.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../${SWD_DEV_OUT:-libswd_hip_dev.so} build/swd_graph.o build/swd_osdw_dev.o build/swd_kernels_k0_dev.o build/swd_kernels_k1_dev.o build/swd_kernels_k2_dev.o build/swd_kernels_k3_dev.o build/swd_kernels_k5.o build/swd_kernels_k7_dev.o build/swd_kernels_k8.o build/swd_bp4.o build/swd_sampler.o build/swd_huge.o
