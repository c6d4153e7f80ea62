# A/B of two builds of the library on the GPU box: slidingwindowdecoder_amd/libswd_hip_prev.so (copy of the previous build) vs libswd_hip.so
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for lib in libswd_hip_prev.so libswd_hip.so libswd_hip_prev.so libswd_hip.so; do SWD_LIB=$lib python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids; done
for lib in libswd_hip_prev.so libswd_hip.so; do SWD_ORDER=10 SWD_LIB=$lib python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids; done
for lib in libswd_hip_prev.so libswd_hip.so; do SWD_CONFIG=288 SWD_LIB=$lib python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids; done
for lib in libswd_hip_prev.so libswd_hip.so; do SWD_LIB=$lib SWD_GDG_SHOTS=16384 python scripts/bench_configs.py 3 3small 2>&1 | grep -v amdgpu | cut -c100-260; done
timeout 600 python3 tests/fuzz_vs_oracle.py 30 7000 121 300 osdw 2>&1 | grep -v amdgpu | cut -c1-150 | tail -3
timeout 600 python3 tests/fuzz_vs_oracle.py 30 7001 121 300 osdw 2>&1 | grep -v amdgpu | cut -c1-150 | tail -3
(timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -3)
