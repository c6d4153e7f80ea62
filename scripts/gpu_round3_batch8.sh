cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for lib in libswd_hip.so libswd_hip_devA.so libswd_hip_devB.so libswd_hip_devA.so libswd_hip_devB.so; do SWD_CONFIG=288 SWD_LIB=$lib python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03/ab_288_depth2.log
(SWD_CONFIG=288 SWD_LIB=libswd_hip_devB.so python scripts/phase_profile.py 4096 0) > gpurun_out/r03/phase_288_depth2.log 2>&1
(SWD_CONFIG=288 SWD_LIB=libswd_hip_devA.so python scripts/phase_profile.py 4096 0) > gpurun_out/r03/phase_288_base.log 2>&1
cat gpurun_out/r03/ab_288_depth2.log; head -13 gpurun_out/r03/phase_288_depth2.log; head -13 gpurun_out/r03/phase_288_base.log
