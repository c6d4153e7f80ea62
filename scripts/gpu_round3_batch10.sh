cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for lib in libswd_hip.so libswd_hip_dev.so libswd_hip_devA.so libswd_hip.so libswd_hip_dev.so libswd_hip_devA.so; do SWD_LIB=$lib python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03/ab_kgp.log
(SWD_LIB=libswd_hip_dev.so timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "bb144" 2>&1 | tail -2) >> gpurun_out/r03/ab_kgp.log
cat gpurun_out/r03/ab_kgp.log
