cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for lib in libswd_hip.so libswd_hip_dev.so libswd_hip.so libswd_hip_dev.so; do SWD_LIB=$lib python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03/ab_depth2_headline.log
SWD_CONFIG=288 python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03/ab_depth2_headline.log
(timeout 900 python -m pytest tests/test_gpu_osdw.py tests/test_gpu_pipeline.py tests/test_gpu_shyps.py tests/test_gpu_edges.py -x -q 2>&1 | tail -3) >> gpurun_out/r03/ab_depth2_headline.log
cat gpurun_out/r03/ab_depth2_headline.log
