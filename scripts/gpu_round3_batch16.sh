cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 1500 python -m pytest tests/test_gpu_gdg.py -x -q -s 2>&1 | tail -8) > gpurun_out/r03/gdg_tests3.log 2>&1
(timeout 900 python scripts/bench_configs.py 3mt) > gpurun_out/r03/gdg_mt.jsonl 2>&1
cat gpurun_out/r03/gdg_tests3.log; cut -c1-420 gpurun_out/r03/gdg_mt.jsonl
SWD_LIB=libswd_hip_dev.so python scripts/gdg_phase_profile.py 8192 2>&1 | grep -v amdgpu > gpurun_out/r03/gdg_phase.log; cat gpurun_out/r03/gdg_phase.log
