cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 1800 python -m pytest tests/test_gpu_osdw.py tests/test_gpu_pipeline.py tests/test_gpu_big.py tests/test_gpu_bp4.py tests/test_gpu_shyps.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5) > gpurun_out/r03/sweep_tests.log 2>&1
timeout 600 python3 tests/fuzz_vs_oracle.py 60 5000 6 120 osdw 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300 >> gpurun_out/r03/sweep_tests.log
timeout 600 python3 tests/fuzz_vs_oracle.py 24 5000 260 576 osdw 2100 3000 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300 >> gpurun_out/r03/sweep_tests.log
timeout 600 python3 tests/fuzz_pipeline.py 20 5000 osd_window 90 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300 >> gpurun_out/r03/sweep_tests.log
timeout 600 python3 tests/fuzz_bp4.py 40 5000 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300 >> gpurun_out/r03/sweep_tests.log
SWD_ORDER=10 python scripts/ab_time.py 2>&1 | grep -v amdgpu >> gpurun_out/r03/sweep_tests.log
SWD_ORDER=10 SWD_CONFIG=288 python scripts/ab_time.py 2>&1 | grep -v amdgpu >> gpurun_out/r03/sweep_tests.log
SWD_CONFIG=288 python scripts/phase_profile.py 4096 10 2>&1 | grep "osd_sweep" >> gpurun_out/r03/sweep_tests.log
python scripts/phase_profile.py 4096 10 2>&1 | grep "osd_sweep" >> gpurun_out/r03/sweep_tests.log
cat gpurun_out/r03/sweep_tests.log
