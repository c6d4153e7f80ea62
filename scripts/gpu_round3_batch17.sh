cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 1500 python -m pytest tests/test_gpu_gdg.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -4) > gpurun_out/r03/gdg_tests4.log 2>&1
for md in gdg gd ens bp; do timeout 600 python3 tests/fuzz_vs_oracle.py 40 6000 6 300 $md 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300; done >> gpurun_out/r03/gdg_tests4.log
for d in bpgdg_decoder bpgd_decoder bp_history_decoder; do timeout 600 python3 tests/fuzz_pipeline.py 20 6000 $d 90 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300; done >> gpurun_out/r03/gdg_tests4.log
SWD_LIB=libswd_hip_dev.so python scripts/gdg_phase_profile.py 8192 2>&1 | grep -v amdgpu >> gpurun_out/r03/gdg_tests4.log
(timeout 600 python bench.py --workload gdg --steps 10 --warmup 2) 2>/dev/null | cut -c1-200 >> gpurun_out/r03/gdg_tests4.log
(SWD_GDG_SHOTS=16384 timeout 600 python scripts/bench_configs.py 3 3small) 2>/dev/null | cut -c1-260 >> gpurun_out/r03/gdg_tests4.log
cat gpurun_out/r03/gdg_tests4.log
