cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 1200 python -m pytest tests/test_gpu_big.py -x -q 2>&1 | tail -30) > gpurun_out/r03/big2.log 2>&1
(SWD_CONFIG=global144 timeout 600 python scripts/phase_profile.py 2048 10) > gpurun_out/r03/phase_global144_b.log 2>&1
(timeout 600 python bench.py --workload global144 --steps 5 --warmup 1 --shots 2048) > gpurun_out/r03/bench_global144_b.json 2> gpurun_out/r03/bench_global144_b.err
tail -5 gpurun_out/r03/big2.log; head -14 gpurun_out/r03/phase_global144_b.log; cut -c1-200 gpurun_out/r03/bench_global144_b.json
python scripts/ab_time.py > gpurun_out/r03/ab_renum.log 2>&1
SWD_LIB=libswd_hip_dev.so python scripts/ab_time.py >> gpurun_out/r03/ab_renum.log 2>&1
SWD_POST_RENUM=1 SWD_LIB=libswd_hip_dev.so python scripts/ab_time.py >> gpurun_out/r03/ab_renum.log 2>&1
SWD_POST_RENUM=1 SWD_LIB=libswd_hip_dev.so timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_osdw.py -x -q 2>&1 | tail -3 >> gpurun_out/r03/ab_renum.log
grep -v amdgpu.ids gpurun_out/r03/ab_renum.log
