cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 1200 python -m pytest tests/test_gpu_big.py -x -q 2>&1 | tail -5) > gpurun_out/r03/big3.log 2>&1
(SWD_CONFIG=global144 timeout 600 python scripts/phase_profile.py 2048 10) > gpurun_out/r03/phase_global144_c.log 2>&1
(timeout 600 python bench.py --workload global144 --steps 5 --warmup 1 --shots 2048) > gpurun_out/r03/bench_global144_c.json 2> gpurun_out/r03/bench_global144_c.err
tail -3 gpurun_out/r03/big3.log; head -14 gpurun_out/r03/phase_global144_c.log; cut -c1-200 gpurun_out/r03/bench_global144_c.json
