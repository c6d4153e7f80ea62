cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 1500 python -m pytest tests/test_gpu_gdg.py -x -q -s 2>&1 | tail -40) > gpurun_out/r03/gdg_tests.log 2>&1
(timeout 600 python bench.py --workload gdg --steps 10 --warmup 2) > gpurun_out/r03/bench_gdg.json 2> gpurun_out/r03/bench_gdg.err
(SWD_GDG_SHOTS=16384 timeout 600 python scripts/bench_configs.py 3 3small 3ens) > gpurun_out/r03/gdg_configs.jsonl 2>&1
tail -12 gpurun_out/r03/gdg_tests.log; cut -c1-300 gpurun_out/r03/bench_gdg.json; cut -c1-400 gpurun_out/r03/gdg_configs.jsonl
