cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 1500 python -m pytest tests/test_gpu_gdg.py tests/test_gpu_pipeline.py tests/test_gpu_scheduler.py -x -q 2>&1 | tail -5) > gpurun_out/r03/gdg_tests2.log 2>&1
for md in gdg gd ens; do timeout 600 python3 tests/fuzz_vs_oracle.py 40 4000 6 300 $md 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300; done >> gpurun_out/r03/gdg_tests2.log
for d in bpgdg_decoder bpgd_decoder; do timeout 600 python3 tests/fuzz_pipeline.py 20 4000 $d 90 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300; done >> gpurun_out/r03/gdg_tests2.log
(timeout 600 python bench.py --workload gdg --steps 10 --warmup 2) > gpurun_out/r03/bench_gdg2.json 2> gpurun_out/r03/bench_gdg2.err
(SWD_GDG_NO_DEPTH2=1 timeout 600 python bench.py --workload gdg --steps 10 --warmup 2) > gpurun_out/r03/bench_gdg2_nod2.json 2>/dev/null
(SWD_GDG_SHOTS=16384 timeout 600 python scripts/bench_configs.py 3 3small 3ens) > gpurun_out/r03/gdg_configs2.jsonl 2>&1
cat gpurun_out/r03/gdg_tests2.log; cut -c1-260 gpurun_out/r03/bench_gdg2.json; echo; cut -c1-260 gpurun_out/r03/bench_gdg2_nod2.json; echo; cut -c1-330 gpurun_out/r03/gdg_configs2.jsonl
