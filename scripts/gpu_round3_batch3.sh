cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(SWD_CONFIG=global144 timeout 600 python scripts/phase_profile.py 2048 10) > gpurun_out/r03/phase_global144.log 2>&1
(timeout 600 python bench.py --workload global144 --steps 5 --warmup 1 --shots 2048) > gpurun_out/r03/bench_global144.json 2> gpurun_out/r03/bench_global144.err
timeout 1500 bash scripts/profile_all.sh r03 global144 --steps 4 --warmup 1 --shots 2048 > gpurun_out/r03/prof_global144.log 2>&1
cat gpurun_out/r03/phase_global144.log; cut -c1-400 gpurun_out/r03/bench_global144.json
