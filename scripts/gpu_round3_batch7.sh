cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/r03/suite3.log 2>&1
timeout 1500 bash scripts/profile_all.sh r03 global144 --steps 4 --warmup 1 --shots 2048 > gpurun_out/r03/prof_global144.log 2>&1
timeout 900 bash scripts/profile_all.sh r03 headline > gpurun_out/r03/prof_headline.log 2>&1
(timeout 600 python bench.py) > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err
(timeout 600 python bench.py --workload global144 --shots 2048 --steps 10) > gpurun_out/r03/bench_global144.json 2> gpurun_out/r03/bench_global144.err
tail -4 gpurun_out/r03/suite3.log; cut -c1-200 gpurun_out/r03/bench.json
