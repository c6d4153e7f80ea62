cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/r03/suite4.log 2>&1
(timeout 3000 bash scripts/fuzz_campaign.sh 3000 2) > gpurun_out/r03/fuzz_campaign.log 2>&1
timeout 900 bash scripts/profile_all.sh r03 headline > gpurun_out/r03/prof_headline.log 2>&1
timeout 900 bash scripts/profile_all.sh r03 bb288 > gpurun_out/r03/prof_bb288.log 2>&1
(timeout 600 python bench.py) > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err
tail -4 gpurun_out/r03/suite4.log; cat gpurun_out/r03/fuzz_campaign.log | cut -c1-200; cut -c1-200 gpurun_out/r03/bench.json
