cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/r03/suite.log 2>&1
for wl in headline bb288 gdg bp4; do timeout 900 bash scripts/profile_all.sh r03 $wl > gpurun_out/r03/prof_$wl.log 2>&1; done
timeout 600 bash scripts/lds_conflict_attribution.sh r03 > gpurun_out/r03/conflicts.log 2>&1
(cd $GRAFT_REPO_ROOT && timeout 600 python bench.py > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err)
tail -3 gpurun_out/r03/suite.log; cat gpurun_out/r03/prof_*.log | cut -c1-250; du -sh gpurun_out
