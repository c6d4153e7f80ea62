#!/bin/bash
# round 4, final evidence of this build: global144 re-profiled (its kernel changed), every bench line against the committed profiles,
# the other configurations, a fuzz campaign incl. the ensemble pipelines
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04j; mkdir -p $O
timeout 1500 bash scripts/profile_all.sh r04 global144 --shots 2048 --steps 6 > $O/prof_global144.log 2>&1
timeout 900 bash scripts/profile_all.sh r04 gdg64 --steps 6 > $O/prof_gdg64.log 2>&1
(timeout 900 python bench.py) > $O/bench.json 2> $O/bench.err
for wl in bb288 gdg gdg64 bp4; do (timeout 600 python bench.py --workload $wl --steps 8 --warmup 2) > $O/bench_$wl.json 2>/dev/null; done
(timeout 600 python bench.py --workload global144 --shots 2048 --steps 10 --warmup 2) > $O/bench_global144.json 2>/dev/null
(timeout 1500 python scripts/bench_configs.py) > $O/other_configs.jsonl 2> $O/other_configs.err
(timeout 3000 bash scripts/fuzz_campaign.sh 12000 3) > $O/fuzz_campaign.log 2>&1
for f in bench bench_bb288 bench_gdg bench_gdg64 bench_bp4 bench_global144; do cut -c1-170 $O/$f.json; done; grep -c "0 mismatching" $O/fuzz_campaign.log; grep -v "0 mismatching" $O/fuzz_campaign.log | head -5
