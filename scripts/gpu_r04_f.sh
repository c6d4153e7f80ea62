#!/bin/bash
# round 4 evidence: rocprofv3 passes of every bench workload (kernel-trace statistics, FETCH / WRITE, four SQ passes)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04f; mkdir -p $O
timeout 900 bash scripts/profile_all.sh r04 headline > $O/prof_headline.log 2>&1
timeout 900 bash scripts/profile_all.sh r04 bb288 --steps 6 > $O/prof_bb288.log 2>&1
timeout 900 bash scripts/profile_all.sh r04 gdg --steps 8 > $O/prof_gdg.log 2>&1
timeout 900 bash scripts/profile_all.sh r04 gdg64 --steps 6 > $O/prof_gdg64.log 2>&1
timeout 900 bash scripts/profile_all.sh r04 bp4 --steps 10 > $O/prof_bp4.log 2>&1
timeout 1500 bash scripts/profile_all.sh r04 global144 --shots 2048 --steps 6 > $O/prof_global144.log 2>&1
tail -2 $O/prof_*.log | cut -c1-200
