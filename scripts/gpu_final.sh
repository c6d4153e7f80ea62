#!/bin/bash
# Round-end evidence in one gpurun call after scripts/gpu_validate.sh has produced the profiles: GPU suite, the bench lines against the
# committed profiles (profile_stale must be false), the other configurations.   scripts/gpu_final.sh <tag>
TAG=${1:-r06}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/${TAG}_final
O=gpurun_out/${TAG}_final
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/suite.log 2>&1
timeout 900 bash scripts/profile_all.sh $TAG bp4 --steps 10 > $O/prof_bp4.log 2>&1
(timeout 900 python bench.py) > $O/bench.json 2> $O/bench.err
for wl in bb288 gdg gdg64 bp4; do (timeout 600 python bench.py --workload $wl --steps 8 --warmup 2) > $O/bench_$wl.json 2>/dev/null; done
(timeout 600 python bench.py --workload global144 --shots 2048 --steps 10 --warmup 2) > $O/bench_global144.json 2>/dev/null
(timeout 900 python scripts/bench_configs.py) > $O/other_configs.jsonl 2> $O/other_configs.err
(timeout 600 python scripts/huge_rate.py 512 0.003) > $O/huge_rate.json 2>/dev/null
tail -3 $O/suite.log; cut -c1-200 $O/bench.json
