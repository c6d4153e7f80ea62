#!/bin/bash
# round 4: the suite, the bench lines and the host API rates on the build with the ensemble's tasks
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04e; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12) > $O/suite.log 2>&1
(timeout 600 python bench.py --no-cpu-baseline --steps 30) > $O/bench_stream.json 2> $O/bench.err
(timeout 600 python bench.py --workload gdg64 --steps 8 --warmup 2) > $O/bench_gdg64.json 2>> $O/bench.err
(timeout 600 python bench.py --workload gdg64 --steps 8 --warmup 2 --no-stream) > $O/bench_gdg64_nostream.json 2>> $O/bench.err
(timeout 600 python bench.py --workload gdg --steps 8 --warmup 2) > $O/bench_gdg.json 2>> $O/bench.err
(timeout 600 python bench.py --workload bb288 --steps 6 --warmup 2) > $O/bench_bb288.json 2>> $O/bench.err
(timeout 600 python bench.py --workload bp4 --steps 10 --warmup 2) > $O/bench_bp4.json 2>> $O/bench.err
(timeout 900 python scripts/bench_configs.py 3mt 2>/dev/null | cut -c1-60,200-330) > $O/cfg_3mt.log
(timeout 900 python scripts/host_api_rate.py) > $O/host_api_rate.json 2> $O/host_api_rate.err
tail -5 $O/suite.log; for f in bench_stream bench_gdg64 bench_gdg64_nostream bench_gdg bench_bb288 bench_bp4; do cut -c1-200 $O/$f.json; done; cat $O/cfg_3mt.log; cut -c1-1100 $O/host_api_rate.json
