#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04c; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $O/suite.log 2>&1
(timeout 900 python scripts/bench_configs.py 3mt 2>$O/cfg_3mt.err) > $O/cfg_3mt.jsonl
(timeout 600 python bench.py --no-cpu-baseline --steps 30) > $O/bench_stream.json 2> $O/bench_stream.err
(timeout 600 python bench.py --no-cpu-baseline --steps 30 --no-stream --no-order0) > $O/bench_nostream.json 2>> $O/bench_stream.err
(timeout 600 python bench.py --workload bp4 --steps 10 --warmup 2) > $O/bench_bp4.json 2> $O/bench_bp4.err
(timeout 600 python bench.py --workload gdg64 --steps 6 --warmup 2) > $O/bench_gdg64.json 2> $O/bench_gdg64.err
(timeout 600 python bench.py --workload gdg --steps 6 --warmup 2) > $O/bench_gdg.json 2> $O/bench_gdg.err
(timeout 900 python scripts/host_api_rate.py) > $O/host_api_rate.json 2> $O/host_api_rate.err
tail -12 $O/suite.log; cut -c1-330 $O/cfg_3mt.jsonl; for f in bench_stream bench_nostream bench_bp4 bench_gdg64 bench_gdg; do cut -c1-240 $O/$f.json; done; cut -c1-900 $O/host_api_rate.json; tail -3 $O/*.err | cut -c1-300
