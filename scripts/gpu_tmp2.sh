#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04t; mkdir -p $O
(timeout 900 python -m pytest tests -x -q -m gpu -k "bp4 or camel or cabi or smoke" 2>&1 | tail -3) > $O/pytest_bp4.log
(for sd in 81 82 83; do timeout 600 python3 tests/fuzz_bp4.py 60 $sd 2>&1 | grep -v amdgpu | tail -1 | cut -c1-300; done) > $O/fuzz_bp4.log
(timeout 900 python scripts/bp4_codes_rate.py 2>&1 | grep -v amdgpu) > $O/bp4_codes.log
(SWD_BP4_NT=256 timeout 900 python scripts/bp4_codes_rate.py 2>&1 | grep -v amdgpu) > $O/bp4_codes_nt256.log
(timeout 300 python bench.py --workload bp4 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-260) > $O/bp4_bench.json
(timeout 600 python scripts/bench_configs.py bp4 bp4shyps 2>&1 | grep -v amdgpu | cut -c1-400) > $O/bp4_host.jsonl
cat $O/pytest_bp4.log $O/fuzz_bp4.log; cut -c1-130 $O/bp4_codes.log; echo nt256; cut -c1-130 $O/bp4_codes_nt256.log; cat $O/bp4_bench.json $O/bp4_host.jsonl
