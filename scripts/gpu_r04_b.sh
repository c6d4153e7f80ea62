#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04b; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -x -k "packed" 2>&1 | tail -40) > $O/packed.log 2>&1
for pct in 100 67 34; do SWD_GRID_PCT=$pct SWD_LIB=libswd_hip_512.so python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids; done > $O/grid512.log 2>&1
for pct in 100 67 34; do SWD_GRID_PCT=$pct python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids; done >> $O/grid512.log 2>&1
(SWD_LIB=libswd_hip_512.so python scripts/phase_profile.py 4096 0 2>&1 | grep -v amdgpu.ids | head -16) > $O/phase512.log 2>&1
(python scripts/phase_profile.py 4096 0 2>&1 | grep -v amdgpu.ids | head -16) > $O/phase256.log 2>&1
cat $O/packed.log; cat $O/grid512.log; cat $O/phase512.log $O/phase256.log
