#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04h; mkdir -p $O
for lib in libswd_hip_dev.so libswd_hip_osdcall.so libswd_hip_dev.so libswd_hip_osdcall.so; do SWD_ORDER=10 SWD_LIB=$lib python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids; done > $O/ab_osdcall.log 2>&1
for lib in libswd_hip_dev.so libswd_hip_osdcall.so; do SWD_LIB=$lib python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids; done >> $O/ab_osdcall.log 2>&1
(SWD_LIB=libswd_hip_osdcall.so timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -k "matches_reference_run" 2>&1 | tail -2) >> $O/ab_osdcall.log
cat $O/ab_osdcall.log
