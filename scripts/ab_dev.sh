#!/bin/bash
# A/B of development builds (scripts/devbuild.sh) in one gpurun call: scripts/ab_dev.sh <out dir> <lib suffixes ...>; every build is timed
# twice at OSD order 10 (the headline), interleaved; the first suffix also runs the recorded-run parity tests
OUT=$1; shift
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do for v in "$@"; do SWD_ORDER=10 SWD_LIB=libswd_hip_$v.so python scripts/ab_time.py 2>&1 | grep -v amdgpu.ids >> $OUT/ab_dev.log; done; done
SWD_LIB=libswd_hip_$1.so timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_osdw.py -x -q -k "bb144 or bb72" 2>&1 | tail -3 >> $OUT/ab_dev.log
cat $OUT/ab_dev.log
