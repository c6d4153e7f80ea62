#!/bin/bash
# A/B of bp4 builds in one gpurun call: scripts/ab_bp4.sh <out dir> <lib suffixes...>   (libswd_hip_<suffix>.so next to the library)
OUT=$1; shift
mkdir -p $OUT
for v in "$@"; do
  for rep in 1 2; do
    SWD_LIB=libswd_hip_$v.so timeout 300 python bench.py --workload bp4 --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$v', round(j['value']), round(j['ms_per_step'], 3), round(j['roofline']['avg_kernel_ms'], 3), j['config'].get('exit_classes_bp_osd_rank0'))" >> $OUT/ab_bp4.log
  done
done
SWD_LIB=libswd_hip_$1.so timeout 600 python -m pytest tests/test_gpu_bp4.py -x -q 2>&1 | tail -3 >> $OUT/ab_bp4.log
cat $OUT/ab_bp4.log
