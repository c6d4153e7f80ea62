#!/bin/bash
# A/B of a bp4 environment switch in one gpurun call: scripts/ab_bp4_env.sh <out dir> <ENV=1 switch>   (e.g. SWD_BP4_NOSPLIT=1)
OUT=$1; SW=$2
mkdir -p $OUT
for rep in 1 2; do
  for mode in default "$SW"; do
    env $( [ "$mode" = default ] && echo SWD_DUMMY=1 || echo $mode ) timeout 300 python bench.py --workload bp4 --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$mode', round(j['value']), round(j['ms_per_step'], 3), round(j['roofline']['avg_kernel_ms'], 3), j['config'].get('exit_classes_bp_osd_rank0'))" >> $OUT/ab.log
  done
done
echo default >> $OUT/ab.log; timeout 300 python scripts/bp4_codes_rate.py 2>/dev/null | cut -c1-200 >> $OUT/ab.log
echo $SW >> $OUT/ab.log; env $SW timeout 300 python scripts/bp4_codes_rate.py 2>/dev/null | cut -c1-200 >> $OUT/ab.log
timeout 600 python -m pytest tests/test_gpu_bp4.py tests/test_gpu_shyps.py -x -q 2>&1 | tail -3 >> $OUT/ab.log
for s in 51 52; do timeout 600 python3 tests/fuzz_bp4.py 40 $s 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300 >> $OUT/ab.log; done
cat $OUT/ab.log
