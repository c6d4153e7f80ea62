#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04d; mkdir -p $O
rm -f $O/modes2.log
(SWD_LIB=libswd_hip_dev.so timeout 600 python -m pytest tests/test_gpu_gdg.py -q -x -k "ensemble" 2>&1 | tail -3) > $O/suite_q2.log
for seed in 7 1 3; do
for lib in libswd_hip.so libswd_hip_dev.so; do
  (SWD_CFG_SEED=$seed SWD_LIB=$lib timeout 600 python scripts/bench_configs.py 3mt 2>/dev/null | python -c "
import sys, json
r=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('seed $seed', '$lib', ' '.join('%.1f ms' % x['ms_per_launch'] for x in r))") >> $O/modes2.log
done; done
cat $O/suite_q2.log $O/modes2.log
