#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04d; mkdir -p $O
rm -f $O/modes.log
(SWD_LIB=libswd_hip_dev.so timeout 600 python -m pytest tests/test_gpu_gdg.py -q -x -k "ensemble or bb144_gdg" 2>&1 | tail -4) > $O/suite_q.log
for seed in 7 1; do
for mode in tasks notasks tickets; do
  E="X=1"; [ $mode = notasks ] && E="SWD_ENS_NO_TASKS=1"; [ $mode = tickets ] && E="SWD_ENS_TICKETS=1"
  (env $E SWD_CFG_SEED=$seed SWD_LIB=libswd_hip_dev.so timeout 600 python scripts/bench_configs.py 3mt 2>/dev/null | python -c "
import sys, json
r=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('seed $seed', '$mode', ' '.join('%.1f ms' % x['ms_per_launch'] for x in r))") >> $O/modes.log
done; done
cat $O/suite_q.log $O/modes.log
