#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04s; mkdir -p $O
(for sd in 15000 15003 15007 15011 15015 15019 15023; do
  timeout 600 python tests/fuzz_pipeline.py 12 $sd ens 170 2>&1 | grep -v amdgpu | grep "MISMATCH\|mismatching\|rejected" | cut -c1-400
  timeout 600 python tests/fuzz_pipeline.py 20 $sd bpgdg_decoder 170 2>&1 | grep -v amdgpu | grep "MISMATCH\|mismatching\|rejected" | cut -c1-400
done) > $O/ens_campaign_170.log 2>&1
cat $O/ens_campaign_170.log
bash scripts/gpu_tmp2.sh
