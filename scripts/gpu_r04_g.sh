#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04g; mkdir -p $O
(timeout 3300 bash scripts/fuzz_campaign.sh 11000 2) > $O/fuzz_campaign.log 2>&1
grep -c "0 mismatching" $O/fuzz_campaign.log; grep -v "0 mismatching" $O/fuzz_campaign.log | head
