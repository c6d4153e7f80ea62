#!/bin/bash
# Everything the committed evidence of a round comes from, in one gpurun call (run from the repository root on the GPU box):
#   scripts/gpu_validate.sh <tag> [fuzz rounds]
# GPU test suite, randomised device-vs-oracle campaign, rocprofv3 passes of every bench workload, the bench lines, the other
# configurations, the host API rates.  Raw output under gpurun_out/<tag>/ and gpurun_out/<tag>_<workload>/; afterwards (anywhere):
# scripts/summarize_profiles.py <tag> <workload>, scripts/isa_mix.py <tag> <workload> -- then take the bench lines again (a second call of
# the last block) so that roofline.profile_stale compares them with THESE profiles.
TAG=${1:-r04}; ROUNDS=${2:-2}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/$TAG
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/$TAG/suite.log 2>&1
(timeout 4000 bash scripts/fuzz_campaign.sh 7000 $ROUNDS) > gpurun_out/$TAG/fuzz_campaign.log 2>&1
timeout 900 bash scripts/profile_all.sh $TAG headline > gpurun_out/$TAG/prof_headline.log 2>&1
timeout 600 bash scripts/profile_stream.sh $TAG headline > gpurun_out/$TAG/prof_stream.log 2>&1
timeout 900 bash scripts/profile_all.sh $TAG bb288 --steps 6 > gpurun_out/$TAG/prof_bb288.log 2>&1
timeout 900 bash scripts/profile_all.sh $TAG gdg --steps 8 > gpurun_out/$TAG/prof_gdg.log 2>&1
timeout 900 bash scripts/profile_all.sh $TAG gdg64 --steps 6 > gpurun_out/$TAG/prof_gdg64.log 2>&1
timeout 900 bash scripts/profile_all.sh $TAG bp4 --steps 10 > gpurun_out/$TAG/prof_bp4.log 2>&1
timeout 1500 bash scripts/profile_all.sh $TAG global144 --shots 2048 --steps 6 > gpurun_out/$TAG/prof_global144.log 2>&1
(timeout 900 python bench.py) > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
for wl in bb288 gdg gdg64 bp4; do (timeout 600 python bench.py --workload $wl --steps 8 --warmup 2) > gpurun_out/$TAG/bench_$wl.json 2>/dev/null; done
(timeout 600 python bench.py --workload global144 --shots 2048 --steps 10 --warmup 2) > gpurun_out/$TAG/bench_global144.json 2>/dev/null
(timeout 1500 python scripts/bench_configs.py) > gpurun_out/$TAG/other_configs.jsonl 2> gpurun_out/$TAG/other_configs.err
(timeout 900 python scripts/host_api_rate.py) > gpurun_out/$TAG/host_api_rate.json 2> gpurun_out/$TAG/host_api_rate.err
tail -3 gpurun_out/$TAG/suite.log; grep -c "0 mismatching" gpurun_out/$TAG/fuzz_campaign.log; grep -v "0 mismatching" gpurun_out/$TAG/fuzz_campaign.log | head; cut -c1-160 gpurun_out/$TAG/bench.json
