#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for wl in headline gdg gdg64; do bash scripts/icache_counters.sh $wl --steps 3; done 2>&1 | grep -v amdgpu | tail -6
