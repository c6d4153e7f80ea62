#!/bin/bash
# Long randomised device-vs-oracle campaign (GPU box): scripts/fuzz_campaign.sh [seed0] [rounds]
# Every line of output is one run's verdict; any mismatch is printed by the fuzz scripts themselves.
s0=${1:-100}; rounds=${2:-3}
for ((i = 0; i < rounds; i++)); do
  sd=$((s0 + i))
  timeout 600 python3 tests/fuzz_vs_oracle.py 60 $sd 6 120 osdw 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400
  timeout 600 python3 tests/fuzz_vs_oracle.py 30 $sd 121 300 osdw 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400
  timeout 600 python3 tests/fuzz_vs_oracle.py 8 $sd 300 700 osdw 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400
  timeout 600 python3 tests/fuzz_vs_oracle.py 24 $sd 260 576 osdw 2100 3000 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400   # column-form OSD elimination
  timeout 900 python3 tests/fuzz_vs_oracle.py 8 $sd 500 1024 osdw 8300 9216 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400   # large-graph kernels (scratch region in HBM)
  timeout 600 python3 tests/fuzz_vs_oracle.py 30 $sd 6 300 ens 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400               # threaded ensemble (kernel kind 7)
  for md in gd gdg bp; do timeout 600 python3 tests/fuzz_vs_oracle.py 40 $sd 6 300 $md 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400; done
  for d in osd_window bpgdg_decoder bpgd_decoder bp_history_decoder ens; do timeout 600 python3 tests/fuzz_pipeline.py 20 $sd $d 90 2>&1 | grep -v amdgpu | tail -3 | cut -c1-500; done
  timeout 600 python3 tests/fuzz_bp4.py 40 $sd 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400
  timeout 900 python3 tests/fuzz_bp4.py 8 $sd 500 1500 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400   # workgroups of 8 .. 16 waves, several qubits per thread beyond 1024
done
