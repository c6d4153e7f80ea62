#!/bin/bash
# Kernel trace of the bench command in its STREAMED step mode (the mode `value` is measured in): consecutive launches on the two lanes
# of swd_pipeline_stream_push_dev overlap, so the per-launch durations of this trace are durations UNDER overlap.  Run on the GPU box from
# the repository root; raw output under gpurun_out/<tag>_<workload>_stream/.   usage: scripts/profile_stream.sh <tag> <workload> [bench args]
# Summary (afterwards, anywhere): scripts/summarize_stream_trace.py <tag> <workload> -> profiles/<tag>_<workload>_stream_summary.json
TAG=${1:-r05}; WL=${2:-headline}; shift $(( $# < 2 ? $# : 2 ))
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_${WL}_stream
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-order0 --no-other-workloads --steps 60 --warmup 3 $* > $O/bench.log 2>&1
find $O -name '*.db' -delete 2>/dev/null
tail -1 $O/bench.log | cut -c1-300
