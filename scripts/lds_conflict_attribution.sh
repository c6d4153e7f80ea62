#!/bin/bash
# Which phase owns the LDS bank conflicts of the headline kernel?  Counter passes of bench.py with the post phase capped at 4
# iterations instead of 200 (and, third run, the pre phase at 4 instead of 8): differences of the per-launch counters belong to
# the iterations that were removed.  Raw output: gpurun_out/<tag>_conflicts/{full,post4,pre4post4}; summary: scripts/summarize_conflicts.py
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_conflicts
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
ARGS="--no-cpu-baseline --no-order0 --steps 3 --warmup 1 --osd-order 0"
rocprofv3 --pmc $PMC --output-format csv -d $O/full -- python3 $R/bench.py $ARGS > $O/full.log 2>&1
export SWD_BENCH_POST_MAX_ITER=4
rocprofv3 --pmc $PMC --output-format csv -d $O/post4 -- python3 $R/bench.py $ARGS > $O/post4.log 2>&1
export SWD_BENCH_PRE_MAX_ITER=4
rocprofv3 --pmc $PMC --output-format csv -d $O/pre4post4 -- python3 $R/bench.py $ARGS > $O/pre4post4.log 2>&1
find $O -name '*.db' -delete 2>/dev/null
tail -qn1 $O/full.log $O/post4.log $O/pre4post4.log | cut -c1-200
