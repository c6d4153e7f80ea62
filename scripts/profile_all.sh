#!/bin/bash
# All rocprofv3 passes behind profiles/<tag>_* (run on the GPU box from the repository root; raw output under gpurun_out/):
#   kernel-trace statistics of the bench command, FETCH_SIZE and WRITE_SIZE in separate --pmc passes, two SQ passes.
# Summaries: scripts/summarize_profiles.py <tag>; scripts/summarize_sq.py <tag>   (run afterwards, anywhere)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-order10"
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write $R/gpurun_out/pmc_sq1 $R/gpurun_out/pmc_sq2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 > $R/gpurun_out/prof_write.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_sq1 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 > $R/gpurun_out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq2 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 > $R/gpurun_out/pmc_sq2.log 2>&1
tail -1 $R/gpurun_out/prof_stats.log | cut -c1-200
