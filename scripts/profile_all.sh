#!/bin/bash
# All rocprofv3 passes behind profiles/<tag>_<workload>_* (run on the GPU box from the repository root; raw output under
# gpurun_out/<tag>_<workload>/):  kernel-trace statistics of the bench command, FETCH_SIZE and WRITE_SIZE in separate --pmc
# passes, four SQ passes.      usage: scripts/profile_all.sh <tag> <workload> [extra bench.py arguments]
# Summaries (afterwards, anywhere): scripts/summarize_profiles.py <tag> <workload>
TAG=${1:-r04}; WL=${2:-headline}; shift $(( $# < 2 ? $# : 2 ))
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_${WL}
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $WL --no-cpu-baseline --no-order0 --no-stream --no-other-workloads $*"  # (one launch at a time: the durations in the trace are those of single launches)
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 > $O/write.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/sq1 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 > $O/sq2.log 2>&1
# VALU instructions by type (priced by width in bench.py: fp64-rate kinds 4 cycles per wave64 instruction, the rest 2)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $O/sq3 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 > $O/sq3.log 2>&1
# LDS instructions by kind + the LDS unit's FIFO-full cycles (is the CU's LDS pipeline what the waves queue for?)
rocprofv3 --pmc SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_LDS_ATOMIC_BANDWIDTH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $O/sq4 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 > $O/sq4.log 2>&1
# keep what travels back small: the per-dispatch CSVs of the counter passes are all the summaries read
find $O -name '*.db' -delete 2>/dev/null
tail -1 $O/stats.log | cut -c1-300
