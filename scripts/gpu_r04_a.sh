#!/bin/bash
# round 4, first GPU call: the suite on the new build, the VALU classification microbenchmark (plain + under the per-type counters),
# the host API rates, the headline profile passes incl. the VALU-type pass
cd ${GRAFT_REPO_ROOT:-$(pwd)}
R=$(pwd); O=$R/gpurun_out/r04a; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $O/suite.log 2>&1
(timeout 300 scripts/ubench/valu_class) > $O/valu_class.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $O/valu_pmc -- $R/scripts/ubench/valu_class > $O/valu_pmc.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_FLOPS_FP64 SQ_BUSY_CYCLES --output-format csv -d $O/valu_pmc2 -- $R/scripts/ubench/valu_class > $O/valu_pmc2.log 2>&1
find $O -name '*.db' -delete 2>/dev/null
cd $R
(timeout 900 python scripts/host_api_rate.py) > $O/host_api_rate.json 2> $O/host_api_rate.err
timeout 1200 bash scripts/profile_all.sh r04 headline > $O/prof_headline.log 2>&1
tail -3 $O/suite.log; cat $O/host_api_rate.json | cut -c1-1500; tail -3 $O/host_api_rate.err; head -12 $O/valu_class.log
