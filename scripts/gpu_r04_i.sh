#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04i; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/suite.log 2>&1
tail -8 $O/suite.log
