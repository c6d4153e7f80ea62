cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 1200 python -m pytest tests/test_gpu_big.py -x -q 2>&1 | tail -40) > gpurun_out/r03/big.log 2>&1
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/r03/suite2.log 2>&1
tail -5 gpurun_out/r03/big.log; tail -8 gpurun_out/r03/suite2.log
