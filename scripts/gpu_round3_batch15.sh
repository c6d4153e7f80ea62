cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 1800 python -m pytest tests/test_gpu_pipeline.py -x -q -s -k "x_basis" 2>&1 | tail -8) > gpurun_out/r03/xbasis.log 2>&1
cat gpurun_out/r03/xbasis.log
