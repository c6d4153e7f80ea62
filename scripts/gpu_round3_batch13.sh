cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python scripts/phase_profile.py 4096 0 > gpurun_out/r03/phase_headline_o0.log 2>&1
python scripts/phase_profile.py 4096 10 > gpurun_out/r03/phase_headline_o10.log 2>&1
SWD_CONFIG=288 python scripts/phase_profile.py 4096 10 > gpurun_out/r03/phase_288_o10.log 2>&1
head -16 gpurun_out/r03/phase_headline_o0.log; head -14 gpurun_out/r03/phase_headline_o10.log | tail -11;  head -14 gpurun_out/r03/phase_288_o10.log | tail -11
