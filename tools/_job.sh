cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r04 > gpurun_out/r04_profile.log 2>&1
bash tools/pmc_mfma.sh r04 > gpurun_out/r04_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
tail -22 gpurun_out/r04_profile.log; tail -14 gpurun_out/r04_mfma.log
