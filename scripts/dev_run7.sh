cd $GRAFT_REPO_ROOT
bash scripts/gpu_profile.sh r05a > gpurun_out/prof_r05a.log 2>&1
bash scripts/gpu_profile_fsel.sh r05a_fsel > gpurun_out/prof_r05a_fsel.log 2>&1
tail -3 gpurun_out/prof_r05a.log gpurun_out/prof_r05a_fsel.log
