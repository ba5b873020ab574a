cd $GRAFT_REPO_ROOT
timeout 1400 python -m pytest tests -m gpu -q > gpurun_out/gputest_r5c.log 2>&1; tail -4 gpurun_out/gputest_r5c.log
bash scripts/gpu_profile.sh r05b > gpurun_out/prof_r05b.log 2>&1
bash scripts/gpu_profile_fsel.sh r05b_fsel > gpurun_out/prof_r05b_fsel.log 2>&1
tail -c 600 gpurun_out/prof_r05b.log
