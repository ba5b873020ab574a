#!/bin/bash
# rocprofv3 evidence for the selector's batch forms (run on the GPU box through gpurun): scripts/gpu_profile_fsel.sh <tag>
# 256 frames per call: fsel_solo_kernel (the default there); the same batch forced onto the teams (fsel_frame_kernel_mf) beside it.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=gpurun_out/prof_$1
mkdir -p $OUT
CMD="python scripts/dev_fsel_time.py 256 3"
KF='--kernel-include-regex (fsel_solo|fsel_frame|fsel_setup)'
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- $CMD > $OUT/solo.txt 2> $OUT/trace.log
timeout 300 rocprofv3 --kernel-trace $KF --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- $CMD > /dev/null 2> $OUT/pmc_fetch.log
timeout 300 rocprofv3 --kernel-trace $KF --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- $CMD > /dev/null 2> $OUT/pmc_write.log
timeout 300 rocprofv3 --kernel-trace $KF --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $OUT/pmc_sq -o run -- $CMD > /dev/null 2> $OUT/pmc_sq.log
AVM_FSEL_SOLO=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_teams -o run -- $CMD > $OUT/teams.txt 2> $OUT/trace_teams.log
tail -1 $OUT/solo.txt; tail -1 $OUT/teams.txt
