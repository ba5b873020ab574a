#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ (run on the GPU box through gpurun).
# Counter passes are separate runs with --kernel-trace only (no sys/hip/hsa trace domains).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=gpurun_out/prof_$1
mkdir -p $OUT
# what the numbers belong to: bench.py refuses a committed traffic figure whose kernel source has changed since
python -c "import bench; print(bench.kernel_source_sha256())" > $OUT/kernel_source_sha256.txt
# in-process input generation (forked workers under the profiler's signal handlers can hang) and, for the counter passes,
# only the window kernels (the selector's thousands of small launches serialize under --pmc)
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --fsel-problems 4 --gen-procs 1 --distinct 512"
PMCCMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-fsel --gen-procs 1 --distinct 512"
KF='--kernel-include-regex (window_solve|marginalize|prior_eig|prior_chol|preint)' 
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
timeout 300 rocprofv3 --kernel-trace $KF --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- $PMCCMD > /dev/null 2> $OUT/pmc_fetch.log
timeout 300 rocprofv3 --kernel-trace $KF --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- $PMCCMD > /dev/null 2> $OUT/pmc_write.log
timeout 300 rocprofv3 --kernel-trace $KF --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $OUT/pmc_sq -o run -- $PMCCMD > /dev/null 2> $OUT/pmc_sq.log
python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.log
python scripts/dev_prof.py 256 dense > $OUT/phase_breakdown_dense.txt 2>&1
python scripts/dev_prof.py 256 sparse > $OUT/phase_breakdown_sparse.txt 2>&1
tail -1 $OUT/bench.json | cut -c1-400
