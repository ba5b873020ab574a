cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1400 python -m pytest tests/test_sweeps.py tests/test_prior_truth.py tests/test_prior_parity.py tests/test_marg_mp.py tests/test_gpu_parity.py tests/test_host_cpp.py tests/test_extended_solve.py -m gpu -q -s -k "sweep or prior or marg or chained or roll or stream or host or extended or truth" > gpurun_out/lit_tests.log 2>&1; tail -30 gpurun_out/lit_tests.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_lit.json 2> gpurun_out/bench_lit.err; tail -c 1500 gpurun_out/bench_lit.json
