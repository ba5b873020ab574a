cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nanoflann_nn.py tests/test_fsel_solo.py tests/test_fsel_mp.py tests/test_fsel_truth.py -m gpu -x -q -s > gpurun_out/kd_tests.log 2>&1; tail -15 gpurun_out/kd_tests.log
for P in 1 16 256; do python scripts/dev_fsel_time.py $P 5; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/kd_time.txt
