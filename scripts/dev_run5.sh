cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
AVM_MARG_NOISE_REL=1e-18 timeout 600 python -m pytest tests/test_prior_parity.py -m gpu -q -s -k "cholesky_square_root" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/nr18_test.txt; cat gpurun_out/nr18_test.txt
cp anticipated-vins-mono_amd/libavm_hip.so /tmp/shipped.so; cp build/variants/libavm_hip_fstrace.so anticipated-vins-mono_amd/libavm_hip.so
for H in 10 13; do AVM_FSEL_TRACE=1 python tests/tools/fsel_single.py 3 $H 2>&1 | grep -v amdgpu.ids | tail -4; done > gpurun_out/fsel_trace.txt; cat gpurun_out/fsel_trace.txt
cp /tmp/shipped.so anticipated-vins-mono_amd/libavm_hip.so
