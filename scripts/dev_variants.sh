#!/bin/bash
# Times window_solve_kernel for every build/variants/libavm_hip_*.so (compiler-flag experiments), restoring the shipped library afterwards.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
P=anticipated-vins-mono_amd
cp $P/libavm_hip.so /tmp/libavm_hip_shipped.so
run() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fsel --distinct 256 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['kernel_ms'])"; }
run shipped
for f in build/variants/libavm_hip_*.so; do cp $f $P/libavm_hip.so; run $(basename $f .so); done
cp /tmp/libavm_hip_shipped.so $P/libavm_hip.so
