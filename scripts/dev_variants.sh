#!/bin/bash
# Times the solve kernels for every build/variants/libavm_hip_*.so (compiler-flag / macro experiments), restoring the shipped library afterwards.
#   scripts/dev_variants.sh [name ...]     (default: every variant in build/variants)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
P=anticipated-vins-mono_amd
cp $P/libavm_hip.so /tmp/libavm_hip_shipped.so
run() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fsel --no-host-latency --distinct 256 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', {k: round(v,3) for k,v in j['kernel_ms'].items()}, 'ragged', round(j['sparse_tracks']['kernel_ms']['window_solve'],3), 'x1024', round(j['extended_problem']['kernel_ms']['window_solve'],3), 'single', round(j['latency_single_window_ms']['kernel_ms']['window_solve'],3))"; }
run shipped
if [ $# -gt 0 ]; then L=""; for n in "$@"; do L="$L build/variants/libavm_hip_$n.so"; done; else L=$(ls build/variants/libavm_hip_*.so); fi
for f in $L; do cp $f $P/libavm_hip.so; run $(basename $f .so); done
cp /tmp/libavm_hip_shipped.so $P/libavm_hip.so
run shipped_again   # (the first run of a call is not always representative: the shipped build brackets the variants)
for f in $L; do cp $f $P/libavm_hip.so; run $(basename $f .so)_again; done
cp /tmp/libavm_hip_shipped.so $P/libavm_hip.so
