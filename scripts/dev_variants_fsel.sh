#!/bin/bash
# Times the selector (256 / 64 / 16 frames per call, solo-form phase clocks of frame 0) for build/variants/libavm_hip_*.so, restoring the shipped library afterwards.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
P=anticipated-vins-mono_amd
cp $P/libavm_hip.so /tmp/libavm_hip_shipped.so
run() { echo "== $1"; for n in 256 64; do python scripts/dev_fsel_time.py $n 4 2>&1 | grep "frames per call"; done; AVM_FSEL_LAZY_STATS=1 python scripts/dev_fsel_time.py 256 1 2>&1 | grep "fsel solo kernel" | tail -1 | cut -c1-330; }
run shipped
if [ $# -gt 0 ]; then L=""; for n in "$@"; do L="$L build/variants/libavm_hip_$n.so"; done; else L=$(ls build/variants/libavm_hip_*.so); fi
for f in $L; do cp $f $P/libavm_hip.so; run $(basename $f .so); done
cp /tmp/libavm_hip_shipped.so $P/libavm_hip.so
