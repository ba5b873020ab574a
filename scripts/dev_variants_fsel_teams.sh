#!/bin/bash
# Times the selector's teams form (16 / 32 frames per call) for build/variants/libavm_hip_*.so, the shipped build before and after.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
P=anticipated-vins-mono_amd
cp $P/libavm_hip.so /tmp/libavm_hip_shipped.so
run() { echo "== $1"; for n in 16 32 1; do python scripts/dev_fsel_time.py $n 6 2>&1 | grep "frames per call"; done; }
run shipped
if [ $# -gt 0 ]; then L=""; for n in "$@"; do L="$L build/variants/libavm_hip_$n.so"; done; else L=$(ls build/variants/libavm_hip_*.so); fi
for f in $L; do cp $f $P/libavm_hip.so; run $(basename $f .so); done
cp /tmp/libavm_hip_shipped.so $P/libavm_hip.so
run shipped_again
