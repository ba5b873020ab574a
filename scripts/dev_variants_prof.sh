#!/bin/bash
# per-phase profile (scripts/dev_prof.py) for every build/variants/libavm_hip_*.so, restoring the shipped library afterwards
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
P=anticipated-vins-mono_amd
cp $P/libavm_hip.so /tmp/libavm_hip_shipped.so
for f in build/variants/libavm_hip_*.so; do cp $f $P/libavm_hip.so; echo "== $(basename $f .so)"; python scripts/dev_prof.py 256 dense 2>/dev/null | grep -E "^preint|A: frames\+imu0|B: feat sums/PART|F: feature"; done
cp /tmp/libavm_hip_shipped.so $P/libavm_hip.so
