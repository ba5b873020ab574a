#!/bin/bash
# Per-wavefront time inside the throughput solve's factorization: the four -DAVM_PROF_CHOL=<wavefront> variants (scripts/build_variant.sh pc<w> "-DAVM_PROF_CHOL=<w>" tp),
# one window per CU (AVM_TP_GRID=256) and two.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
P=anticipated-vins-mono_amd
cp $P/libavm_hip.so /tmp/libavm_hip_shipped.so
for g in 256 512; do
  for w in ${WAVES:-0 1 2 3}; do
    cp build/variants/libavm_hip_pc$w.so $P/libavm_hip.so
    echo "grid $g wavefront $w: $(AVM_SOLVE_TP=1 AVM_TP_GRID=$g python scripts/dev_prof.py 4096 dense nomarg 2>&1 | grep -E "factorization, one|tp: factorization" | tr '\n' ' ')"
  done
done
cp /tmp/libavm_hip_shipped.so $P/libavm_hip.so
