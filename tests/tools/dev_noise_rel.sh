#!/bin/bash
# Development (GPU box): the prior tests and the eight streams under other values of the noise test's constant (AVM_MARG_NOISE_REL replaces
# the default avm_options::marg_noise_rel = 1e-16 for callers that pass the default).  Results: profiles/r05_noise_rel.md
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for nr in 1e-16 1e-18 1e-20 1e-22 1e-24; do
  echo "=== AVM_MARG_NOISE_REL=$nr"
  AVM_MARG_NOISE_REL=$nr timeout 900 python -m pytest tests/test_prior_truth.py tests/test_prior_parity.py tests/test_marg_mp.py -m gpu -q -s 2>&1 | grep -E "^\[streams|FAILED|passed|failed" 
  AVM_MARG_NOISE_REL=$nr python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fsel 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['sparse_tracks']
print('ragged batch: prior_eig ms', round(s['kernel_ms']['prior_eig'],3), s['prior_square_roots'], 'solves/s', round(s['value']))"
done
