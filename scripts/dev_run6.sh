cd $GRAFT_REPO_ROOT
for nr in 1e-16 1e-18; do echo "== $nr"; AVM_MARG_NOISE_REL=$nr timeout 900 python -m pytest tests/test_prior_truth.py -m gpu -q -s -k eight 2>&1 | grep -E "^\[streams|passed|failed"; done
