cd $GRAFT_REPO_ROOT
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fsel --distinct 256 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: round(v,3) for k,v in j['kernel_ms'].items()}, 'ragged', j['sparse_tracks']['kernel_ms'], j['sparse_tracks']['prior_square_roots'])"
timeout 1200 python -m pytest tests/test_prior_truth.py tests/test_prior_parity.py tests/test_marg_mp.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5
