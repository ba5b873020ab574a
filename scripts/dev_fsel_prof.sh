cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/fs
for P in 16 1; do
rm -rf gpurun_out/fs/trace$P
rocprofv3 --kernel-trace --stats -d gpurun_out/fs/trace$P -o run -- python scripts/dev_fsel_time.py $P 3 2> /dev/null | tail -1
python - <<PY
import sqlite3
db="gpurun_out/fs/trace$P/run_results.db"
for r in sqlite3.connect(db).cursor().execute("select name,total_calls,total_duration,average from top_kernels").fetchall()[:4]:
    print("   ", r[0][:60], r[1], round(r[2]/1e3,2), "ms total", round(r[3],2), "us avg")
PY
done
