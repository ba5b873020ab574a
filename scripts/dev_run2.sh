cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python scripts/dev_tp_grid.py 64 128 256 512 > gpurun_out/grid_shipped.txt 2>&1
for v in shipped ipra; do
  [ $v = ipra ] && cp build/variants/libavm_hip_ipra.so anticipated-vins-mono_amd/libavm_hip.so
  OUT=gpurun_out/pmc_$v; mkdir -p $OUT
  PMCCMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-fsel --gen-procs 1 --distinct 512"
  KF='--kernel-include-regex (window_solve)'
  timeout 300 rocprofv3 --kernel-trace $KF --pmc FETCH_SIZE -d $OUT/f -o run -- $PMCCMD > /dev/null 2> $OUT/f.log
  timeout 300 rocprofv3 --kernel-trace $KF --pmc WRITE_SIZE -d $OUT/w -o run -- $PMCCMD > /dev/null 2> $OUT/w.log
  timeout 300 rocprofv3 --kernel-trace $KF --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM -d $OUT/s -o run -- $PMCCMD > /dev/null 2> $OUT/s.log
  python - <<PY
import sqlite3,glob
for d in ("f","w","s"):
    for db in glob.glob("$OUT/%s/**/*.db"%d, recursive=True):
        cur=sqlite3.connect(db).cursor()
        tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        t=[x for x in tabs if x.startswith('counters_collection')]
        for r in cur.execute("select kernel_name,counter_name,count(*),avg(value),avg(duration) from %s group by kernel_name,counter_name"%t[0]):
            print("$v", r[0][:40], r[1], r[2], "%.5g"%r[3], "%.0f"%r[4])
PY
done > gpurun_out/pmc_cmp.txt 2>&1
cat gpurun_out/grid_shipped.txt gpurun_out/pmc_cmp.txt
