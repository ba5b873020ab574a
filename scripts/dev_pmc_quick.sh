cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/pmc_tp1; mkdir -p $OUT
PMCCMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-fsel --gen-procs 1 --distinct 512"
KF='--kernel-include-regex (window_solve)'
timeout 300 rocprofv3 --kernel-trace $KF --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $OUT/pmc_sq -o run -- $PMCCMD > $OUT/b.json 2> $OUT/pmc_sq.log
timeout 300 rocprofv3 --kernel-trace $KF --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES -d $OUT/pmc_sq2 -o run -- $PMCCMD > /dev/null 2> $OUT/pmc_sq2.log
python - <<'PY'
import sqlite3,glob
for d in ("pmc_sq","pmc_sq2"):
    for db in glob.glob(f"gpurun_out/pmc_tp1/{d}/*.db"):
        for r in sqlite3.connect(db).cursor().execute("select kernel_name,counter_name,count(*),avg(value),avg(duration) from counters_collection group by kernel_name,counter_name"):
            print(r[0][:40], r[1], r[2], "%.4g"%r[3], "%.0f"%r[4])
PY
tail -3 $OUT/pmc_sq2.log
