# Instruction-fetch counters of the window kernels (run on the GPU box through gpurun): is the straight-line code of these kernels
# (eval_jac 60 KB, chol_regs<0..3> 4 x 38 KB, the kernel bodies 68 / 81 KB) served by the 64 KB instruction cache?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/pmc_icache; mkdir -p $OUT
PMCCMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-fsel --gen-procs 1 --distinct 512"
KF='--kernel-include-regex (window_solve|marginalize|preint_kernel)'
timeout 300 rocprofv3 --kernel-trace $KF --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $OUT/p1 -o run -- $PMCCMD > $OUT/b.json 2> $OUT/p1.log
timeout 300 rocprofv3 --kernel-trace $KF --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQC_ICACHE_BUSY_CYCLES SQ_WAVES -d $OUT/p2 -o run -- $PMCCMD > /dev/null 2> $OUT/p2.log
python - <<'PY'
import sqlite3,glob
for d in ("p1","p2"):
    for db in glob.glob(f"gpurun_out/pmc_icache/{d}/*.db"):
        for r in sqlite3.connect(db).cursor().execute("select kernel_name,counter_name,count(*),avg(value),avg(duration) from counters_collection group by kernel_name,counter_name"):
            print(r[0][:40], r[1], r[2], "%.4g"%r[3], "%.0f"%r[4])
PY
tail -3 $OUT/p1.log $OUT/p2.log
