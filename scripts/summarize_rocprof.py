"""Turn the rocprofv3 (rocpd sqlite) outputs of scripts/gpu_profile.sh into committed summaries.
usage: python scripts/summarize_rocprof.py gpurun_out/prof_<tag> profiles/<tag>"""
import json
import os
import sqlite3
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
out = []


def q(db, sql):
    return sqlite3.connect(db).cursor().execute(sql).fetchall()


tr = os.path.join(src, "trace", "run_results.db")
out.append("## rocprofv3 --kernel-trace --stats (durations in ms; the rocpd top_kernels view reports microseconds)\n")
out.append("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
for name, calls, tot, avg, pct in q(tr, "select name,total_calls,total_duration,average,percentage from top_kernels"):
    out.append(f"| `{name[:90]}` | {calls} | {tot/1e3:.1f} | {avg/1e3:.2f} | {pct:.1f} |")
out.append("")
for tag, title in (("pmc_fetch", "FETCH_SIZE (KB per dispatch as reported; gfx950 wide-coalesced reads are under-reported 2x, MI355X_MICROARCH.md §HBM)"),
                   ("pmc_write", "WRITE_SIZE (KB per dispatch as reported)"), ("pmc_sq", "SQ counters (per dispatch averages)")):
    db = os.path.join(src, tag, "run_results.db")
    if not os.path.exists(db):
        continue
    out.append(f"## --pmc pass: {title}\n")
    rows = q(db, "select kernel_name,counter_name,count(*),avg(value),avg(duration) from counters_collection group by kernel_name,counter_name order by kernel_name,counter_name")
    out.append("| kernel | counter | dispatches | avg value | avg dispatch ns |\n|---|---|---|---|---|")
    for kn, cn, n, v, d in rows:
        if kn.startswith("__amd") or "at::native" in kn or "elementwise" in kn:
            continue
        out.append(f"| `{kn[:70]}` | {cn} | {n} | {v:.4g} | {d:.0f} |")
    out.append("")
for f in ("bench_under_rocprof.json", "bench.json"):
    p = os.path.join(src, f)
    if os.path.exists(p):
        try:
            line = [l for l in open(p) if l.startswith("{")][-1]
            out.append(f"## {f}\n\n```json\n{json.dumps(json.loads(line), indent=1)}\n```\n")
        except Exception as e:  # noqa
            out.append(f"## {f}: unreadable ({e})\n")

# ---- per-kernel fabric traffic (bench.py's roofline.traffic reads this file)
def pmc_avg(tag, counter):
    db = os.path.join(src, tag, "run_results.db")
    if not os.path.exists(db):
        return {}
    return {kn: (n, v) for kn, n, v in q(db, f"select kernel_name,count(*),avg(value) from counters_collection where counter_name='{counter}' group by kernel_name")}


fetch, write = pmc_avg("pmc_fetch", "FETCH_SIZE"), pmc_avg("pmc_write", "WRITE_SIZE")
traffic = {}
for kn in fetch:
    short = kn.split("(")[0].split("::")[-1]
    if not short.endswith("_kernel") or kn not in write:
        continue
    f_kb, w_kb = fetch[kn][1], write[kn][1]
    traffic[short] = {
        "FETCH_SIZE_KB_reported": f_kb,
        "WRITE_SIZE_KB_reported": w_kb,
        "dispatches": fetch[kn][0],
        "traffic_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 reports half of wide coalesced reads); WRITE_SIZE as reported; "
                "counts L2<->fabric requests incl. Infinity-Cache hits",
    }
sq = {}
for cn in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA"):
    db = os.path.join(src, "pmc_sq", "run_results.db")
    if os.path.exists(db):
        for kn, v, d in q(db, f"select kernel_name,avg(value),avg(duration) from counters_collection where counter_name='{cn}' group by kernel_name"):
            short = kn.split("(")[0].split("::")[-1]
            if short.endswith("_kernel"):
                sq.setdefault(short, {})[cn] = v
                sq[short]["dispatch_ns"] = d
for short, c in sq.items():
    if short in traffic and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # MI355X: 256 CUs x 4 SIMDs at 2.4 GHz (MI355X_MICROARCH.md); SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs
        simd_cycles = c["dispatch_ns"] * 2.4 * 256 * 4
        traffic[short]["sq"] = c
        traffic[short]["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
        traffic[short]["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        traffic[short]["fabric_GBs"] = traffic[short]["traffic_bytes_per_launch"] / c["dispatch_ns"]
hp = os.path.join(src, "kernel_source_sha256.txt")
if traffic and os.path.exists(hp):
    traffic["_kernel_source_sha256"] = open(hp).read().strip()
if traffic:
    traffic["_command"] = "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --fsel-problems 4  (4096 windows, dense tracks, MARGIN_OLD)"
    json.dump(traffic, open(dst + "_pmc_traffic.json", "w"), indent=1)
    print("wrote", dst + "_pmc_traffic.json")

open(dst + ".md", "w").write("\n".join(out) + "\n")
print("wrote", dst + ".md")
