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
out.append("## rocprofv3 --kernel-trace --stats (durations in us)\n")
out.append("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
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
open(dst + ".md", "w").write("\n".join(out) + "\n")
print("wrote", dst + ".md")
