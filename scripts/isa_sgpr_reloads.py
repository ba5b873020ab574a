#!/usr/bin/env python3
"""Where a kernel reloads a SPILLED SGPR through scratch memory: `scratch_load_dword vN` followed by `v_readlane_b32 sX, vN` - the register that holds
spilled SGPRs was spilled itself, so the predicate / address that needs the SGPR starts with a trip to memory.  Per source line (20-line buckets) of one
kernel, from a -gline-tables-only build:   scripts/isa_sgpr_reloads.py window_solve.hip marginalize_tp_kernel -DAVM_TP=1"""
import collections, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_lines import build
from isa_mix import LLVM
src, kernel, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
co = build(src, defs)
txt = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "-l", "--symbolize-operands", co], text=True)
cur = line = None
rows = []
for l in txt.split("\n"):
    m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
    if m and not re.match(r"^[0-9a-f]+ <L\d+>:", l):
        cur = m.group(1); continue
    m = re.match(r"^; (\S+):(\d+)", l)
    if m:
        line = (m.group(1).split("/")[-1], int(m.group(2))); continue
    m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*?)\s*//", l)
    if m and cur and kernel in cur:
        rows.append((line, m.group(1), m.group(2)))
cnt = collections.Counter()
for i, (ln, op, args) in enumerate(rows):
    m = re.match(r"(v\d+), off", args) if op == "scratch_load_dword" else None
    if m:
        for k in range(1, 6):
            if i + k < len(rows) and rows[i + k][1] == "v_readlane_b32" and (", " + m.group(1) + ",") in (", " + rows[i + k][2]):
                cnt[(ln[0], ln[1] // 20 * 20)] += 1; break
print(kernel, "instructions", len(rows), "scratch loads", sum(1 for r in rows if r[1].startswith("scratch_load")), "SGPR reloads through scratch", sum(cnt.values()))
for k, v in sorted(cnt.items()):
    print("  %s:%d.. %d" % (k[0], k[1], v))
