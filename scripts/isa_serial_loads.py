#!/usr/bin/env python3
"""Dependent trips to memory the compiler built: a global / scratch load followed at once (within a few instructions) by `s_waitcnt vmcnt(0)` - the
next load of the same loop or sequence cannot be in flight meanwhile.  Per function and source line (-gline-tables-only build):
    scripts/isa_serial_loads.py window_solve.hip -DAVM_TP=1 [--func eval_jac]"""
import collections, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_lines import build
from isa_mix import LLVM
args = sys.argv[1:]
func = None
if "--func" in args:
    i = args.index("--func"); func = args[i + 1]; del args[i:i + 2]
co = build(args[0], args[1:])
txt = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "-l", "--symbolize-operands", co], text=True)
cur = line = None
rows = collections.defaultdict(list)
for l in txt.split("\n"):
    m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
    if m and not re.match(r"^[0-9a-f]+ <L\d+>:", l):
        cur = m.group(1); continue
    m = re.match(r"^; (\S+):(\d+)", l)
    if m:
        line = "%s:%s" % (m.group(1).split("/")[-1], m.group(2)); continue
    m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*?)\s*//", l)
    if m and cur:
        rows[cur].append((line, m.group(1), m.group(2)))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.split("\n")
for f, nm in zip(rows, names):
    if func and func not in nm:
        continue
    r = rows[f]
    cnt = collections.Counter()
    for i, (ln, op, a) in enumerate(r):
        if op.startswith("global_load"):
            for k in range(1, 5):
                if i + k < len(r) and r[i + k][1].startswith("global_load"):
                    break
                if i + k < len(r) and r[i + k][1] == "s_waitcnt" and "vmcnt(0)" in r[i + k][2]:
                    cnt[ln] += 1; break
    if cnt:
        print("%-60s %d loads waited for one by one" % (re.sub(r"\(.*", "", nm)[:60], sum(cnt.values())))
        for ln, v in cnt.most_common(8):
            print("      %s  %d" % (ln, v))
