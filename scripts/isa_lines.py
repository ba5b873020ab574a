#!/usr/bin/env python3
"""Static instruction counts per SOURCE LINE of one function (companion of isa_mix.py): the -gline-tables-only build of a csrc file,
disassembled with line info.  scripts/isa_lines.py window_solve.hip -DAVM_TP=1 --func eval_jac [--range 950:1200] [--top 40]"""
import argparse, collections, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_mix import classify, CSRC, LLVM

def build(src, defs):
    tmp = tempfile.mkdtemp()
    flags = ["-O3", "-std=c++17", "-fconstexpr-steps=16000000", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-pass-failed", "-gline-tables-only"]
    if os.path.basename(src).startswith("window_solve"):
        flags += ["-mllvm", "-sink-insts-to-avoid-spills"]
        if not any(d.startswith("-DAVM_X") for d in defs):
            flags += ["-mllvm", "-enable-ipra", "-fno-optimize-sibling-calls"]
        else:
            flags += ["-mllvm", "-amdgpu-prealloc-sgpr-spill-vgprs"]
    b, co = tmp + "/k.bundle", tmp + "/k.co"
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + defs + ["--cuda-device-only", "-c", src, "-o", b], cwd=CSRC)
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + b, "--output=" + co])
    return co

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src", nargs="?"); ap.add_argument("--co"); ap.add_argument("--func", required=True)
    ap.add_argument("--range"); ap.add_argument("--top", type=int, default=0); ap.add_argument("--file", default=None)
    args, defs = ap.parse_known_args()
    co = args.co or build(args.src, defs)
    txt = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "-l", "--symbolize-operands", co], text=True)
    cur, line = None, None
    cnt = collections.defaultdict(collections.Counter)
    for l in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
        if m and not re.match(r"^[0-9a-f]+ <L\d+>:", l):
            cur = m.group(1); continue
        m = re.match(r"^; (\S+):(\d+)", l)
        if m:
            line = (m.group(1).split("/")[-1], int(m.group(2))); continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\b", l)
        if m and cur and args.func in cur:
            cnt[line][classify(m.group(1))] += 1
    rows = sorted(cnt.items())
    if args.range:
        lo, hi = map(int, args.range.split(":"))
        rows = [r for r in rows if lo <= r[0][1] <= hi and (args.file is None or r[0][0] == args.file)]
    if args.top:
        rows = sorted(rows, key=lambda r: -sum(r[1].values()))[: args.top]
    tot = collections.Counter()
    print("%-22s %5s %5s %5s %5s %5s %5s %5s %5s %5s %5s" % ("file:line", "all", "fp64", "mfma", "lanex", "valu", "salu", "ds", "vmem", "scr", "ctl"))
    for (f, n), c in rows:
        tot.update(c)
        print("%-22s %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d" % ("%s:%d" % (f, n), sum(c.values()), c["fp64"], c["mfma"], c["lanex"], c["valu"], c["salu"], c["ds"], c["vmem"], c["scratch"], c["branch"] + c["wait"] + c["barrier"] + c["other"]))
    c = tot
    print("%-22s %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d" % ("total", sum(c.values()), c["fp64"], c["mfma"], c["lanex"], c["valu"], c["salu"], c["ds"], c["vmem"], c["scratch"], c["branch"] + c["wait"] + c["barrier"] + c["other"]))

if __name__ == "__main__":
    main()
