#!/usr/bin/env python3
"""Static instruction mix of every function of one gfx950 code object (cross-compiles, no GPU needed).

    scripts/isa_mix.py window_solve.hip -DAVM_TP=1            (csrc/Makefile's flags for that build)
    scripts/isa_mix.py --co build/isa/tp.co                   (an already unbundled code object)
    scripts/isa_mix.py ... --md                               (markdown table, as committed under profiles/)
    scripts/isa_mix.py ... --func eval_jac --blocks           (per basic block of one function: where the non-FP64 VALU sits)

Classes (one per instruction, first match):
    mfma      v_mfma_*                                         64 (16x16x4 f64) / 18 (4x4x4 f64) cycles of the FP64 pipe
    fp64      VALU whose arithmetic is FP64 (v_*_f64, v_cvt_*f64*, v_fmac/fma/mul/add/min/max/rcp/rsq/ldexp/cmp ... _f64)
    lanex     cross-lane traffic: v_readlane / v_writelane / v_readfirstlane / v_permlane* / any *_dpp / ds_bpermute / ds_swizzle
    valu      every other v_* (integer, address arithmetic, v_mov, v_cndmask, 32-bit compares, ...)
    salu      s_* arithmetic / moves / compares (without the control classes below)
    ds        ds_* (LDS) except the lane-exchange forms
    vmem      global_* / buffer_* / flat_*
    scratch   scratch_* (private memory: spills)
    branch    s_cbranch_* / s_branch / s_call / s_setpc / s_swappc
    wait      s_waitcnt* / s_nop / s_sleep
    barrier   s_barrier
    other     s_setprio, s_endpgm, ...
What the hardware counters see: SQ_INSTS_VALU = fp64 + lanex(VALU forms) + valu + mfma; SQ_INSTS_MFMA = mfma.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "anticipated-vins-mono_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
CLASSES = ["mfma", "fp64", "lanex", "valu", "salu", "ds", "vmem", "scratch", "branch", "wait", "barrier", "other"]


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        if op.endswith("_dpp") or "readlane" in op or "writelane" in op or "readfirstlane" in op or "permlane" in op:
            return "lanex"
        if "_f64" in op or "f64_" in op:
            return "fp64"
        return "valu"
    if op.startswith("ds_"):
        return "lanex" if ("bpermute" in op or "permute" in op or "swizzle" in op) else "ds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op == "s_barrier":
        return "barrier"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_call", "s_setpc", "s_swappc", "s_getpc")):
        return "branch"
    if op.startswith(("s_setprio", "s_endpgm", "s_code_end", "s_sethalt", "s_trap", "s_icache", "s_dcache", "s_memtime", "s_memrealtime")):
        return "other"
    if op.startswith("s_"):
        return "salu"
    return "other"


def build_co(src, defs):
    """csrc/Makefile's flags for the solve builds (IPRA for the base and throughput builds, not for -DAVM_X)."""
    tmp = tempfile.mkdtemp()
    flags = ["-O3", "-std=c++17", "-fconstexpr-steps=16000000", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-pass-failed"]
    if os.path.basename(src).startswith("window_solve"):
        flags += ["-mllvm", "-sink-insts-to-avoid-spills"]
        if not any(d.startswith("-DAVM_X") for d in defs):
            flags += ["-mllvm", "-enable-ipra", "-fno-optimize-sibling-calls"]
        else:
            flags += ["-mllvm", "-amdgpu-prealloc-sgpr-spill-vgprs"]
    bundle, co = os.path.join(tmp, "k.bundle"), os.path.join(tmp, "k.co")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + defs + ["--cuda-device-only", "-c", src, "-o", bundle], cwd=CSRC)
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + bundle, "--output=" + co])
    return co


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    res = {}
    for n, d in zip(names, out):
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        d = re.sub(r"^avm::", "", d)
        d = re.sub(r"\(.*$", "", d)  # drop the argument list
        res[n] = d
    return res


INSN = re.compile(r"^\s+([a-z][a-z0-9_]+)\b(.*?)//\s*([0-9A-Fa-f]{12}):")
FUNC = re.compile(r"^[0-9a-f]+ <([^>]+)>:")
LABEL = re.compile(r"^[0-9a-f]+ <(L[0-9A-Za-z_]+|[^>]+\+0x[0-9a-f]+)>:")


def parse(co):
    txt = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--symbolize-operands", co], text=True)
    funcs = collections.OrderedDict()
    cur = None
    for line in txt.split("\n"):
        m = FUNC.match(line)
        if m and not LABEL.match(line):
            cur = funcs.setdefault(m.group(1), [])
            cur.append(("label", m.group(1), ""))
            continue
        if cur is None:
            continue
        if LABEL.match(line):
            cur.append(("label", LABEL.match(line).group(1), ""))
            continue
        m = INSN.match(line)
        if m:
            cur.append(("insn", m.group(1), m.group(2)))
    return funcs


def mix(insns):
    c = collections.Counter()
    for kind, op, _ in insns:
        if kind == "insn":
            c[classify(op)] += 1
    return c


def blocks(insns):
    """basic blocks (split at labels and after branches) with their mixes, and the back edges that make a block part of a loop"""
    out, cur, name = [], [], "entry"
    for kind, op, rest in insns:
        if kind == "label":
            if cur:
                out.append((name, cur))
            cur, name = [], op
            continue
        cur.append((kind, op, rest))
        if classify(op) == "branch":
            out.append((name, cur))
            cur, name = [], name + "'"
    if cur:
        out.append((name, cur))
    return out


def fmt_row(name, c, md):
    valu_all = c["mfma"] + c["fp64"] + c["lanex"] + c["valu"]
    tot = sum(c.values())
    cols = [name, tot, valu_all, c["fp64"], "%.0f%%" % (100.0 * c["fp64"] / max(valu_all - c["mfma"], 1)), c["mfma"], c["lanex"], c["valu"], c["salu"], c["ds"],
            c["vmem"], c["scratch"], c["branch"], c["wait"], c["barrier"]]
    if md:
        return "| " + " | ".join(str(x) for x in cols) + " |"
    return "%-34s %6d %6d %6d %5s %5d %6d %6d %6d %5d %5d %7d %6d %5d %4d" % tuple(cols)


HEAD = ["function", "all", "VALU", "FP64", "FP64/VALU", "MFMA", "lane-x", "other VALU", "SALU", "DS", "VMEM", "scratch", "branch", "wait", "barrier"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src", nargs="?")
    ap.add_argument("--co")
    ap.add_argument("--md", action="store_true")
    ap.add_argument("--func")
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("--min", type=int, default=0, help="with --blocks: only blocks with at least this many instructions")
    args, defs = ap.parse_known_args()
    co = args.co or build_co(args.src, defs)
    funcs = parse(co)
    names = demangle(list(funcs))
    if args.func:
        for f, insns in funcs.items():
            if args.func not in names[f]:
                continue
            print("# " + names[f])
            if args.blocks:
                print("%-34s %6s %6s %6s %5s %5s %6s %6s %6s %5s %5s %7s %6s %5s %4s" % tuple(HEAD))
                for bn, b in blocks(insns):
                    c = mix(b)
                    if sum(c.values()) >= args.min:
                        print(fmt_row(bn[-34:], c, False))
            else:
                ops = collections.Counter(op for k, op, _ in insns if k == "insn")
                for op, n in ops.most_common():
                    print("%6d  %-40s %s" % (n, op, classify(op)))
        return
    if args.md:
        print("| " + " | ".join(HEAD) + " |")
        print("|" + "---|" * len(HEAD))
    else:
        print("%-34s %6s %6s %6s %5s %5s %6s %6s %6s %5s %5s %7s %6s %5s %4s" % tuple(HEAD))
    for f, insns in funcs.items():
        print(fmt_row(("`%s`" % names[f]) if args.md else names[f][:34], mix(insns), args.md))


if __name__ == "__main__":
    main()
