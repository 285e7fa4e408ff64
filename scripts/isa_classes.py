#!/usr/bin/env python3
"""Static instruction counts by class of one or more kernels of a HIP source file, old against new.

Compiles the file for gfx950 (the product's flags) at the working tree and, with --rev, at a git revision, finds each kernel by a regular
expression on its mangled name and counts its instructions by class: wide / narrow fp64 matrix instructions, fp64 vector, other vector,
scalar, waits, LDS reads / writes / lane permutes, memory.  Static counts of the whole kernel (prologue, the rare paths and the epilogue
included), not a trace: good for "what did this change add and remove".

usage: isa_classes.py [--src desman_amd/csrc/kernels_nmft.hip] [--rev HEAD~3] KERNEL_REGEX [KERNEL_REGEX ...]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_asm(path, incdir):
    d = tempfile.mkdtemp(prefix="isacls_", dir="/tmp")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + incdir, "-I" + os.path.join(ROOT, "include"),
           "--save-temps", "-c", path, "-o", os.path.join(d, "o.o")]
    subprocess.run(cmd, cwd=d, check=True, capture_output=True)
    for f in os.listdir(d):
        if f.endswith("gfx950.s"):
            return open(os.path.join(d, f)).read()
    raise RuntimeError("no device assembly")


def classify(op):
    if op.startswith("v_mfma_f64_4x4"): return "mfma_4x4x4"
    if op.startswith("v_mfma"): return "mfma_16x16x4"
    if op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"): return "lds_permute"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "lds_write"
    if op.startswith("ds_"): return "lds_other"
    if op.startswith("v_") and "f64" in op: return "valu_f64"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait_nop"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "memory"
    return "other"


def count(asm, rx):
    out = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*$", asm, re.M):
        name = m.group(1)
        if not re.search(rx, name) or name.endswith("$local"):
            continue
        end = asm.index("s_endpgm", m.end())
        c = collections.Counter()
        for l in asm[m.end():end].splitlines():
            l = l.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            c[classify(l.split()[0])] += 1
        out.append((name, c))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default="desman_amd/csrc/kernels_nmft.hip")
    ap.add_argument("--rev", default=None)
    ap.add_argument("kernels", nargs="+")
    a = ap.parse_args()
    src = os.path.join(ROOT, a.src)
    inc = os.path.dirname(src)
    sets = [("working tree", compile_asm(src, inc))]
    if a.rev:
        t = tempfile.mkdtemp(prefix="isacls_src_", dir="/tmp")
        old = os.path.join(t, os.path.basename(src))
        open(old, "w").write(subprocess.run(["git", "-C", ROOT, "show", "%s:%s" % (a.rev, a.src)], check=True, capture_output=True, text=True).stdout)
        sets.insert(0, (a.rev, compile_asm(old, inc)))
    cols = ["mfma_16x16x4", "mfma_4x4x4", "valu_f64", "valu_other", "salu", "wait_nop", "lds_read", "lds_write", "lds_permute", "memory", "other"]
    print("%-14s %-52s " % ("source", "kernel") + " ".join("%12s" % c for c in cols) + "        total")
    for rx in a.kernels:
        for tag, asm in sets:
            for name, c in count(asm, rx):
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
                print("%-14s %-52s " % (tag, dem.split("(")[0][-52:]) + " ".join("%12d" % c[k] for k in cols) + " %12d" % sum(c.values()))


if __name__ == "__main__":
    sys.exit(main())
