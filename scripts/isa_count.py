#!/usr/bin/env python3
"""ISA-level instruction counts of stage 1 of the mu/E pass (stats_agg_kernel), phase by phase.

Compiles desman_amd/csrc/kernels_stats.hip for gfx950 with -DDSM_ISA_MARKS (dsm_binom.h: ISA_MARK leaves a `; MARK name` comment in the
assembly at the start of every phase of the cell / item code; the comments are `asm volatile`, so the marked build's schedule can differ
slightly from the product's -- its TOTAL is compared with the unmarked build's below), walks the kernel's assembly in layout order and
charges every instruction to the last mark seen.  Static counts are turned into wave-instructions per 64-cell slot with the trip counts of
the data: phases inside a loop (`*_loop`, found from the assembler's loop comments) are multiplied by the mean trip count PER ITEM of the
wavefront (= the maximum over its 64 lanes), which scripts/isa_count.py measures on the CPU by drawing the same binomials for the bench's
table (numpy; law-equivalent draws, not the kernel's streams).

usage: isa_count.py [--kernel REGEX] [--V 10000 --S 64 --G 8] [--src PATH] [--extra "-DX ..."]   -> table on stdout
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compile_asm(src, extra, marks=True):
    d = tempfile.mkdtemp(prefix="isa_", dir="/tmp")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "desman_amd", "csrc"),
           "-I" + os.path.join(ROOT, "include"), "--save-temps", "-c", src, "-o", os.path.join(d, "o.o")] + (["-DDSM_ISA_MARKS"] if marks else []) + extra
    subprocess.run(cmd, cwd=d, check=True, capture_output=True)
    for f in os.listdir(d):
        if f.endswith("gfx950.s"):
            return open(os.path.join(d, f)).read()
    raise RuntimeError("no device assembly")


def kernel_body(asm, mangled_re):
    m = re.search(r"^(%s):" % mangled_re, asm, re.M)
    if not m:
        raise RuntimeError("kernel not found: " + mangled_re)
    end = asm.index("s_endpgm", m.end())
    return m.group(1), asm[m.end():end].splitlines()


def classify(op):
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


FP64 = re.compile(r"_f64|v_mad_u64|v_mad_i64")


PARENTS = set()


def blocks(lines):
    """basic blocks in layout order: dicts {label, loop (header label or None), depth, phase (last mark seen), ops [mnemonics]}"""
    out = []
    cur = dict(label="entry", loop=None, depth=0, phase="prologue", ops=[])
    phase = "prologue"
    for ln in lines:
        t = ln.strip()
        mk = re.match(r";\s*MARK\s+(\S+)", t)
        if mk:
            phase = mk.group(1)
            if not cur["ops"]:
                cur["phase"] = phase
            else:                                     # a mark inside a block splits it (same loop membership)
                out.append(cur)
                cur = dict(label=cur["label"] + "+", loop=cur["loop"], depth=cur["depth"], phase=phase, ops=[])
            continue
        lab = re.match(r"(\.LBB\d+_\d+):", t) or re.match(r";\s*%bb\.(\d+):", t)
        if lab:
            if cur["ops"] or cur["label"] == "entry":
                out.append(cur)
            cur = dict(label=lab.group(1) if t.startswith(".") else "bb." + lab.group(1), loop=None, depth=0, phase=phase, ops=[])
        m = re.search(r"in Loop: Header=(\S+) Depth=(\d+)", t)
        if m:
            cur["loop"], cur["depth"] = m.group(1), int(m.group(2))
        m = re.search(r"Parent Loop (\S+) Depth=(\d+)", t)
        if m:
            PARENTS.add(m.group(1))
            cur.setdefault("parents", []).append((m.group(1), int(m.group(2))))
        m = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", t)
        if m:
            cur["loop"], cur["depth"] = "BB" + cur["label"][4:] if cur["label"].startswith(".LBB") else cur["label"], int(m.group(1))
        if not t or t.startswith((";", "//")) or (t.startswith(".") and not t.startswith(".LBB")) or t.endswith(":"):
            continue
        if t.startswith(".LBB"):
            continue
        cur["ops"].append(t.split()[0])
    out.append(cur)
    return out


def tally(ops):
    d = {}
    for op in ops:
        c = classify(op)
        d[c] = d.get(c, 0) + 1
        if c == "valu" and FP64.search(op):
            d["valu_fp64"] = d.get("valu_fp64", 0) + 1
        if op.startswith("v_rcp_f64"):
            d["rcp_f64"] = d.get("rcp_f64", 0) + 1
    return d


def add(a, b, f=1.0):
    for k, v in b.items():
        a[k] = a.get(k, 0) + v * f
    return a


def loop_kind(ops):
    """which of the kernel's inner loops a loop body is, by what only it contains"""
    s = set(ops)
    if "v_alignbit_b32" in s and not any(o.startswith("v_mul_f64") for o in ops):
        return "reads"
    if any(o.startswith("v_fma_f64") for o in ops) and any(o.startswith("ds_read") for o in ops):
        return "binv"
    if sum(o.startswith("v_mul_f64") for o in ops) >= 2 and not any(o.startswith("ds_") for o in ops):
        return "pw"
    if any(o.startswith("ds_read_b64") for o in ops) and any(o.startswith("v_add_f64") for o in ops):
        return "gamma"
    if any(o.startswith("global_atomic") for o in ops) or any(o.startswith("v_mbcnt") for o in ops):
        return "handover"
    return "other"


def walk(lines):
    PARENTS.clear()
    """-> ({phase: tally of the straight-line code of the slot loop}, {loop kind: [tally of one trip, copies]}, rare tally, once tally)"""
    bl = blocks(lines)
    loops = {}
    for b in bl:
        if b["depth"] >= 2:
            loops.setdefault(b["loop"], []).extend(b["ops"])
    # the slot loop is the depth-1 loop; inner loops are those whose blocks have depth >= 2 AND are not the item loop of a rolled build
    kinds = {}
    inner_headers = {}
    for h, ops in loops.items():
        inner_headers[h] = "item" if h in PARENTS else loop_kind(ops)
    # a loop around the hand-over loop is the hand-over path (rare), not the item loop of a rolled build
    for b in bl:
        if b.get("parents") and inner_headers.get(b["loop"]) == "handover":
            for ph, pd in b["parents"]:
                if pd == b["depth"] - 1 and pd >= 2:
                    inner_headers[ph] = "handover"
    straight, rare, once = {}, {}, {}
    per_kind = {}
    for b in bl:
        t = tally(b["ops"])
        if b["depth"] == 0:
            add(once, t)
        elif b["depth"] == 1 or inner_headers.get(b["loop"]) == "other":
            if any(o.startswith(("v_div_scale_f64", "v_div_fixup_f64", "v_div_fmas_f64")) for o in b["ops"]) or b["phase"] == "cell_handover":
                add(rare, t)
            else:
                add(straight.setdefault(b["phase"], {}), t)
        else:
            k = inner_headers[b["loop"]]
            if k == "handover":
                add(rare, t)
            else:
                add(per_kind.setdefault(k, {}), t)
    ncopies = {}
    for h, k in inner_headers.items():
        ncopies[k] = ncopies.get(k, 0) + 1
    return straight, per_kind, ncopies, rare, once


def trip_counts(V, S, G, vmax=3000):
    """mean over (variant, observed base) items of the wavefront's loop lengths: bits of the largest count (pw), longest search (binv),
    most reads drawn one by one (reads) -- at the generating state of the bench's table"""
    import numpy as np
    from desman_amd.synth import synth_counts
    counts, tt, gg = synth_counts(V, S, G, 1234)
    eta = 0.96 * np.eye(4) + 0.01
    rng = np.random.default_rng(0)
    V = min(V, vmax)
    tt = tt[:V]
    counts = counts[:V]
    onehot = (tt[:, :, None] == np.arange(4)[None, None, :]).astype(float)
    Gam = np.einsum("sg,vga->vsa", gg, onehot)
    lpv = 64
    for l in (32, 16):
        if (S + l - 1) // l * l < (S + lpv - 1) // lpv * lpv:
            lpv = l
    nch = (S + lpv - 1) // lpv
    ng = 64 // lpv
    tot = dict(pw=0.0, binv=0.0, reads=0.0, any_binv=0.0, any_reads=0.0, items=0.0)
    for b in range(4):
        x = counts[:, :, b]
        W = Gam * eta[:, b][None, None, :]
        wm = W.max(-1)
        ws = W.sum(-1) - wm
        q = ws / (ws + wm)
        m = rng.binomial(x, q)
        k = np.where(ws > wm, x - m, m)
        act = x > 0
        bits = np.where(act, np.floor(np.log2(np.maximum(x, 1))).astype(int) + 1, 0)
        # a wavefront = ng variants x lpv samples of one chunk: pad S to nch * lpv, group ng consecutive tasks
        def wave_max(a):
            pad = np.zeros((a.shape[0], nch * lpv), a.dtype)
            pad[:, :S] = a
            t = pad.reshape(a.shape[0] * nch, lpv)                 # tasks (variant, chunk)
            n = (t.shape[0] + ng - 1) // ng * ng
            t = np.concatenate([t, np.zeros((n - t.shape[0], lpv), a.dtype)])
            return t.reshape(n // ng, ng * lpv).max(1)
        tot["pw"] += wave_max(bits).sum()
        tot["binv"] += wave_max(np.where(act, k, 0)).sum()
        tot["reads"] += wave_max(np.where(act, m, 0)).sum()
        tot["any_binv"] += (wave_max(np.where(act, k, 0)) > 0).sum()
        tot["any_reads"] += (wave_max(np.where(act, m, 0)) > 0).sum()
        tot["items"] += wave_max(act.astype(int)).shape[0]
    n = tot.pop("items")
    return {k_: v / n for k_, v in tot.items()}, lpv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=None, help="regex of the mangled kernel name (default: the instantiation of the shape)")
    ap.add_argument("--V", type=int, default=10000)
    ap.add_argument("--S", type=int, default=64)
    ap.add_argument("--G", type=int, default=8)
    ap.add_argument("--src", default=os.path.join(ROOT, "desman_amd", "csrc", "kernels_stats.hip"))
    ap.add_argument("--extra", default="")
    a = ap.parse_args()
    trips, lpv = trip_counts(a.V, a.S, a.G)
    kre = a.kernel or r"_Z16stats_agg_kernelILi%dELi2ELb0EEv14StatsAggParams" % lpv
    extra = a.extra.split()
    name, lines = kernel_body(compile_asm(a.src, extra, True), kre)
    straight, per_kind, ncopies, rare, once = walk(lines)
    _, lines0 = kernel_body(compile_asm(a.src, extra, False), kre)
    s0, k0, n0, r0, o0 = walk(lines0)
    print("# %s   (V, S, G) = (%d, %d, %d), %d lanes per variant" % (name, a.V, a.S, a.G, lpv))
    print("# loop lengths per item of a wavefront (max over its lanes; CPU draw at the generating state): pw %.2f bits, binv %.2f steps "
          "(%.0f%% of items search), reads %.2f (%.0f%% of items draw reads)" % (trips["pw"], trips["binv"], 100 * trips["any_binv"], trips["reads"], 100 * trips["any_reads"]))
    hdr = "%-34s %6s %6s %6s %5s %5s %6s" % ("", "valu", "fp64", "salu", "lds", "vmem", "branch")
    row = lambda nm, d, f=1.0: "%-34s %6.0f %6.0f %6.0f %5.0f %5.0f %6.0f" % (nm, f * d.get("valu", 0), f * d.get("valu_fp64", 0), f * d.get("salu", 0), f * d.get("lds", 0),
                                                                          f * d.get("vmem", 0), f * d.get("branch", 0))

    def report(straight, per_kind, ncopies, rare, once, title, detail):
        print("\n## " + title)
        print(hdr)
        tot = {}
        if detail:
            print("straight-line code of a slot, by the mark it follows in the assembly's layout (approximate: the scheduler moves code across marks):")
            for ph in sorted(straight, key=lambda x: (ORDER.index(x) if x in ORDER else 99, x)):
                print(row("  " + ph, straight[ph]))
        st = {}
        for d in straight.values():
            add(st, d)
        print(row("straight-line, per 64-cell slot", st))
        add(tot, st)
        items_per_slot = 4.0
        for k, trip_key, per in (("gamma", None, a.G * 64.0 / lpv), ("item", None, 4.0), ("pw", "pw", None), ("binv", "binv", None), ("reads", "reads", None)):
            if k not in per_kind:
                continue
            copies = ncopies[k]
            one = {kk: v / copies for kk, v in per_kind[k].items()}             # one trip of one copy
            trips_slot = per if per is not None else trips[trip_key] * items_per_slot
            print(row("loop %-5s: one trip (%d cop%s)" % (k, copies, "y" if copies == 1 else "ies"), one))
            print(row("            x %.1f trips per slot" % trips_slot, one, trips_slot))
            add(tot, one, trips_slot)
        print(row("TOTAL per 64-cell slot (model)", tot))
        print(row("rare paths (IEEE divisions, hand-over), static", rare))
        print(row("once per wavefront (prologue + epilogue), static", once))
        return tot

    ORDER = ["slot_setup", "cell_load", "cell_gamma", "cell_philox", "item_w", "item_seed", "item_argmax", "item_binom_setup", "item_div1", "item_pw", "item_u01",
             "item_div2", "item_binv", "item_binv_end", "item_reads_setup", "item_reads", "item_reads_end", "item_map", "item_esum", "cell_handover", "cell_table", "epilogue"]
    report(straight, per_kind, ncopies, rare, once, "build with phase marks (-DDSM_ISA_MARKS)", True)
    t = report(s0, k0, n0, r0, o0, "product build (no marks)", False)
    print("\n# model: %.0f VALU + %.0f SALU/branch wave-instructions per slot; %d slots -> %.3g VALU wave-instructions per launch" %
          (t.get("valu", 0), t.get("salu", 0) + t.get("branch", 0), (a.V * ((a.S + lpv - 1) // lpv) + 64 // lpv - 1) // (64 // lpv),
           t.get("valu", 0) * ((a.V * ((a.S + lpv - 1) // lpv) + 64 // lpv - 1) // (64 // lpv))))


if __name__ == "__main__":
    main()
