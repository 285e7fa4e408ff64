#!/usr/bin/env python3
"""Register / spill / scratch / LDS numbers of the built kernels, from the code objects inside desman_amd/lib/obj/*.o
(`llvm-objdump --offloading` + `llvm-readelf --notes`): one line per kernel whose demangled name matches the pattern.
usage: kernel_regs.py [regex ...]      default: the hot instantiations of the Gibbs loop and the NMFT update"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
pats = sys.argv[1:] or [r"^(void )?tau_kernel<(32|64), (2|3|4|6|8), true, true>", r"^(void )?stats_agg_kernel<(32|64), 2, false>",
                        r"^(void )?nmft_persist_kernel<", r"^(void )?nmft_mfma_kernel<", r"^(void )?dirichlet_kernel\(", r"^(void )?stats_big_kernel<2>",
                        r"^(void )?stats_stage2_kernel<2>"]
rows = []
with tempfile.TemporaryDirectory() as d:
    for o in sorted(glob.glob(os.path.join(ROOT, "desman_amd", "lib", os.environ.get("DSM_OBJDIR", "obj"), "*.o"))):
        t = os.path.join(d, os.path.basename(o))
        shutil.copy(o, t)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", t], capture_output=True)
        for co in glob.glob(t + ".*gfx950*"):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                blk = ".agpr_count:" + blk
                f = {m.group(1): m.group(2).strip() for m in re.finditer(r"\.(\w+):\s+(.+)", blk)}
                name = f.get("name", "?")
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
                dem = dem.split("(")[0] + ("(" if "(" in dem else "")
                if any(re.search(p, dem) for p in pats):
                    rows.append((dem.rstrip("("), f.get("vgpr_count"), f.get("agpr_count"), f.get("sgpr_count"), f.get("vgpr_spill_count"),
                                 f.get("sgpr_spill_count"), f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size")))
print("%-52s %5s %5s %5s %10s %10s %12s %10s" % ("kernel", "vgpr", "agpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch_B/ln", "lds_B"))
for r in sorted(set(rows)):
    print("%-52s %5s %5s %5s %10s %10s %12s %10s" % r)
