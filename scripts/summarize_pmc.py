#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter_collection.csv files per kernel -> one CSV (+ traffic JSON).

usage: summarize_pmc.py OUT.csv DIR [DIR ...]   (DIR holds *_counter_collection.csv)
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE
on gfx950 counts exactly half of the bytes of a wide (16 B/lane) coalesced read stream, so the corrected
read traffic is 2 x FETCH_SIZE x 1024 B.  Both raw and corrected figures are written.
"""
import collections
import csv
import glob
import json
import os
import sys

out, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for d in agg.values() for c in d})
with open(out, "w") as f:
    f.write("kernel,dispatches," + ",".join(names) + "\n")
    for k, d in sorted(agg.items()):
        f.write('"%s",%d,' % (k, max(len(v) for v in d.values())) +
                ",".join("%.0f" % (sum(d[c]) / len(d[c])) if c in d else "" for c in names) + "\n")
traffic = {}
for k, d in agg.items():
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        fe = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [0])), 1)
        wr = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [0])), 1)
        traffic[k] = dict(fetch_kib_raw=fe, write_kib_raw=wr, read_bytes_corrected=2.0 * fe * 1024.0,
                          write_bytes=wr * 1024.0, bytes_per_launch=2.0 * fe * 1024.0 + wr * 1024.0)
for k, d in agg.items():            # VALU instructions per launch: the bound of the two Gibbs kernels
    if "SQ_INSTS_VALU" in d:
        traffic.setdefault(k, {})["valu_insts"] = sum(d["SQ_INSTS_VALU"]) / len(d["SQ_INSTS_VALU"])
json.dump(traffic, open(os.path.splitext(out)[0] + "_traffic.json", "w"), indent=1, sort_keys=True)
