"""Timeline of one Gibbs iteration from a rocprofv3 kernel trace: kernel durations and the idle gaps between them.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-nmft
       python scripts/trace_gaps.py DIR"""
import csv, glob, sys, collections
f = [x for x in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)][0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")) for r in csv.DictReader(open(f))))
main = [r for r in rows if "mt_fill" not in r[2]]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(main[:-1], main[1:]):
    dur[a[2]].append(a[1] - a[0])
    gap[a[2] + " -> " + b[2]].append(b[0] - a[1])
tail = lambda v: sorted(v)[len(v) // 2] / 1e3
for k, v in dur.items():
    if len(v) > 50: print("%-28s n=%5d median %.2f us" % (k, len(v), tail(v)))
for k, v in gap.items():
    if len(v) > 50: print("gap %-44s n=%5d median %.2f us" % (k, len(v), tail(v)))
