"""rocprofv3 counter_collection.csv (one row per dispatch and counter) -> the small table the repository keeps under profiles/:
Kernel, Counter, Dispatches, MeanPerDispatch.  usage: python tools/summarize_pmc.py <dir with *counter_collection.csv> <out.csv>"""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[(r["Kernel_Name"], r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
# the same pass's kernel trace (rocprofv3 --kernel-trace --pmc ...): mean duration per kernel, as a pseudo counter DURATION_NS
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        try:
            a = acc[(r["Kernel_Name"], "DURATION_NS")]
            a[0] += 1
            a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        except (KeyError, ValueError):
            break
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Counter", "Dispatches", "MeanPerDispatch"])
    for (k, c), (n, v) in sorted(acc.items()):
        w.writerow([k, c, n, round(v / n, 1)])
print("wrote", sys.argv[2], len(acc), "rows")
