"""rocprofv3 --kernel-trace csv -> the kernels of the LAST `count` launches whose name matches nothing in `skip`, in launch order
with durations: one forward's sequence.  usage: python tools/trace_sequence.py <dir> <first kernel substring> [n_sequences_from_end=2]"""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", ""))))
rows.sort()
first = sys.argv[2]
starts = [i for i, r in enumerate(rows) if first in r[2]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
# sequences = between consecutive occurrences of the marker kernel
marks = starts[-(k + 1):]
for a, b in zip(marks[:-1], marks[1:]):
    seq = rows[a:b]
    tot = sum(e - s for s, e, *_ in seq) / 1e3
    print("---- sequence of %d kernels, %.1f us of kernels, %.1f us wall" % (len(seq), tot, (seq[-1][1] - seq[0][0]) / 1e3))
    for s, e, n, g, w in seq:
        print("%8.1f us  grid %-10s wg %-5s %s" % ((e - s) / 1e3, g, w, n.replace("void ml3d::", "").replace("ml3d::", "")[:110]))
