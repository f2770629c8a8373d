"""rocprofv3 --kernel-trace output (…_kernel_trace.csv) -> how busy the GPU was: per HIP stream / hardware queue the kernel time and
the union of its kernels' intervals, the time with 0 / 1 / 2 / 3+ kernels running at once, the longest kernels, and the gaps between
consecutive kernels of the busiest stream.  Window: the last `frac` of the trace (the steady state of a bench loop).
usage: python tools/trace_timeline.py <dir or csv> [frac=0.5]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?")), r.get("Queue_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
w0 = t1 - int((t1 - t0) * frac)
rows = [r for r in rows if r[0] >= w0]
span = (t1 - w0) / 1e6
print("window %.2f ms, %d kernels" % (span, len(rows)))
by = defaultdict(list)
for s, e, n, st, q in rows:
    by[(st, q)].append((s, e, n))
for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in v) / 1e6
    gaps = sorted(((v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)))
    small = [g for g in gaps if g < 200]
    print("stream %s queue %s: %5d kernels, busy %.2f ms (%.0f %% of window), median gap %.1f us, gaps<200us sum %.2f ms, mean kernel %.1f us"
          % (k[0], k[1], len(v), busy, 100 * busy / span, gaps[len(gaps) // 2] if gaps else 0, sum(small) / 1e3, busy * 1e3 / len(v)))
ev = []
for s, e, *_ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
conc = defaultdict(int)
cur, last = 0, w0
for t, d in ev:
    conc[min(cur, 3)] += t - last
    cur += d; last = t
tot = sum(conc.values())
print("concurrency: " + ", ".join("%d%s kernels %.1f %%" % (k, "+" if k == 3 else "", 100 * v / tot) for k, v in sorted(conc.items())))
agg = defaultdict(lambda: [0, 0])
for s, e, n, *_ in rows:
    a = agg[n.split("(")[0][:70]]; a[0] += 1; a[1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-70s %5d x %7.1f us = %6.2f ms" % (n, c, t / c / 1e3, t / 1e6))
