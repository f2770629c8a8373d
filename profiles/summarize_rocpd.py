"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the `--stats`-style per-kernel table.
usage: python profiles/summarize_rocpd.py <results.db> [out.csv]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        n = re.sub(r"\s*\[clone.*", "", n)
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (n, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]))
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
