"""Per-kernel mean of every hardware counter in a rocprofv3 --pmc run (rocpd sqlite).
usage: python profiles/summarize_pmc.py <results.db> [out.csv]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
    cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c or c == "pmc_name"][0]
    vcol = "counter_value" if "counter_value" in cols else "value"
    rows = cur.execute("select name, dispatch_id, %s, %s, duration from pmc_events" % (cname, vcol)).fetchall()
    per = {}
    for kn, disp, cn, v, dur in rows:
        kn = re.sub(r"\s*\[clone.*", "", kn or "?")
        d = per.setdefault((kn, cn), {})
        d[disp] = d.get(disp, 0.0) + float(v)       # sum over instances / XCDs of one dispatch
        per.setdefault((kn, "_duration_ns"), {})[disp] = float(dur)
    lines = ["Kernel,Counter,Dispatches,MeanPerDispatch"]
    for (kn, cn), d in sorted(per.items()):
        lines.append('"%s",%s,%d,%.1f' % (kn, cn, len(d), sum(d.values()) / len(d)))
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
