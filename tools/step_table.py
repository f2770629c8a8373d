"""Per-STEP view of a rocprofv3 kernel table: ms per step, launches per step and register / LDS use per kernel, plus the GPU-busy
total per step (sum over streams) -- the numbers profiles/DESIGN_rounds_1_to_4.md §3 / §8 argue with.
usage: python tools/step_table.py <results.db | kernel_stats.csv> <steps> [top]
  <steps> = timed + warm-up steps of the profiled command (bench.py: --steps + --warmup)"""
import csv
import re
import sqlite3
import sys


def from_db(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end - start), max(vgpr_count), max(lds_size), max(scratch_size) "
                       "from kernels group by name").fetchall()
    return [(re.sub(r"\s*\[clone.*", "", n), int(c), float(t), v, l, s) for n, c, t, v, l, s in rows]


def from_csv(path):
    return [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), None, None, None) for r in csv.DictReader(open(path))]


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    rows.sort(key=lambda r: -r[2])
    total = sum(r[2] for r in rows)
    print("GPU-busy per step %.3f ms over %d kernels, %.0f launches per step" %
          (total / steps / 1e6, len(rows), sum(r[1] for r in rows) / steps))
    print("%-92s %9s %9s %8s %5s %7s" % ("kernel", "ms/step", "calls/st", "avg us", "vgpr", "lds"))
    for n, c, t, v, l, s in rows[:top]:
        print("%-92s %9.3f %9.1f %8.1f %5s %7s%s" % (n[:92], t / steps / 1e6, c / steps, t / c / 1e3, v if v is not None else "-",
                                                      l if l is not None else "-", "  SCRATCH %s" % s if s else ""))


if __name__ == "__main__":
    main()
