"""Condense rocprofv3 result databases (rocpd sqlite) into small CSV summaries.

  python tools/rocprof_summary.py stats   <results.db> <out.csv>   per-kernel calls / total / average duration
  python tools/rocprof_summary.py counter <results.db> <out.csv>   per-kernel average of each collected counter
"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*", "", name)


def main():
    mode, db_path, out_path = sys.argv[1:4]
    db = sqlite3.connect(db_path)
    rows = []
    if mode == "stats":
        q = "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc"
        total = sum(r[2] for r in db.execute(q))
        rows.append(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
        for name, n, tot, avg, lo, hi in db.execute(q):
            rows.append([short(name), n, f"{tot / 1e6:.4f}", f"{avg / 1e3:.2f}", f"{lo / 1e3:.2f}", f"{hi / 1e3:.2f}",
                         f"{100 * tot / total:.2f}"])
    else:
        q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
             "group by kernel_name, counter_name order by 5 desc")
        rows.append(["kernel", "counter", "dispatches", "avg_per_dispatch", "sum"])
        for name, cname, n, avg, tot in db.execute(q):
            rows.append([short(name), cname, n, f"{avg:.6g}", f"{tot:.6g}"])
    with open(out_path, "w", newline="") as f:
        csv.writer(f).writerows(rows)
    for r in rows[:16]:
        print(",".join(str(x) for x in r))


if __name__ == "__main__":
    main()
