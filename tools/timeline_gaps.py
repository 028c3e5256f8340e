"""GPU timeline of a rocprofv3 --kernel-trace --memory-copy-trace run (rocpd sqlite): per-kernel busy
time, idle time between consecutive GPU operations, and the operations that follow the largest gaps.

  python tools/timeline_gaps.py <results.db> [skip_fraction]
"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"[<(].*", "", name)


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    ops = [(s, e, short(n)) for n, s, e in db.execute("select name, start, end from kernels")]
    try:
        ops += [(s, e, "copy:" + str(n)) for n, s, e in db.execute("select name, start, end from memory_copies")]
    except sqlite3.OperationalError:
        pass
    ops.sort()
    ops = ops[int(len(ops) * skip):]  # steady state only
    span = ops[-1][1] - ops[0][0]
    busy = collections.Counter()
    gap_after = collections.Counter()
    gap_n = collections.Counter()
    last_end, last_name = ops[0][0], "start"
    for s, e, n in ops:
        busy[n] += e - s
        if s > last_end:
            gap_after[(last_name, n)] += s - last_end
            gap_n[(last_name, n)] += 1
        if e > last_end:
            last_end, last_name = e, n
    tb = sum(busy.values())
    print(f"span {span / 1e6:.3f} ms, busy {tb / 1e6:.3f} ms ({100 * tb / span:.1f} %), idle {(span - tb) / 1e6:.3f} ms")
    print("-- busy by operation")
    for n, t in busy.most_common(25):
        print(f"{t / 1e6:9.3f} ms  {n}")
    print("-- idle by (previous -> next)")
    for k, t in gap_after.most_common(20):
        print(f"{t / 1e6:9.3f} ms  n={gap_n[k]:4d} avg {t / gap_n[k] / 1e3:7.1f} us  {k[0]} -> {k[1]}")


if __name__ == "__main__":
    main()
