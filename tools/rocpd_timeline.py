#!/usr/bin/env python3
"""Kernel timeline / per-kernel statistics out of a rocprofv3 results database (rocpd sqlite: what `rocprofv3 --kernel-trace -d DIR -o NAME`
writes as DIR/NAME_results.db on this image).
  python tools/rocpd_timeline.py DB stats                        per-kernel calls / total / average / min / max (us)
  python tools/rocpd_timeline.py DB timeline [FIRST [COUNT]]     dispatches in start order: start (us, relative), duration, queue, kernel, grid
  python tools/rocpd_timeline.py DB after KERNEL_SUBSTRING [COUNT]   the timeline starting at the LAST dispatch whose name contains the substring
  python tools/rocpd_timeline.py DB nth KERNEL_SUBSTRING N [COUNT] [BACK]   ... starting BACK rows before the N-th (0-based) such dispatch"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute("select k.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x "
                        "from %s d join %s k on d.kernel_id = k.id order by d.start" % (kd, ks)))


def short(n):
    n = re.sub(r"\(.*", "", n)
    m = re.match(r"_Z\d+(k_\w+?)(I[LbEe0-9_]+)?(Ev|PK|P[a-z]|[0-9]|j|S_).*", n)
    return (n if not n.startswith("_Z") else re.sub(r"^_ZN?\d*", "", n))[:48]


mode = sys.argv[2] if len(sys.argv) > 2 else "stats"
if mode == "stats":
    agg = {}
    for r in rows:
        a = agg.setdefault(short(r[0]), [0, 0.0, 1e18, 0.0])
        d = (r[2] - r[1]) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print("%-50s %7s %12s %10s %9s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-50s %7d %12.1f %10.2f %9.2f %9.2f %6.2f" % (name, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / total))
else:
    if mode == "after":
        idx = [i for i, r in enumerate(rows) if sys.argv[3] in r[0]]
        first = idx[-1] if idx else 0
        count = int(sys.argv[4]) if len(sys.argv) > 4 else 80
    elif mode == "nth":
        idx = [i for i, r in enumerate(rows) if sys.argv[3] in r[0]]
        nth = int(sys.argv[4]) if len(sys.argv) > 4 else 0
        back = int(sys.argv[6]) if len(sys.argv) > 6 else 0
        first = max(0, (idx[nth] if len(idx) > nth else 0) - back)
        count = int(sys.argv[5]) if len(sys.argv) > 5 else 80
    else:
        first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
        count = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    t0 = rows[first][1]
    for r in rows[first:first + count]:
        print("%10.1f %9.1f  q%-2s %-48s grid %d x %d x %d  wg %d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], short(r[0]), r[4], r[5], r[6], r[7]))
