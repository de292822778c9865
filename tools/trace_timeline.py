#!/usr/bin/env python3
"""The last few MSMs of a rocprofv3 kernel trace of the two-lane pipeline as a timeline: every kernel with its queue, start and end
relative to the first one shown, and for every k_accum_l0 which kernels of the OTHER queue ran inside its interval."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    name = r.get("Kernel_Name") or r.get("Name")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name.split("(")[0].replace("void ", "")))
ks.sort()
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0          # only accumulations at least this long (the large MSMs of a mixed trace)
back = int(sys.argv[3]) if len(sys.argv) > 3 else 7
accs = [i for i, k in enumerate(ks) if k[3].startswith("k_accum_l0") and (k[1] - k[0]) / 1e3 >= min_us]
first = accs[-back] if len(accs) >= back else accs[0]
t0 = ks[first][0]
sel = [k for k in ks[first:] if k[0] - t0 < 4.5e6]
for s, e, q, n in sel:
    print("%9.1f %9.1f us  %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
print()
for s, e, q, n in sel:
    if n.startswith("k_accum_l0"):
        inside = [(n2, max(s, s2), min(e, e2)) for s2, e2, q2, n2 in sel if q2 != q and s2 < e and e2 > s]
        print("accumulate on q%s %.1f..%.1f us (%.1f): beside it on the other queue: %s" % (q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3,
              ", ".join("%s %.0f us" % (n2, (b - a) / 1e3) for n2, a, b in inside) or "nothing"))
