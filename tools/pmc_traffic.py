#!/usr/bin/env python3
"""Summarises two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench into the
per-kernel HBM traffic table that bench.py's `roofline.traffic` cites:

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/pmc_FETCH_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/pmc_WRITE_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
  python tools/pmc_traffic.py out/pmc_FETCH_SIZE out/pmc_WRITE_SIZE > profiles/rNN_pmc_traffic_msm_n2e20.json

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both
counters are in KB; FETCH_SIZE undercounts wide coalesced reads by 2x (so x2), WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                per_dispatch[row["Dispatch_Id"]] += float(row["Counter_Value"])
                name = row["Kernel_Name"].split("(")[0]
                if name.startswith("void "):                    # template instances: "void k_accum_l0<false>(...)"
                    name = name[5:]
                names[row["Dispatch_Id"]] = name.split("<")[0]
        for d, v in per_dispatch.items():
            acc[names[d]].append(v)
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        if not (k.startswith("k_") or k.startswith("rpd::k_")):
            continue
        f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1)
        w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
        rows.append({"kernel": k, "launches": len(fetch.get(k, [])), "FETCH_SIZE_KB_avg": round(f, 1), "WRITE_SIZE_KB_avg": round(w, 1),
                     "hbm_bytes_per_launch_guide_corrected": int(1024 * (2 * f + w)), "hbm_bytes_per_launch_raw": int(1024 * (f + w))})
    rows.sort(key=lambda r: -r["hbm_bytes_per_launch_guide_corrected"])
    print(json.dumps({
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
                   "--no-cpu-baseline (two separate passes; tools/pmc_traffic.py)",
        "workload": "MSM n=2^20, c=16, two MSMs in flight (round 6: chunks of 29 entries = three rounds of wave slots, wave priority for the stages beside the accumulation)",
        "correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2; "
                      "WRITE_SIZE exact; unit KB",
        "note": "k_accum_l0 gathers 64-B points (4 x dwordx4 per lane from one random 64-B-aligned address): for that width the "
                "counter's x2 correction is uncalibrated (requested bytes = 16.8M x 64 B + 67 MB indices = 1.14e9 B), so bench.py reports the RAW "
                "figure (hbm_bytes_per_launch_raw) as roofline.traffic and the corrected one beside it. The 64 MB point "
                "array is re-read once per window (16x) and is served by L2 / Infinity Cache, not HBM.",
        "kernels": rows}, indent=1))


if __name__ == "__main__":
    main()
