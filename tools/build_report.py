#!/usr/bin/env python3
"""Builds libbpmi.so with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
VGPRs, scratch bytes per lane, occupancy (waves per SIMD), LDS bytes.   python tools/build_report.py [name-filter]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "python-bulletproofs_amd", "csrc", "bpmi.hip")
LIB = os.path.join(REPO, "python-bulletproofs_amd", "libbpmi.so")


def main():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result",
           "-Rpass-analysis=kernel-resource-usage", "-o", LIB, SRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        sys.exit(r.returncode)
    rows, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark: .*?(Function Name|Name): (\S+)", ln)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip().split("(")[0]}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)")):
            m = re.search(pat, ln)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    print("%-34s %6s %6s %8s %4s %8s" % ("kernel", "VGPRs", "SGPRs", "scratch", "occ", "LDS"))
    for r_ in rows:
        if flt in r_["name"]:
            print("%-34s %6s %6s %8s %4s %8s" % (r_["name"][:34], r_.get("vgpr"), r_.get("sgpr"), r_.get("scratch"), r_.get("occ"), r_.get("lds")))


if __name__ == "__main__":
    main()
