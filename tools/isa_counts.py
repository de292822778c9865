#!/usr/bin/env python3
"""Counts the instructions of the mixed addition (xyzz_madd, csrc/curve.hpp: the bucket update of k_accum_l0) in the
gfx950 ISA: the device code is built with -DBPMI_ISA_PROBE, which adds a probe kernel that runs exactly the main path
of the addition (xyzz_madd_pr + xyzz_madd_finish: the common case, accumulator and addend finite and different)
on operands it loads from memory; every instruction of that kernel is tallied by mnemonic (memory instructions and waits
separately).

    python tools/isa_counts.py                       # prints the JSON object
    python tools/isa_counts.py --write               # also writes profiles/r04_isa_counts.json and the ISA excerpt

bench.py reads profiles/r04_isa_counts.json for `alu_roofline.frac_vs_raw_mad` (multiply-adds actually issued per
second against the chip's raw v_mad_u64_u32 rate)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "python-bulletproofs_amd", "csrc", "bpmi.hip")
INSTR = re.compile(r"^\s+((?:v_|s_|ds_|global_|buffer_|flat_|scratch_)\w+)")


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "bpmi.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DBPMI_ISA_PROBE", "-S", "--cuda-device-only",
                               "-Wno-unused-command-line-argument", "-o", asm, SRC], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    m = re.search(r"^_Z10k_accum_l0ILb0E.*?; Occupancy: \d+", text, re.S | re.M)
    kernel = m.group(0)
    probe = re.search(r"^_Z16k_isa_probe_madd.*?; Occupancy: \d+", text, re.S | re.M).group(0)
    # the whole probe kernel is the main path plus the loads / stores of its operands and their address arithmetic;
    # memory instructions, waits and the kernel prologue / epilogue are tallied separately
    counts, other = collections.Counter(), collections.Counter()
    for ln in probe.splitlines():
        mm = INSTR.match(ln)
        if not mm:
            continue
        op = mm.group(1)
        if op.startswith(("global_", "s_load", "s_waitcnt", "s_endpgm", "buffer_", "flat_", "scratch_")):
            other[op] += 1
        else:
            counts[op] += 1
    total = sum(counts.values())
    vgpr = re.search(r"; NumVgprs: (\d+)", kernel).group(1)
    occ = re.search(r"; Occupancy: (\d+)", kernel).group(1)
    whole = sum(1 for ln in kernel.splitlines() if INSTR.match(ln))
    out = {"k_accum_l0_madd_main_path": {
        "instructions_per_madd": total,
        "v_mad_u64_u32_per_madd": counts.get("v_mad_u64_u32", 0),
        "s_nop_per_madd": counts.get("s_nop", 0),
        "by_mnemonic": dict(counts.most_common()),
        "probe_kernel_memory_and_wait_instructions": dict(other),
        "kernel_vgprs": int(vgpr), "kernel_occupancy_waves_per_simd": int(occ), "kernel_instructions_total": whole,
        "how": "hipcc -O3 --offload-arch=gfx950 -DBPMI_ISA_PROBE -S; every non-memory instruction of k_isa_probe_madd = xyzz_madd_pr + xyzz_madd_finish, the code k_accum_l0 inlines per sorted entry; the loop's own per-entry work (64-byte load, 8x32 -> 9x29 limb conversion, sign select, run-boundary test) is not in this count (tools/isa_counts.py)",
        "round_1_for_comparison": {"instructions_per_madd": 1953, "v_mad_u64_u32_per_madd": 942, "note": "same span measured on the round-1 code (commit c7275e6)"}}}
    print(json.dumps(out, indent=1))
    if "--write" in sys.argv:
        with open(os.path.join(REPO, "profiles", "r04_isa_counts.json"), "w") as f:
            json.dump(out, f, indent=1)
        with open(os.path.join(REPO, "profiles", "r04_isa_k_isa_probe_madd.s"), "w") as f:
            f.write(probe + "\n")


if __name__ == "__main__":
    main()
