"""Constants, synthetic inputs and the host budget shared by the benchmark's modules."""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# The two MSM lanes must sit on DIFFERENT hardware queues to overlap.  HIP maps streams onto GPU_MAX_HW_QUEUES (default 4)
# queues round-robin in creation order; once RCCL and the framework have created theirs, both lanes can land on one queue and
# the pipeline degrades to the one-lane rate (measured: 1.27 ms per step against 1.13 with 8 queues, profiles/r02_hw_queues.txt).
# Must be set before the HIP runtime initialises, i.e. before torch is imported.
# Round 6: 16.  The batch verifier keeps eight batches in flight, each on an engine with two or three streams of its own; on eight queues
# those streams share queues and a kernel waits behind another batch's kernel on its queue: C5 +10-30 % (wire format 1), +5-16 % (format 3)
# with sixteen, the MSM pipeline, C2, C3 and the prover unchanged (profiles/r06_hw_queues_ab.txt).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

Q = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
PCIE_PEAK_GBS = 63.0           # MI355X_MICROARCH.md: host link PCIe Gen5 x16, 63 GB/s (spec), one direction
ALGO_BYTES_PER_PAIR = 96       # 32-B scalar + 64-B affine point
IPA_ALGO_BYTES_PER_ELEMENT = 960   # whole proof, per element of the n-vector
# multiply-add content of one bucket update (xyzz_madd, csrc/curve.hpp): v_mad_u64_u32 per wave-lane, counted in
# the ISA of k_accum_l0 (profiles/r04_isa_counts.json); and the chip's raw rate for that instruction
MADS_PER_MADD = None           # filled from profiles/r04_isa_counts.json when present
RAW_MAD_TOPS = 28.85           # T lane-ops/s, tools/fe_microbench.hip (profiles/r01_fe_microbench.txt)
MULS_PER_MADD = 10.5           # 8M + 2S plus carries/subtractions in multiplication-equivalents
FE_MUL_PEAK_G = 221.4          # the product's own fe_mul in isolation, G multiplications/s (profiles/r03_fe_microbench.txt, V8; round 1's fe_mul: 196)


def synth_scalars(n, seed):
    """e_i = SHA-256("bpmi/scalar" || seed || LE64(i)) mod q -> (bytes, list of ints)."""
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    out = bytearray(32 * n)
    vals = [0] * n
    sha = hashlib.sha256
    for i in range(n):
        v = int.from_bytes(sha(pre + i.to_bytes(8, "little")).digest(), "big") % Q
        vals[i] = v
        out[32 * i: 32 * i + 32] = v.to_bytes(32, "little")
    return bytes(out), vals


def isa_counts():
    try:
        for name in ("r05_isa_counts.json", "r04_isa_counts.json", "r03_isa_counts.json", "r02_isa_counts.json"):
            path = os.path.join(REPO, "profiles", name)
            if os.path.exists(path):
                with open(path) as f:
                    return json.load(f)["k_accum_l0_madd_main_path"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def committed_traffic(logn):
    """Per-launch HBM bytes of the dominant kernel from the committed rocprofv3 --pmc passes (NOT measured in this run):
    (raw bytes, guide-corrected bytes, source).  The RAW figure (FETCH_SIZE + WRITE_SIZE as counted) is the one reported as
    `roofline.traffic`: the guide's x2 correction of FETCH_SIZE is calibrated on wide coalesced streaming reads (128-B requests tallied
    at 64 B), and this kernel's reads are 64-B gathers -- the bytes it REQUESTS are 1.14 GB per launch, below the corrected 3.25 GB and
    consistent with the raw 1.67 GB.  The corrected figure is reported beside it."""
    for name in ("r06_pmc_traffic_msm_n2e20.json", "r05_pmc_traffic_msm_n2e20.json", "r04_pmc_traffic_msm_n2e20.json", "r03_pmc_traffic_msm_n2e20.json",
                 "r02_pmc_traffic_msm_n2e20.json", "r01_pmc_traffic_msm_n2e20.json"):
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                for row in json.load(f)["kernels"]:
                    if row["kernel"] == "k_accum_l0" and logn == 20:
                        return (row.get("hbm_bytes_per_launch_raw"), row["hbm_bytes_per_launch_guide_corrected"],
                                "profiles/%s (FETCH_SIZE + WRITE_SIZE, separate --pmc passes; committed file, not this run)" % name)
        except (OSError, KeyError, ValueError):
            pass
    return None, None, None


# ---- extra: config C5, batch verification of 2^14 64-bit range proofs -----------------------------


def usable_cpus():
    usable = len(os.sched_getaffinity(0))
    q = cpu_quota()
    if q:
        usable = max(1, min(usable, int(q)))
    return usable


def cpu_quota():
    """CPUs this container may use at once (cgroup v2 cpu.max), or None when unlimited."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except Exception:
        return None
