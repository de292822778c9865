"""Builds libbpmi.so (gfx950 only) in-tree with hipcc.  No JIT cache: the .so sits
next to the sources so it travels with the repository snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbpmi.so")
SOURCES = ["bpmi.hip"]
HEADERS = ["field.hpp", "field_gen.hpp", "curve.hpp", "scalar.hpp", "context.hpp", "device_util.hpp", "msm_kernels.hpp",
           "point_kernels.hpp", "fold_ops_host.hpp", "scalar_kernels.hpp", "msm_host.hpp", "host_tail.hpp", "rp_batch_host.hpp", "host_pool.hpp", "rp_algebra_host.hpp", "transcript_host.hpp", "rp_wire_v2_host.hpp", "rp_batch_kernels.hpp", "rp_prove_kernels.hpp", "rp_prove_host.hpp", "scalar_gen.hpp", os.path.join("..", "..", "include", "bpmi.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-result", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(LIB)
