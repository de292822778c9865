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


def hipcc_version():
    try:
        out = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout
        lines = [ln.strip() for ln in out.splitlines() if ln.strip()]
        return "; ".join(ln for ln in lines if ln.startswith(("HIP version", "AMD clang version")))[:200] or (lines[0] if lines else "unknown")
    except Exception as e:          # the record must not fail the build
        return "unknown (%s)" % type(e).__name__


def build(force=False, verbose=False, report=None):
    """Compile libbpmi.so for gfx950 when it is missing or older than a source (force: always).  `report` (a callable taking one
    line of text) is told what happened -- "compiled" with the compiler's version and the seconds it took, or "reused" with the reason
    -- so that a build record says whether anything was compiled (the prebuilt .so travels with the snapshot)."""
    import time
    report = report or (lambda line: None)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    if not force and not needs_build():
        report("libbpmi.so: reused (up to date: newer than its %d sources and build.py; sha256 %s)" % (len(deps), sha256_of(LIB)[:16]))
        return LIB
    why = "forced" if force and os.path.exists(LIB) else ("missing" if not os.path.exists(LIB) else "older than a source")
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-result", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    t0 = time.time()
    subprocess.check_call(cmd)
    report("libbpmi.so: compiled for gfx950 in %.1f s (was %s) with %s; sha256 %s" % (time.time() - t0, why, hipcc_version(), sha256_of(LIB)[:16]))
    return LIB


def sha256_of(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv, report=print)
    print(LIB)
