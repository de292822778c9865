#!/bin/bash
out=${1:-gpurun_out/r04i}
mkdir -p "$out"
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_ipa.py tests/test_gpu_rangeproofs.py tests/test_gpu_fuzz.py tests/test_gpu_like_reference.py -x -q -m gpu 2>&1 | tail -6
python - <<'PY' 2>&1 | tee "$out/mid_kernel_latency.txt"
import os, sys, time, hashlib
sys.path.insert(0, os.getcwd())
import bulletproofs_amd
from bulletproofs_amd.ec import secp256k1
from bulletproofs_amd.engine import default_engine
Q = secp256k1.q
eng = default_engine()
N = 1 << 14
def sha_scalars(n, seed):
    p = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(p + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))
kb = sha_scalars(N, 1); sb = sha_scalars(N, 2)
d_k = eng.upload(kb); d_G = eng.upload(secp256k1.G.to_le64() * N); d_p = eng.alloc(64 * N)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, N, d_p.ptr)); eng.sync()
d_s = eng.upload(sb)
pb = d_p.download()
print("ms per MSM one at a time | ms per PAIR (bpmi_msm2, host buffers)   [mid_min: -1 = never]")
for n in (512, 1024, 1536, 2049, 3000, 4097, 6000, 8193, 8448):
    row = []
    for mid in (-1, 1, -1, 1):
        eng.set_option("mid_min", mid); eng.set_option("mid_single_min", mid)
        for _ in range(6): eng.msm_dev(d_p, d_s, n)
        t = time.perf_counter()
        for _ in range(60): eng.msm_dev(d_p, d_s, n)
        one = (time.perf_counter() - t) / 60 * 1e3
        for _ in range(4): eng.msm2_bytes(pb, sb, n, pb, kb, n)
        t = time.perf_counter()
        for _ in range(40): eng.msm2_bytes(pb, sb, n, pb, kb, n)
        two = (time.perf_counter() - t) / 40 * 1e3
        row.append("%.3f|%.3f" % (one, two))
    print("n=%5d   without %s %s   with k_msm_mid %s %s" % (n, row[0], row[2], row[1], row[3]), flush=True)
eng.set_option("mid_min", 0); eng.set_option("mid_single_min", 0)
PY
for o in "mid_min=-1" "mid_min=0" "mid_min=-1" "mid_min=0"; do echo "== $o"; timeout 200 python tools/c3_round_times.py 20 $o 2>&1 | tail -14; done > "$out/c3_rounds.txt"; grep -E "==|total|^ +(4096|64|2) " "$out/c3_rounds.txt"
