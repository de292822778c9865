"""MSM at n = 2^20 through the HOST-buffer entry point (bpmi_msm: 96 MiB over PCIe per call)
vs the device-resident entry point (bpmi_msm_dev) -- the rate DESIGN.md section 7 quotes as
'PCIe-inclusive'; never the bench value."""
import os, sys, time, random
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
Q = secp256k1.q
eng = default_engine()
n = 1 << 20
rnd = random.Random(3)
ks = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(n))
pts = eng.ec_mul_batch_bytes(secp256k1.G.to_le64() * n, ks, n)
es = b"".join(rnd.randrange(Q).to_bytes(32, "little") for _ in range(n))
d_p, d_s = eng.upload(pts), eng.upload(es)
r0 = eng.msm_dev(d_p, d_s, n)
assert eng.msm_bytes(pts, es, n) == r0
for name, f in (("device-resident (bpmi_msm_dev)", lambda: eng.msm_dev(d_p, d_s, n)), ("host buffers (bpmi_msm)", lambda: eng.msm_bytes(pts, es, n))):
    t = time.perf_counter()
    for _ in range(10): f()
    dt = (time.perf_counter() - t) / 10
    print("%-34s %.3f ms  %.3e pairs/s" % (name, dt * 1e3, n / dt), flush=True)
