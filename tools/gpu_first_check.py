"""First-light check of the MSM pipeline on a real GPU against the oracle."""
import importlib.util, os, sys, time, random
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd as bp
from bulletproofs_amd.engine import Engine
from oracle import cbind, ec
from oracle.bp_ref import elliptic_hash

Q = ec.secp256k1.q
eng = Engine()
rnd = random.Random(1)
G = ec.secp256k1.G
base = [elliptic_hash(b"%d" % i) for i in range(64)]

def check(pts, es, label, **opts):
    for k, v in opts.items():
        eng.set_option(k, v)
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    got = eng.msm_bytes(pb, sb, len(pts))
    want = cbind.msm_bytes(pb, sb, len(pts))
    ok = got == want
    print(("ok  " if ok else "FAIL"), label, len(pts), opts)
    for k in opts:
        eng.set_option(k, 0)
    return ok

allok = True
for n in (1, 2, 3, 5, 17, 64):
    pts = [rnd.choice(base) for _ in range(n)]
    es = [rnd.randrange(Q) for _ in range(n)]
    allok &= check(pts, es, "random")
    allok &= check(pts, es, "random tail=device", tail=1)
    allok &= check(pts, es, "random c=7", window_bits=7)
# larger, structured points P_i = P0 + i*D built by the C oracle batch mul
def big_points(n):
    ks = [(12345 + 7919 * i) % Q for i in range(n)]
    return cbind.ec_mul_batch([G] * n, ks), ks
for n in (1000, 4096, 70000):
    pts, ks = big_points(n)
    es = [rnd.randrange(Q) for _ in range(n)]
    t = time.time()
    allok &= check(pts, es, "big random")
    print("   %.2fs" % (time.time() - t))
    allok &= check(pts, [1] * n, "all ones")
    allok &= check(pts, [Q - 1] * (n // 2) + [0] * (n - n // 2), "q-1 / 0")
    allok &= check(pts, [es[0]] * n, "all same scalar")
    allok &= check([pts[0]] * n, es, "all same point")
    allok &= check(pts, es, "chunk=8", chunk=8)
    allok &= check(pts, es, "c=10", window_bits=10)
    allok &= check(pts, es, "c=16", window_bits=16)
print("ALL OK" if allok else "SOME FAILED")
sys.exit(0 if allok else 1)
