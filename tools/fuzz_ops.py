#!/usr/bin/env python3
"""Randomised differential test of the non-MSM entry points against the C oracle / Python
integers: bpmi_ipa_* (every round's L, R, final a, b; random lengths and fold thresholds),
bpmi_ec_lincomb2_batch, bpmi_ec_mul_batch, bpmi_ec_sum, bpmi_sc_dot, bpmi_sc_fold,
bpmi_ec_decompress_batch.   python tools/fuzz_ops.py [seconds]"""
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.engine import default_engine  # noqa: E402
from oracle import bp_ref as R, cbind  # noqa: E402
from oracle.ec import INF, secp256k1  # noqa: E402

Q = secp256k1.q
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(os.environ.get("FUZZ_SEED", "99"))
rnd = random.Random(seed)
eng = default_engine()
pool = cbind.ec_mul_batch([secp256k1.G] * 2100, [rnd.randrange(1, Q) for _ in range(2100)])
le = lambda v: (v % Q).to_bytes(32, "little")
fails = cases = 0


def check(ok, what):
    global fails, cases
    cases += 1
    if not ok:
        fails += 1
        print("MISMATCH", what, "seed", seed, "case", cases, flush=True)


def special_scalar():
    return rnd.choice((0, 1, 2, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2, rnd.randrange(Q), rnd.randrange(1 << 64)))


t0 = time.time()
while time.time() - t0 < budget:
    op = rnd.randrange(7)
    if op == 0:      # IPA
        n = 1 << rnd.randrange(0, 11)
        eng.set_option("ipa_big_m", rnd.choice((0, 32, 64, 256, 1024)))
        eng.set_option("ipa_small_m", rnd.choice((0, 0, 1, 2, 4, 8, 32, 128)))        # the product fold at that length (round 4)
        eng.set_option("small_pair", rnd.choice((1, 1, 0)))
        eng.set_option("ipa_small_step", rnd.choice((0, 0, 1)))
        eng.set_option("fold_wnaf", rnd.choice((2, 2, 1, 0)))
        eng.set_option("pair_phases", rnd.choice((0, 1)))
        pts = [rnd.choice(pool) for _ in range(2 * n + 1)]
        g, h, u = pts[:n], pts[n:2 * n], pts[2 * n]
        a = [special_scalar() for _ in range(n)]
        b = [special_scalar() for _ in range(n)]
        st = eng.ipa_create(cbind.pack_points(g), cbind.pack_points(h), cbind.pack_scalars(a), cbind.pack_scalars(b), n, cbind.pack_points([u]))
        ok = True
        while len(st) > 1:
            half = len(st) // 2
            L, Rr = st.round_LR()
            cl, cr = cbind.sc_dot(a[:half], b[half:]), cbind.sc_dot(a[half:], b[:half])
            ok &= L == cbind.pack_points([cbind.msm(g[half:] + h[:half] + [u], a[:half] + b[half:] + [cl])])
            ok &= Rr == cbind.pack_points([cbind.msm(g[:half] + h[half:] + [u], a[half:] + b[:half] + [cr])])
            x = rnd.randrange(1, Q)
            xi = pow(x, -1, Q)
            st.fold(x, xi)
            g = cbind.ec_lincomb2_batch(g[:half], g[half:], xi, x)
            h = cbind.ec_lincomb2_batch(h[:half], h[half:], x, xi)
            a = cbind.sc_fold(a[:half], a[half:], x, xi)
            b = cbind.sc_fold(b[:half], b[half:], xi, x)
        ok &= st.finish() == (a[0], b[0])
        st.close()
        eng.set_option("ipa_big_m", 0)
        eng.set_option("ipa_small_m", 0)
        eng.set_option("small_pair", 1)
        eng.set_option("ipa_small_step", 0)
        eng.set_option("fold_wnaf", 2)
        eng.set_option("pair_phases", 0)
        check(ok, "ipa n=%d" % n)
    elif op == 1:    # lincomb2
        n = rnd.randrange(1, 400)
        p1 = [rnd.choice(pool) if rnd.random() > 0.05 else INF for _ in range(n)]
        p2 = [rnd.choice((rnd.choice(pool), p1[i], -p1[i])) for i in range(n)]
        k1, k2 = special_scalar(), special_scalar()
        got = eng.ec_lincomb2_batch_bytes(cbind.pack_points(p1), cbind.pack_points(p2), le(k1), le(k2), n)
        check(got == cbind.pack_points(cbind.ec_lincomb2_batch(p1, p2, k1, k2)), "lincomb2 n=%d" % n)
    elif op == 2 and rnd.random() < 0.04:    # mul batch, the GLV fixed-window path (n >= 32 768): against the bit-serial ladder + oracle samples
        lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
        n = rnd.randrange(32768, 70000)
        ps = [rnd.choice(pool) if rnd.random() > 0.01 else INF for _ in range(n)]
        ks = [rnd.choice((special_scalar(), (special_scalar() * lam + special_scalar()) % Q, rnd.randrange(Q), (1 << rnd.randrange(256)) % Q)) for _ in range(n)]
        pb, kb = cbind.pack_points(ps), cbind.pack_scalars(ks)
        got = eng.ec_mul_batch_bytes(pb, kb, n)
        eng.set_option("mul_batch_glv", 0)
        ref = eng.ec_mul_batch_bytes(pb, kb, n)
        eng.set_option("mul_batch_glv", 1)
        idx = [rnd.randrange(n) for _ in range(64)]
        want = cbind.pack_points(cbind.ec_mul_batch([ps[i] for i in idx], [ks[i] for i in idx]))
        check(got == ref and b"".join(got[64 * i: 64 * i + 64] for i in idx) == want, "mul_batch glv n=%d" % n)
    elif op == 2:    # mul batch
        n = rnd.randrange(1, 400)
        ps = [rnd.choice(pool) if rnd.random() > 0.05 else INF for _ in range(n)]
        ks = [special_scalar() for _ in range(n)]
        got = eng.ec_mul_batch_bytes(cbind.pack_points(ps), cbind.pack_scalars(ks), n)
        check(got == cbind.pack_points(cbind.ec_mul_batch(ps, ks)), "mul_batch n=%d" % n)
    elif op == 3:    # ec_sum
        n = rnd.randrange(1, 1500)
        ps = [rnd.choice(pool) for _ in range(n)]
        if rnd.random() < 0.3:
            ps = ps[: n // 2] + [-p for p in ps[: n // 2]]
        want = INF
        for p in ps:
            want = want + p
        check(eng.ec_sum_bytes(cbind.pack_points(ps), len(ps)) == cbind.pack_points([want]), "ec_sum n=%d" % len(ps))
    elif op == 4:    # sc_dot
        n = rnd.randrange(1, 5000)
        a = [special_scalar() for _ in range(n)]
        b = [special_scalar() for _ in range(n)]
        got = int.from_bytes(eng.sc_dot_bytes(cbind.pack_scalars(a), cbind.pack_scalars(b), n), "little")
        check(got == sum(x * y for x, y in zip(a, b)) % Q, "sc_dot n=%d" % n)
    elif op == 5:    # sc_fold
        n = rnd.randrange(1, 3000)
        a = [special_scalar() for _ in range(n)]
        b = [special_scalar() for _ in range(n)]
        x, y = special_scalar(), special_scalar()
        got = eng.sc_fold_bytes(cbind.pack_scalars(a), cbind.pack_scalars(b), le(x), le(y), n)
        check(got == cbind.pack_scalars([(x * u + y * v) % Q for u, v in zip(a, b)]), "sc_fold n=%d" % n)
    else:            # decompress
        n = rnd.randrange(1, 600)
        ps = [rnd.choice(pool) for _ in range(n)]
        comp = b"".join(R.point_to_bytes(p) for p in ps)
        out, ok = eng.ec_decompress_batch_bytes(comp, n)
        check(out == cbind.pack_points(ps) and ok == bytes([1]) * n, "decompress n=%d" % n)
print("fuzz_ops: %d cases, %d mismatches, %.0f s, seed %d" % (cases, fails, time.time() - t0, seed))
sys.exit(1 if fails else 0)
