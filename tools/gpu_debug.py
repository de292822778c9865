import os, sys, random, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if len(sys.argv) > 1:
    import bulletproofs_amd
    from bulletproofs_amd.engine import Engine
    from oracle import cbind, ec
    n, chunk, c = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    Q = ec.secp256k1.q
    rnd = random.Random(1)
    ks = [(12345 + 7919 * i) % Q for i in range(n)]
    pts = cbind.ec_mul_batch([ec.secp256k1.G] * n, ks)
    es = [rnd.randrange(Q) for _ in range(n)]
    eng = Engine()
    eng.set_option("chunk", chunk); eng.set_option("window_bits", c)
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    got = eng.msm_bytes(pb, sb, n)
    print("n=%d chunk=%d c=%d ->" % (n, chunk, c), "ok" if got == cbind.msm_bytes(pb, sb, n) else "WRONG", flush=True)
else:
    for args in ((1000, 4096, 0), (1000, 0, 0), (300, 2, 0), (64, 1, 0), (1000, 0, 4), (20, 1, 4), (9, 1, 4), (200, 1, 8)):
        r = subprocess.run([sys.executable, __file__] + [str(a) for a in args], capture_output=True, text=True,
                           env=dict(os.environ, BPMI_DEBUG_SYNC="1"))
        last = [l for l in r.stderr.splitlines() if l.startswith("[bpmi]")][-1:] 
        print(args, "rc", r.returncode, r.stdout.strip(), last, flush=True)
