"""Batch verification of range proofs (single or aggregated) that share (g, h, u, gs, hs):
every proof's four EC verification equations are combined with fresh random weights into
ONE multi-scalar multiplication that must evaluate to the identity (SURVEY.md section
8f item 3; not present in the reference, which verifies one proof at a time).

For proof k (notation of the reference: rangeproof_verifier.py:55-99,
inner_product_verifier.py:44-58,127-147), with hsp_i = y^-i * hs_i folded into scalars:

  E1  (t_hat - delta) g + taux h - z^2 V - x T1 - x^2 T2                         = 0
      (aggregated, m values: - sum_j z^(j+2) V_j, and z^(2 + i//n) 2^(i%n) in E2;
       rangeproof_aggreg_verifier.py:55-108)
  E2  P_new - A - x S + sum_i z gs_i - sum_i (z y^i + z^2 2^i) y^-i hs_i
        + mu h - (x_ip t_hat) u                                                  = 0
  E3  u_new - x_ip u                                                             = 0
  E4  sum_i (a s_i) gs_i + sum_i (b s_i^-1 y^-i) hs_i + (a b) u_new
        - P_new - sum_j (x_j^2 L_j + x_j^-2 R_j)                                 = 0

E1 is the range-proof polynomial check, E2 / E3 are Verifier1's two point equalities, E4
is Verifier2's final check.  All byte-level transcript checks of the three verifiers are
performed exactly as the individual verifiers do (host side) when a proof is added.
With independent uniform weights w1..w4 per proof, a batch containing any failing equation
passes with probability about 1/q.  Cost: 19 points per proof + 2n + 3 shared points in
one MSM, instead of ~8 latency-bound GPU calls per proof.

The shards of a multi-GPU batch are disjoint sets of proofs: every rank builds the MSM of
its own proofs (with its own share of the shared-generator coefficients) and the 64-byte
partial results are combined with one all_gather + fold (distributed.ShardedMSM.combine).
"""
import secrets

from .. import engine as _engine
from ..ec import Point, secp256k1
from ..innerproduct.inner_product_verifier import Verifier1, Verifier2
from .rangeproof_verifier import RangeVerifier

Q = secp256k1.q
_ZERO64 = bytes(64)


def _le32(v):
    return (v % Q).to_bytes(32, "little")


class BatchRangeVerifier:
    def __init__(self, g, h, gs, hs, u, msm=None, rng=None, engine=None):
        """msm(points_bytes, scalar_bytes, n) -> 64 bytes; default: the HIP engine.
        rng() -> int: source of the random weights; default: secrets (CSPRNG).
        engine: the engine.Engine this verifier's GPU work runs on (default: the process-wide one).  Verifiers with
        engines of their own can work from different threads at the same time -- the library calls release the GIL
        -- so the upload of one batch overlaps the kernels of another."""
        assert len(gs) == len(hs)
        self.g, self.h, self.gs, self.hs, self.u = g, h, gs, hs, u
        self.n = len(gs)
        self._msm = msm
        self._engine = engine
        self._rng = rng or (lambda: secrets.randbits(320))
        self._custom_rng = rng is not None
        self._shared_pts = g.to_le64() + h.to_le64() + u.to_le64() + b"".join(p.to_le64() for p in gs) + \
            b"".join(p.to_le64() for p in hs)
        self.reset()

    def _eng(self):
        return self._engine or _engine.default_engine()

    def reset(self):
        self.c_g = self.c_h = self.c_u = 0
        self._gs_const = self._hs_const = 0   # terms that are equal for every generator index
        self.c_gs = [0] * self.n
        self.c_hs = [0] * self.n
        self._pts = []          # per-proof points, 64-byte strings
        self._scs = []          # matching scalars (ints mod q)
        self._raw_pts, self._raw_scs, self._raw_count = [], [], 0     # merged states: already packed
        # device-resident batches of add_wire_native: the buffers are kept for the next batch of the same shape (hipMalloc /
        # hipFree per batch cost more than they look: hipFree waits for the whole device, which also stalls another
        # verifier's batch in flight); release() returns them
        spare = getattr(self, "_spare", [])
        for d_p, d_s, _ in getattr(self, "_dev_chunks", ()):
            if len(spare) < 2:
                spare.append((d_p, d_s))
            else:
                d_p.free()
                d_s.free()
        self._spare = spare
        self._dev_chunks = []
        self.count = 0

    def release(self):
        """Free every device buffer this verifier holds (spares of reset(), the cached shared generators)."""
        self.reset()
        for d_p, d_s in self._spare:
            d_p.free()
            d_s.free()
        self._spare = []
        for name in ("_d_shared_pts", "_d_shared_scs"):
            buf = getattr(self, name, None)
            if buf is not None:
                buf.free()
                setattr(self, name, None)
        bufs = getattr(self, "_wire_bufs", None)
        if bufs is not None:
            bufs[1].free()
            bufs[2].free()
            self._wire_bufs = None

    def _chunk_buffers(self, eng, npairs):
        for i, (d_p, d_s) in enumerate(self._spare):
            if d_p.nbytes == 64 * npairs and d_s.nbytes == 32 * npairs and d_p.engine is eng:
                del self._spare[i]
                return d_p, d_s
        return eng.alloc(64 * npairs), eng.alloc(32 * npairs)

    def _weight(self):
        w = self._rng() % Q
        return w or 1

    def add(self, V, proof):
        """Host-side transcript checks of RangeVerifier / Verifier1 / Verifier2 (raise
        Exception("Proof invalid") exactly as they do), then accumulate the proof's
        weighted equations.  `V` is one commitment (single proof over all len(gs) bits) or
        a list of m commitments (aggregated proof, len(gs) = n * m)."""
        nm, q = self.n, Q
        Vs = list(V) if isinstance(V, (list, tuple)) else [V]
        aggregated = isinstance(V, (list, tuple))
        m = len(Vs)
        rv = RangeVerifier(Vs[0], self.g, self.h, self.gs, self.hs, self.u, proof)
        rv.assertThat(m >= 1 and nm % m == 0)
        n = nm // m
        rv.verify_transcript()                               # A, S, T1, T2 bytes; reads y, z, x
        x, y, z = rv.x.x % q, rv.y.x % q, rv.z.x % q
        ip = proof.innerProof
        v1 = Verifier1(self.gs, None, self.u, None, proof.t_hat, ip)
        v1.verify_transcript()                               # x_ip = H(transcript)
        x_ip = int(ip.transcript.split(b"&")[1]) % q
        p2 = ip.proof2
        Verifier2(self.gs, None, None, None, p2).verify_transcript()   # L, R bytes; every x_j re-hashed
        log_n = nm.bit_length() - 1
        rv.assertThat(len(p2.xs) == log_n and len(p2.Ls) == log_n and len(p2.Rs) == log_n)
        xs = [xj.x % q for xj in p2.xs]
        a, b = p2.a.x % q, p2.b.x % q
        t_hat, taux, mu = proof.t_hat.x % q, proof.taux.x % q, proof.mu.x % q

        w1, w2, w3, w4 = self._weight(), self._weight(), self._weight(), self._weight()
        # one modular inversion for (x_1 .. x_k, y) (Montgomery's trick)
        vals = xs + [y]
        pre, acc = [], 1
        for v in vals:
            pre.append(acc)
            acc = acc * v % q
        rv.assertThat(acc != 0)                               # a zero challenge has no inverse: a bad proof, as on the native path
        inv = pow(acc, -1, q)
        invs = [0] * len(vals)
        for k in range(len(vals) - 1, -1, -1):
            invs[k] = inv * pre[k] % q
            inv = inv * vals[k] % q
        xinvs, yinv = invs[:-1], invs[-1]
        # s-vector by doubling (get_ss, inner_product_verifier.py:91-102) with the weights
        # folded in:  sg_i = w4 a s_i   and   sh_i = w4 b s_i^-1 y^-i  (the step that creates
        # index bit k multiplies the bit-1 half by y^-(2^k) as well)
        sg, sh = [w4 * a % q], [w4 * b % q]
        ypow2 = yinv
        for xv, xi in zip(reversed(xs), reversed(xinvs)):
            hi_g, hi_h = xv, xi * ypow2 % q
            sg = [s_ * xi % q for s_ in sg] + [s_ * hi_g % q for s_ in sg]
            sh = [s_ * xv % q for s_ in sh] + [s_ * hi_h % q for s_ in sh]
            ypow2 = ypow2 * ypow2 % q
        zpow = [pow(z, 2 + j, q) for j in range(m)]          # z^2 for a single proof
        w2z = w2 * z % q
        c_gs, c_hs = self.c_gs, self.c_hs
        self._gs_const = (self._gs_const + w2z) % q          # the same w2 z on every gs_i ...
        self._hs_const = (self._hs_const - w2z) % q          # ... and -w2 z on every hs_i
        r = 2 * yinv % q                                     # (2/y)^i runs geometrically inside a value block
        yn_inv = pow(yinv, n, q)
        blk = 1                                              # y^-(n j)
        i = 0
        for j in range(m):
            geo = w2 * zpow[j] % q * blk % q                 # w2 z^(2+j) 2^(i%n) y^-i  at i = n j
            for _ in range(n):
                c_gs[i] += sg[i]
                c_hs[i] += sh[i] - geo
                geo = geo * r % q
                i += 1
            blk = blk * yn_inv % q
        # sum_{i<nm} y^i by doubling
        ysum, ypw, length = 1, y, 1
        while length < nm:
            ysum = ysum * (1 + ypw) % q
            ypw = ypw * ypw % q
            length *= 2
        if aggregated:
            delta = ((z - z * z) * ysum - sum(pow(z, j + 2, q) for j in range(1, m + 1)) * ((1 << n) - 1)) % q
        else:
            delta = ((z - z * z) * ysum - pow(z, 3, q) * ((1 << n) - 1)) % q
        self.c_g = (self.c_g + w1 * (t_hat - delta)) % q
        self.c_h = (self.c_h + w1 * taux + w2 * mu) % q
        self.c_u = (self.c_u - w2 * x_ip % q * t_hat - w3 * x_ip) % q
        pts = Vs + [proof.T1, proof.T2, proof.A, proof.S, ip.P_new, ip.u_new] + list(p2.Ls) + list(p2.Rs)
        scs = [-w1 * zp for zp in zpow] + [-w1 * x, -w1 * x % q * x, -w2, -w2 * x, w2 - w4, w3 + w4 * a % q * b]
        scs += [-w4 * xv % q * xv for xv in xs] + [-w4 * xi % q * xi for xi in xinvs]
        self._pts.append(b"".join(p.to_le64() for p in pts))
        self._scs.extend(scs)
        self.count += 1

    # ---- many proofs in wire format, host work spread over worker processes ---------------
    def state(self):
        """Everything add() / add_wire*() has accumulated, as plain picklable data (see merge).  Device-resident chunks of
        add_wire_native (points and scalars of the proofs, which never came to the host) are downloaded for it."""
        dev_pts, dev_scs, dev_n = [], [], 0
        for d_p, d_s, cnt in self._dev_chunks:
            dev_pts.append(d_p.download()[:64 * cnt])
            dev_scs.append(d_s.download()[:32 * cnt])
            dev_n += cnt
        if dev_n:
            return (self.c_g, self.c_h, self.c_u, self._gs_const, self._hs_const, list(self.c_gs), list(self.c_hs),
                    b"".join(self._pts) + b"".join(self._raw_pts) + b"".join(dev_pts),
                    b"".join(_le32(v) for v in self._scs) + b"".join(self._raw_scs) + b"".join(dev_scs),
                    len(self._scs) + self._raw_count + dev_n, self.count)
        return (self.c_g, self.c_h, self.c_u, self._gs_const, self._hs_const, list(self.c_gs), list(self.c_hs),
                b"".join(self._pts) + b"".join(self._raw_pts),
                b"".join(_le32(v) for v in self._scs) + b"".join(self._raw_scs), len(self._scs) + self._raw_count, self.count)

    def merge(self, st):
        """Add another verifier's state() (same generators) to this one: the combination is
        linear, so shards of a batch can be prepared anywhere and summed."""
        c_g, c_h, c_u, gk, hk, c_gs, c_hs, pts, scs, nsc, count = st
        assert len(c_gs) == self.n and len(pts) == 64 * nsc and len(scs) == 32 * nsc
        self.c_g = (self.c_g + c_g) % Q
        self.c_h = (self.c_h + c_h) % Q
        self.c_u = (self.c_u + c_u) % Q
        self._gs_const = (self._gs_const + gk) % Q
        self._hs_const = (self._hs_const + hk) % Q
        for i in range(self.n):
            self.c_gs[i] += c_gs[i]
            self.c_hs[i] += c_hs[i]
        self._raw_pts.append(pts)
        self._raw_scs.append(scs)
        self._raw_count += nsc
        self.count += count

    def start_workers(self, workers):
        """A pool of `workers` processes for add_wire (spawned, so they never inherit a GPU
        context; they do integer and hash work only)."""
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        self.stop_workers()
        gens = (self.g.to_le64(), self.h.to_le64(), self.u.to_le64(), b"".join(p.to_le64() for p in self.gs),
                b"".join(p.to_le64() for p in self.hs))
        self._pool = ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"),
                                         initializer=_worker_init, initargs=(gens,))
        self._workers = workers
        list(self._pool.map(_worker_ping, range(workers)))      # imports done before the first batch
        return self

    def stop_workers(self):
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown()
        self._pool, self._workers = None, 0

    def add_wire(self, Vs, blobs, decompress=None, chunk=128):
        """Add len(blobs) single-value proofs given in wire format (codec.proof_to_bytes) with
        their commitments Vs (Points).  All points of all proofs are decompressed in one GPU
        launch; parsing, the transcript checks and the scalar algebra of add() run in the
        worker pool (start_workers) when there is one, else in this process.
        decompress(comp_bytes, n) -> (points_bytes, ok_flags): default the HIP engine's."""
        from .codec import compressed_points
        assert len(Vs) == len(blobs)
        comp = [compressed_points(b) for b in blobs]
        counts = [len(c) // 33 for c in comp]
        dec = decompress or self._eng().ec_decompress_batch_bytes
        pts, ok = dec(b"".join(comp), sum(counts))
        if 0 in bytes(ok):
            raise Exception("Proof invalid")
        jobs, pos = [], 0
        for lo in range(0, len(blobs), chunk):
            hi = min(lo + chunk, len(blobs))
            npts = sum(counts[lo:hi])
            jobs.append((blobs[lo:hi], pts[64 * pos: 64 * (pos + npts)], b"".join(V.to_le64() for V in Vs[lo:hi])))
            pos += npts
        pool = getattr(self, "_pool", None)
        if pool is None:
            _worker_init((self.g.to_le64(), self.h.to_le64(), self.u.to_le64(), b"".join(p.to_le64() for p in self.gs),
                          b"".join(p.to_le64() for p in self.hs)))
            results = [_worker_add(*job) for job in jobs]
        else:
            results = list(pool.map(_worker_add, *zip(*jobs))) if jobs else []
        for st in results:
            self.merge(st)

    def add_wire_native(self, Vs, blobs, decompress=None, threads=None, offsets=None, prepare="auto"):
        """add_wire with the per-proof host work in native code (bpmi_rp_batch_prepare, csrc/
        rp_batch_host.hpp: parsing, the three transcript checks, the weighted scalars; `threads` host
        threads): ~150 us of interpreter per proof become a few microseconds, and nothing in this
        function loops over proofs in Python.  Vs: one commitment per proof, or -- aggregated proofs --
        one list of m commitments per proof (the same m for the whole call), or the commitments already
        packed as count * m 64-byte points (bytes).  Same verdicts as add() except that numbers in
        transcripts must be canonical decimal.

        blobs: a list of wire proofs, or ONE bytes-like object holding them back to back together with
        `offsets` (count + 1 positions) -- how proofs arrive from a socket; saves the join of the list.  The object may
        be an engine.HostBuffer (page-locked memory, Engine.host_alloc): the upload of the device preparation then runs
        at link speed.
        With the default engine the decoded points never leave the GPU: they are decompressed straight into
        the point array of the batch's MSM (bpmi_ec_decompress_batch_dev), and the random weights are
        derived natively from one fresh 32-byte seed.

        prepare: "device" -- the preparation itself runs on the GPU too (bpmi_rp_batch_prepare_dev, csrc/rp_batch_kernels.hpp:
        one lane per proof; the wire bytes are uploaded once and nothing but the 5 + 2n shared coefficients and the verdict
        comes back); "host" -- bpmi_rp_batch_prepare on `threads` host threads; "auto" (default) -- device whenever the
        default engine is in use.  Both produce the same numbers for the same weights."""
        import ctypes
        import os
        from itertools import accumulate
        from .. import _native
        if offsets is None:
            count = len(blobs)
            offsets = [0, *accumulate(map(len, blobs))]
            joined = b"".join(blobs)
        else:
            count = len(offsets) - 1
            joined = blobs
            # the table comes from the caller: it must stay inside the buffer it indexes (the native code checks it against the
            # length it is TOLD, so that length must be the real one)
            total = joined.nbytes if hasattr(joined, "ptr") else len(joined)
            # (first / last here; the native code walks the whole table against the real size -- no Python loop over 2^14 proofs)
            if count < 0 or offsets[0] < 0 or (count > 0 and not (0 <= offsets[count] <= total)):
                raise ValueError("offsets must be non-decreasing positions inside the %d-byte proof buffer" % total)
        if not count:
            return
        k = self.n.bit_length() - 1
        npts = count * (6 + 2 * k)
        if isinstance(Vs, (bytes, bytearray, memoryview)):          # commitments already packed: count * m points of 64 bytes
            vbytes = bytes(Vs)
            m = len(vbytes) // (64 * count)
            if m < 1 or len(vbytes) != 64 * count * m or self.n % m:
                raise Exception("Proof invalid")
        else:
            assert len(Vs) == count
            aggregated = isinstance(Vs[0], (list, tuple))
            m = len(Vs[0]) if aggregated else 1
            if aggregated:
                if any(len(v) != m for v in Vs) or m < 1 or self.n % m:
                    raise Exception("Proof invalid")
                Vs = [V for group in Vs for V in group]
            vbytes = b"".join([V.to_le64() for V in Vs])
        on_device = decompress is None and self._msm is None
        weights = seed = None
        if getattr(self, "_custom_rng", False):
            weights = b"".join(self._weight().to_bytes(32, "little") for _ in range(4 * count))
        else:
            seed = os.urandom(32)               # four 248-bit weights per proof are derived from it natively
        offs = offsets if isinstance(offsets, ctypes.Array) else (ctypes.c_uint64 * (count + 1))(*offsets)
        if prepare == "auto":
            prepare = "device" if on_device else "host"
        if prepare == "device":
            if not on_device:
                raise ValueError("prepare='device' needs the default engine (no custom msm / decompress)")
            return self._add_wire_device(vbytes, joined, offs, count, m, k, weights, seed)
        bufs = getattr(self, "_native_bufs", None)
        if bufs is None or bufs[0] != (count, m, npts):      # scratch of a previous call of the same shape is reused (no zero-fill)
            bufs = ((count, m, npts), (ctypes.c_char * (32 * count * m))(), (ctypes.c_char * (32 * npts))(),
                    (ctypes.c_char * (32 * (5 + 2 * self.n)))(), (ctypes.c_char * (33 * npts))())
            self._native_bufs = bufs
        _, v_sc, p_sc, shared, comp = bufs
        bad = ctypes.c_int64(-1)
        if threads is None:
            threads = min(32, len(os.sched_getaffinity(0)))
        if hasattr(joined, "ptr"):
            joined = joined.view[:offsets[count]]
        jbuf = (ctypes.c_char * len(joined)).from_buffer_copy(joined) if isinstance(joined, memoryview) else joined
        rc = _native.load().bpmi_rp_batch_prepare(self.n, m, count, jbuf, len(joined), ctypes.cast(offs, ctypes.c_void_p), weights, seed, threads,
                                                  v_sc, p_sc, shared, comp, ctypes.cast(ctypes.pointer(bad), ctypes.c_void_p))
        if rc != 0:
            raise Exception("bpmi_rp_batch_prepare failed (%d)" % rc)
        if bad.value >= 0:
            raise Exception("Proof invalid")
        if on_device:
            eng = self._eng()
            nv = count * m
            d_pts, d_scs = self._chunk_buffers(eng, nv + npts)
            d_pts.upload(vbytes)
            ok = ctypes.create_string_buffer(npts)
            eng._ck(eng.lib.bpmi_ec_decompress_batch_dev(eng.ctx, ctypes.cast(comp, ctypes.c_void_p), npts, d_pts.ptr + 64 * nv, ctypes.cast(ok, ctypes.c_void_p)))
            if 0 in ok.raw:
                raise Exception("Proof invalid")
            eng._ck(eng.lib.bpmi_upload(eng.ctx, d_scs.ptr, ctypes.cast(v_sc, ctypes.c_char_p), 32 * nv))
            eng._ck(eng.lib.bpmi_upload(eng.ctx, d_scs.ptr + 32 * nv, ctypes.cast(p_sc, ctypes.c_char_p), 32 * npts))
            self._dev_chunks.append((d_pts, d_scs, nv + npts))
        else:
            dec = decompress or self._eng().ec_decompress_batch_bytes
            pts, ok = dec(comp.raw, npts)
            if 0 in bytes(ok):
                raise Exception("Proof invalid")
            self._raw_pts.append(vbytes)
            self._raw_scs.append(v_sc.raw)
            self._raw_pts.append(pts)
            self._raw_scs.append(p_sc.raw)
            self._raw_count += count * m + npts
        self._absorb_shared(shared.raw, count)

    def _add_wire_device(self, vbytes, joined, offs, count, m, k, weights, seed):
        """add_wire_native with the preparation on the GPU: one call uploads the wire bytes, prepares every proof, decodes its
        points into the batch's point array and returns the shared coefficients and the verdict."""
        import ctypes
        eng = self._eng()
        nv, npts = count * m, count * (6 + 2 * k)
        d_pts, d_scs = self._chunk_buffers(eng, nv + npts)
        try:
            d_pts.upload(vbytes)
            shared = ctypes.create_string_buffer(32 * (5 + 2 * self.n))
            bad = ctypes.c_int64(-1)
            if isinstance(joined, bytes):
                src = joined
            elif hasattr(joined, "ptr"):                        # engine.HostBuffer: page-locked receive buffer
                src = joined.ptr
            else:
                try:
                    src = ctypes.addressof((ctypes.c_char * len(joined)).from_buffer(joined))
                except TypeError:                               # read-only buffer
                    src = bytes(joined)
            nbytes = joined.nbytes if hasattr(joined, "ptr") else len(joined)       # the buffer's REAL size: the native bounds check of the offset table is against it
            eng._ck(eng.lib.bpmi_rp_batch_prepare_dev(eng.ctx, self.n, m, count, src, nbytes, ctypes.cast(offs, ctypes.c_void_p), weights, seed,
                                                      d_scs.ptr, d_scs.ptr + 32 * nv, d_pts.ptr + 64 * nv, shared,
                                                      ctypes.cast(ctypes.pointer(bad), ctypes.c_void_p)))
            if bad.value >= 0:
                raise Exception("Proof invalid")
        except BaseException:
            d_pts.free()
            d_scs.free()
            raise
        self._dev_chunks.append((d_pts, d_scs, nv + npts))
        self._absorb_shared(shared.raw, count)

    def partial_wire(self, Vs, blobs, offsets=None):
        """The 64-byte value of ONE batch of wire proofs, everything in one native call (bpmi_rp_batch_verify_dev): upload in
        slices with the point decoding beside it, GPU preparation, the shared coefficients folded on the device, one MSM -- no
        host round trip between the preparation and the MSM and no Python loop over anything.  Stand-alone: nothing is added to
        this verifier's accumulators (a verifier that also holds proofs added otherwise folds the two values with ec_sum).
        Vs / blobs / offsets as for add_wire_native (commitments packed -- bytes or a page-locked HostBuffer --, or Points / lists of
        Points).  Raises
        Exception("Proof invalid") when a proof fails its byte-level checks or has an invalid point; the MSM's verdict is the
        returned value (64 zero bytes = valid)."""
        import ctypes
        import os
        from itertools import accumulate
        if self._msm is not None:
            raise ValueError("partial_wire needs the default engine (no custom msm)")
        if offsets is None:
            count = len(blobs)
            offsets = [0, *accumulate(map(len, blobs))]
            joined = b"".join(blobs)
        else:
            count = len(offsets) - 1
            joined = blobs
            total = joined.nbytes if hasattr(joined, "ptr") else len(joined)
            if count < 0 or offsets[0] < 0 or (count > 0 and not (0 <= offsets[count] <= total)):
                raise ValueError("offsets must be non-decreasing positions inside the %d-byte proof buffer" % total)
        if not count:
            return _ZERO64
        k = self.n.bit_length() - 1
        if hasattr(Vs, "ptr") and hasattr(Vs, "nbytes"):           # a page-locked HostBuffer (engine.host_alloc): uploaded without a staging copy
            vbytes, vlen = Vs.ptr, Vs.nbytes
            m = vlen // (64 * count)
            if m < 1 or vlen != 64 * count * m or self.n % m:
                raise Exception("Proof invalid")
        elif isinstance(Vs, (bytes, bytearray, memoryview)):
            vbytes = bytes(Vs)
            m = len(vbytes) // (64 * count)
            if m < 1 or len(vbytes) != 64 * count * m or self.n % m:
                raise Exception("Proof invalid")
        else:
            assert len(Vs) == count
            aggregated = isinstance(Vs[0], (list, tuple))
            m = len(Vs[0]) if aggregated else 1
            if aggregated:
                if any(len(v) != m for v in Vs) or m < 1 or self.n % m:
                    raise Exception("Proof invalid")
                Vs = [V for group in Vs for V in group]
            vbytes = b"".join([V.to_le64() for V in Vs])
        weights = seed = None
        if getattr(self, "_custom_rng", False):
            weights = b"".join(self._weight().to_bytes(32, "little") for _ in range(4 * count))
        else:
            seed = os.urandom(32)
        offs = offsets if isinstance(offsets, ctypes.Array) else (ctypes.c_uint64 * (count + 1))(*offsets)
        eng = self._eng()
        if getattr(self, "_d_shared_pts", None) is None or self._d_shared_pts.engine is not eng:
            self._d_shared_pts, self._d_shared_scs = eng.upload(self._shared_pts), eng.alloc(max(32 * (3 + 2 * self.n), 16))
        npairs = count * (m + 6 + 2 * k)
        bufs = getattr(self, "_wire_bufs", None)
        if bufs is None or bufs[0] != npairs or bufs[1].engine is not eng:
            if bufs is not None:
                bufs[1].free()
                bufs[2].free()
            bufs = (npairs, eng.alloc(64 * npairs), eng.alloc(32 * npairs))
            self._wire_bufs = bufs
        if isinstance(joined, bytes):
            src, nbytes = joined, len(joined)
        elif hasattr(joined, "ptr"):
            src, nbytes = joined.ptr, joined.nbytes
        else:
            nbytes = len(joined)
            try:
                src = ctypes.addressof((ctypes.c_char * nbytes).from_buffer(joined))
            except TypeError:
                src = bytes(joined)
        out = ctypes.create_string_buffer(64)
        bad = ctypes.c_int64(-1)
        eng._ck(eng.lib.bpmi_rp_batch_verify_dev(eng.ctx, self.n, m, count, src, nbytes, ctypes.cast(offs, ctypes.c_void_p), weights, seed, vbytes,
                                                 self._d_shared_pts.ptr, bufs[1].ptr, bufs[2].ptr, out, ctypes.cast(ctypes.pointer(bad), ctypes.c_void_p)))
        if bad.value >= 0:
            raise Exception("Proof invalid")
        return out.raw

    def verify_wire(self, Vs, blobs, offsets=None, sharded=None):
        """True if every proof of this batch of wire proofs is valid; raises Exception("Proof invalid") otherwise (partial_wire +
        the comparison with the identity; `sharded`: a distributed.ShardedMSM that folds the ranks' values first)."""
        part = self.partial_wire(Vs, blobs, offsets)
        if sharded is not None:
            part = sharded.combine(part)
        if part != _ZERO64:
            raise Exception("Proof invalid")
        return True

    def _absorb_shared(self, sh, count):
        vals = [int.from_bytes(sh[32 * i: 32 * i + 32], "little") for i in range(5 + 2 * self.n)]
        self.c_g = (self.c_g + vals[0]) % Q
        self.c_h = (self.c_h + vals[1]) % Q
        self.c_u = (self.c_u + vals[2]) % Q
        self._gs_const = (self._gs_const + vals[3]) % Q
        self._hs_const = (self._hs_const + vals[4]) % Q
        for i in range(self.n):
            self.c_gs[i] += vals[5 + i]
            self.c_hs[i] += vals[5 + self.n + i]
        self.count += count

    def partial(self):
        """The 64-byte value of this verifier's accumulated combination (one MSM)."""
        shared = [self.c_g, self.c_h, self.c_u] + [v + self._gs_const for v in self.c_gs] + \
            [v + self._hs_const for v in self.c_hs]
        pts = self._shared_pts + b"".join(self._pts) + b"".join(self._raw_pts)
        scs = b"".join(_le32(v) for v in shared) + b"".join(_le32(v) for v in self._scs) + b"".join(self._raw_scs)
        npts = 3 + 2 * self.n + len(self._scs) + self._raw_count
        if self._dev_chunks:
            return self._partial_dev(pts, scs, npts)
        msm = self._msm or self._eng().msm_bytes
        return msm(pts, scs, npts)

    def _partial_dev(self, pts, scs, npts):
        """One MSM over the host-side part (shared generators, proofs added as objects) and the device-resident chunks of
        add_wire_native: up to three segments go to bpmi_msm_segs_dev as they are; more are packed into one buffer on the device."""
        import ctypes
        eng = self._eng()
        chunks = list(self._dev_chunks)
        if len(chunks) > 2:
            total = sum(c[2] for c in chunks)
            big_p, big_s = eng.alloc(64 * total), eng.alloc(32 * total)
            pos = 0
            for d_p, d_s, cnt in chunks:
                eng._ck(eng.lib.bpmi_memcpy_dev(eng.ctx, big_p.ptr + 64 * pos, d_p.ptr, 64 * cnt))
                eng._ck(eng.lib.bpmi_memcpy_dev(eng.ctx, big_s.ptr + 32 * pos, d_s.ptr, 32 * cnt))
                pos += cnt
            eng.sync()
            for d_p, d_s, _ in chunks:
                d_p.free()
                d_s.free()
            chunks = [(big_p, big_s, total)]
            self._dev_chunks = chunks
        only_shared = npts == 3 + 2 * self.n                      # nothing but the shared generators on the host side: the usual wire batch
        if only_shared:
            # the generators never change: they are uploaded once per verifier, their coefficients go into a buffer that is kept
            if getattr(self, "_d_shared_pts", None) is None or self._d_shared_pts.engine is not eng:
                self._d_shared_pts, self._d_shared_scs = eng.upload(pts), eng.alloc(max(len(scs), 16))
            d_hp, d_hs = self._d_shared_pts, self._d_shared_scs.upload(scs)
        else:
            d_hp, d_hs = eng.upload(pts), eng.upload(scs)
        segs = [(d_hp, d_hs, npts)] + chunks
        nseg = len(segs)
        P = (ctypes.c_void_p * nseg)(*[s[0].ptr for s in segs])
        S = (ctypes.c_void_p * nseg)(*[s[1].ptr for s in segs])
        N = (ctypes.c_uint64 * nseg)(*[s[2] for s in segs])
        out = ctypes.create_string_buffer(64)
        try:
            eng._ck(eng.lib.bpmi_msm_segs_dev(eng.ctx, nseg, P, S, N, out))
        finally:
            if not only_shared:
                d_hp.free()
                d_hs.free()
        return out.raw

    def verify(self, sharded=None):
        """True if every added proof is valid; raises Exception("Proof invalid") otherwise.
        `sharded`: a distributed.ShardedMSM whose combine() folds the ranks' partials."""
        part = self.partial()
        if sharded is not None:
            part = sharded.combine(part)
        if part != _ZERO64:
            raise Exception("Proof invalid")
        return True


def batch_verify(Vs, proofs, g, h, gs, hs, u, **kw):
    bv = BatchRangeVerifier(g, h, gs, hs, u, **kw)
    for V, pr in zip(Vs, proofs):
        bv.add(V, pr)
    return bv.verify()


# ---- worker side of add_wire (module level: picklable by name) -------------------------------
_worker_gens = None


def _worker_init(gens):
    global _worker_gens
    g, h, u, gs, hs = gens
    n = len(gs) // 64
    _worker_gens = (Point.from_le64(g), Point.from_le64(h), [Point.from_le64(gs[64 * i: 64 * i + 64]) for i in range(n)],
                    [Point.from_le64(hs[64 * i: 64 * i + 64]) for i in range(n)], Point.from_le64(u))


def _worker_ping(i):
    import time
    time.sleep(0.25)        # long enough that the executor has to start one process per ping
    return i


def _worker_add(blobs, pts, Vs):
    from .codec import assemble, parse_blob
    g, h, gs, hs, u = _worker_gens
    bv = BatchRangeVerifier(g, h, gs, hs, u)
    pos = 0
    for j, blob in enumerate(blobs):
        parsed = parse_blob(blob)
        bv.add(Point.from_le64(Vs[64 * j: 64 * j + 64]), assemble(parsed, pts, pos))
        pos += 6 + 2 * parsed[0]
    return bv.state()
