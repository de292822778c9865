"""Many single-value range proofs over the same generators, proved in ONE device call (SURVEY.md section 2.1 K11;
csrc/rp_prove_kernels.hpp).  The reference proves one at a time -- a loop of
NIRangeProver(v, n, g, h, gs, hs, gamma, u, group, seed).prove() (/root/reference/src/rangeproofs/rangeproof_prover.py:35-91 with
/root/reference/src/innerproduct/inner_product_prover.py:27-44, :84-110 inside) -- and so did this package (2.1 ms a proof, every
step a launch).  Here every protocol step is one launch over the whole batch, the generators' multiples come from tables built once
per prover, and the transcripts are hashed on the device.  Same proofs, byte for byte:

    bp = BatchRangeProver(n, g, h, gs, hs, u)
    blobs = bp.prove_wire(vs, gammas, seeds)        # wire format 2 (rangeproofs/codec.py), one bytes object per proof
    proofs = bp.prove(vs, gammas, seeds)            # the same as Proof objects
    # blobs[i] == proof_to_bytes(NIRangeProver(vs[i], n, g, h, gs, hs, gammas[i], u, group, seeds[i]).prove(), version=2)

Round 6: AGGREGATED proofs too -- a loop of AggregNIRangeProver(vs_i, n, g, h, gs, hs, gammas_i, u, group, seed_i).prove()
(/root/reference/src/rangeproofs/rangeproof_aggreg_prover.py:36-146) over proofs of m values each, n m <= 128:

    bp = BatchRangeProver(n, g, h, gs, hs, u, m=4)        # gs, hs: n m points
    blobs = bp.prove_wire(vss, gammass, seeds)            # vss[i], gammass[i]: the m values / blinding factors of proof i
"""
import ctypes

from .. import engine as _engine
from ..ec import secp256k1
from .codec import proofs_from_bytes

Q = secp256k1.q


def _le32(v):
    return (int(v.x if hasattr(v, "x") else v) % Q).to_bytes(32, "little")


class BatchRangeProver:
    def __init__(self, n, g, h, gs, hs, u, engine=None, m=1, wire_format=2):
        """n: bits per value, m: values per proof (powers of two, 2 <= n m <= 128; m = 1: single-value proofs); g, h, u: points; gs, hs:
        n m points each.  Builds the fixed-base tables on the engine's device (4.4 GB and ~72 ms for n m = 64 with the default 16-bit
        windows; engine option prover_table_bits: 12 bits are 378 MB, 17 ms and 22 % slower proving) and keeps them until close().
        wire_format: 2, or 3 -- the proofs then end with their points' y coordinates (rangeproofs/codec.py), which the batch verifier
        checks instead of taking square roots; the prover has them anyway."""
        if len(gs) != n * m or len(hs) != n * m:
            raise ValueError("gs and hs must have n m points each")
        if wire_format not in (2, 3):
            raise ValueError("wire_format must be 2 or 3")
        self.n, self.m, self.wire_format = n, m, wire_format
        self._engine = engine or _engine.default_engine()
        eng = self._engine
        handle = ctypes.c_void_p()
        eng._ck(eng.lib.bpmi_rp_prover_create_aggregated(eng.ctx, n, m, g.to_le64(), h.to_le64(), u.to_le64(), b"".join(p.to_le64() for p in gs),
                                                         b"".join(p.to_le64() for p in hs), ctypes.byref(handle)))
        self._handle = handle.value
        self._out = None

    def prove_wire_packed(self, vs, gammas, seeds, copy=True):
        """(packed bytes, offsets): proof i = packed[offsets[i]: offsets[i + 1]], wire format 2 (or 3) -- what
        BatchRangeVerifier.add_wire_native / bpmi_rp_batch_verify_dev take as they are.
        vs, gammas: lists of ModP / int (aggregated provers: a list of m values per proof), or ALREADY PACKED bytes (32 bytes
        little-endian per value, reduced mod q, proof after proof): a service that receives its inputs as bytes skips 2 x len Python
        conversions.  seeds: a list of bytes, or (joined bytes, offsets) -- offsets a list or a ctypes c_uint64 array (taken as it is).
        copy=False: `packed` is a memoryview of the prover's page-locked output buffer and `offsets` the ctypes array the library
        filled -- valid until the next call on this prover or its close(); a service that forwards the bytes (a socket, the batch verifier's
        receive buffer) saves the one host copy of the batch (18 MB at 2^14 proofs) and the list of 2^14 Python integers."""
        vm = self.m

        def flat(xs):
            if vm == 1 and not (len(xs) and isinstance(xs[0], (list, tuple))):
                return xs
            for row in xs:
                if len(row) != vm:
                    raise ValueError("every proof of this prover takes %d values" % vm)
            return [x for row in xs for x in row]
        if isinstance(vs, (bytes, bytearray, memoryview)):
            vb, m = bytes(vs), len(vs) // (32 * vm)
        else:
            m = len(vs)
            vb = b"".join([_le32(v) for v in flat(vs)])
        gb = bytes(gammas) if isinstance(gammas, (bytes, bytearray, memoryview)) else b"".join([_le32(x) for x in flat(gammas)])
        if isinstance(seeds, tuple):
            sb, offs = seeds
            if len(offs) != m + 1:
                raise ValueError("values, blinding factors and seeds must have the same length")
            off = offs if isinstance(offs, ctypes.Array) else (ctypes.c_uint64 * (m + 1))(*offs)
        else:
            if len(seeds) != m:
                raise ValueError("values, blinding factors and seeds must have the same length")
            off = (ctypes.c_uint64 * (m + 1))()
            pos = 0
            for i, sd in enumerate(seeds):
                off[i] = pos
                pos += len(sd)
            off[m] = pos
            sb = b"".join(seeds)
        if len(vb) != 32 * m * vm or len(gb) != 32 * m * vm:
            raise ValueError("values, blinding factors and seeds must have the same length")
        eng = self._engine
        eng.set_option("prover_wire_format", self.wire_format)           # (an engine option: set per call, provers may share the engine)
        cap = m * eng.lib.bpmi_rp_prove_batch_proof_bytes(self._handle, 0) + (off[m] - off[0]) + 16      # a proof is a fixed part + its seed
        # the proofs land in a page-locked buffer of the prover (kept between batches): the library copies them there straight from the
        # device, and ONE host copy makes the bytes object (a fresh 18 MB ctypes buffer per batch cost three: zero-fill, staging copy, string_at)
        if self._out is None or len(self._out) < cap:
            if self._out is not None:
                self._out.free()
            self._out = eng.host_alloc(cap + cap // 8)
        out_off = (ctypes.c_uint64 * (m + 1))()
        eng._ck(eng.lib.bpmi_rp_prove_batch(self._handle, m, vb, gb, sb, off, ctypes.c_void_p(self._out.ptr), cap, out_off))
        if not copy:
            return self._out.view[: out_off[m]], out_off
        return bytes(self._out.view[: out_off[m]]), list(out_off)

    def prove_wire(self, vs, gammas, seeds):
        packed, off = self.prove_wire_packed(vs, gammas, seeds)
        return [packed[off[i]: off[i + 1]] for i in range(len(off) - 1)]          # (vs may be packed bytes: 32 per value)

    def prove(self, vs, gammas, seeds):
        """The proofs as Proof objects (every point decompressed in one more launch)."""
        return proofs_from_bytes(self.prove_wire(vs, gammas, seeds), engine=self._engine)

    def last_ms(self):
        """Device milliseconds of the last batch by phase (bpmi_rp_prover_last_ms)."""
        ms = (ctypes.c_double * 7)()
        self._engine._ck(self._engine.lib.bpmi_rp_prover_last_ms(self._handle, ms))
        names = ("A_S", "yz_T1_T2", "x_vectors_Pnew", "ipa_rounds", "wire_bytes", "copy_out", "total")
        return dict(zip(names, ms))

    def close(self):
        if getattr(self, "_handle", None):
            self._engine.lib.bpmi_rp_prover_destroy(self._handle)
            self._handle = None
        if getattr(self, "_out", None) is not None:
            self._out.free()
            self._out = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
