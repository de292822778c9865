"""A wire format for range proofs.  The reference never
serialises a proof -- only point_to_bytes / bytes_to_point exist (src/utils/utils.py:100-131);
this module fixes a canonical byte layout on top of those encodings and decodes MANY proofs at
once, decompressing all their points in one GPU launch (bpmi_ec_decompress_batch).

Layout (integers big-endian):
  "BPRP1" | k (1 B, IPA rounds)
  taux | mu | t_hat | a | b            5 x 32 B
  xs[0..k)                             k x 32 B
  T1 T2 A S u_new P_new Ls[0..k) Rs[0..k)   (6 + 2k) x 33 B, SEC1 compressed; identity = 33 zero bytes
  start_transcript (2 B)
  len (4 B) | range-proof transcript ; len | Protocol-1 transcript ; len | Protocol-2 transcript
A serialised proof is at most 32 KiB (a 64-bit proof is 2.6 KB): the native batch preparation, host and device, calls a longer
one invalid.
"""
import struct

from .. import engine as _engine
from ..ec import Point, secp256k1
from ..innerproduct.inner_product_verifier import Proof1, Proof2
from ..utils.utils import ModP, point_to_bytes
from .common import Proof

MAGIC = b"BPRP1"
Q = secp256k1.q


def _pt33(P):
    b = point_to_bytes(P)
    return b if len(b) == 33 else bytes(33)


def proof_to_bytes(proof) -> bytes:
    ip, p2 = proof.innerProof, proof.innerProof.proof2
    k = len(p2.xs)
    assert len(p2.Ls) == k and len(p2.Rs) == k and k < 256
    out = [MAGIC, bytes([k])]
    out += [(v.x % Q).to_bytes(32, "big") for v in (proof.taux, proof.mu, proof.t_hat, p2.a, p2.b)]
    out += [(x.x % Q).to_bytes(32, "big") for x in p2.xs]
    out += [_pt33(P) for P in [proof.T1, proof.T2, proof.A, proof.S, ip.u_new, ip.P_new] + list(p2.Ls) + list(p2.Rs)]
    out.append(struct.pack(">H", p2.start_transcript))
    for t in (proof.transcript, ip.transcript, p2.transcript):
        out += [struct.pack(">I", len(t)), t]
    return b"".join(out)


def parse_blob(blob):
    """Structural checks and the scalar part of one serialised proof ->
    (k, ints, compressed_points, start_transcript, [3 transcripts]); no point is decoded."""
    if len(blob) < 6 or blob[:5] != MAGIC:
        raise Exception("Proof invalid")
    k = blob[5]
    o = 6
    need = o + 32 * (5 + k) + 33 * (6 + 2 * k) + 2
    if len(blob) < need:
        raise Exception("Proof invalid")
    ints = [int.from_bytes(blob[o + 32 * j: o + 32 * j + 32], "big") for j in range(5 + k)]
    if any(v >= Q for v in ints):
        raise Exception("Proof invalid")
    o += 32 * (5 + k)
    npts = 6 + 2 * k
    comp = blob[o: o + 33 * npts]
    o += 33 * npts
    (start,) = struct.unpack(">H", blob[o: o + 2])
    o += 2
    ts = []
    for _ in range(3):
        if len(blob) < o + 4:
            raise Exception("Proof invalid")
        (ln,) = struct.unpack(">I", blob[o: o + 4])
        o += 4
        if len(blob) < o + ln:
            raise Exception("Proof invalid")
        ts.append(blob[o: o + ln])
        o += ln
    if o != len(blob):
        raise Exception("Proof invalid")
    return k, ints, comp, start, ts


def compressed_points(blob):
    """The 33-byte point encodings of a serialised proof (cheap: no integer is parsed)."""
    if len(blob) < 6 or blob[:5] != MAGIC:
        raise Exception("Proof invalid")
    k = blob[5]
    o = 6 + 32 * (5 + k)
    end = o + 33 * (6 + 2 * k)
    if len(blob) < end:
        raise Exception("Proof invalid")
    return blob[o:end]


def assemble(parsed, pts, pos=0):
    """parse_blob output + the decompressed points (64-byte wire form, this proof's first
    one at index `pos` of `pts`) -> Proof."""
    k, ints, comp, start, ts = parsed
    npts = 6 + 2 * k
    P = [Point.from_le64(pts[64 * (pos + j): 64 * (pos + j) + 64]) for j in range(npts)]
    sc = [ModP(v, Q) for v in ints]
    p2 = Proof2(sc[3], sc[4], sc[5:], P[6: 6 + k], P[6 + k: 6 + 2 * k], ts[2], start)
    p1 = Proof1(P[4], P[5], p2, ts[1])
    return Proof(sc[0], sc[1], sc[2], P[0], P[1], P[2], P[3], p1, ts[0])


def proofs_from_bytes(blobs, engine=None):
    """Decode a list of serialised proofs; every point of every proof is decompressed in ONE
    GPU launch.  Raises Exception("Proof invalid") on a malformed blob or an invalid point."""
    parsed = [parse_blob(blob) for blob in blobs]
    total = sum(6 + 2 * p[0] for p in parsed)
    eng = engine or _engine.default_engine()
    pts, ok = eng.ec_decompress_batch_bytes(b"".join(p[2] for p in parsed), total)
    if any(flag == 0 for flag in ok):
        raise Exception("Proof invalid")
    out, pos = [], 0
    for p in parsed:
        out.append(assemble(p, pts, pos))
        pos += 6 + 2 * p[0]
    return out
