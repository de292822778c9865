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

Format 2 (round 4; csrc/rp_wire_v2_host.hpp) is format 1 without the three transcripts -- they spell out, in base64 and decimal,
what the other fields already say (src/utils/transcript.py:13-33) -- plus what they held that is NOT elsewhere:
  "BPRP2" | k | the same scalar and point sections | y z x x_ip (4 x 32 B) | len (2 B) seed | len (2 B) Protocol-1 seed
1.09 KB for a 64-bit proof.  `wire_v2_to_v1` rebuilds the format-1 bytes; a format-2 proof is valid exactly when they are, and
every parser here takes either format.  Only proofs whose transcripts ARE the canonical ones can be written as format 2
(`proof_to_bytes(proof, version=2)` checks it and raises otherwise).

Format 3 (round 6) is a format-2 proof with the magic "BPRP3", followed by the y coordinate of each of its 6 + 2k points (32 B
big-endian, 0 for the identity): 1.67 KB for a 64-bit proof.  A verifier checks each y -- below p, the parity the encoding's tag
asks for, on the curve with x -- instead of computing it: decoding a point (src/utils/utils.py:119-131) is a square root, 6 + 2k of
them per proof, a quarter of a batch verification's device time.  A format-3 proof is valid exactly when its format-2 part is and
every y is right.  `wire_v3_to_v2` checks and strips the ys; `wire_v2_to_v3` adds them (one GPU launch for a list of proofs).
"""
import struct

from .. import engine as _engine
from ..ec import Point, secp256k1
from ..innerproduct.inner_product_verifier import Proof1, Proof2
from ..utils.utils import ModP, point_to_bytes
from .common import Proof

MAGIC = b"BPRP1"
MAGIC2 = b"BPRP2"
MAGIC3 = b"BPRP3"
Q = secp256k1.q
_P = secp256k1.p


def _pt33(P):
    b = point_to_bytes(P)
    return b if len(b) == 33 else bytes(33)


def _body_len(k):
    return 6 + 32 * (5 + k) + 33 * (6 + 2 * k)


def wire_v3_to_v2(blob) -> bytes:
    """The format-2 proof inside a format-3 one, after checking that every y IS its point's y coordinate."""
    if len(blob) < 6 or blob[:5] != MAGIC3 or blob[5] > 16:
        raise Exception("Proof invalid")
    k = blob[5]
    npts, at = 6 + 2 * k, 6 + 32 * (5 + k)
    if len(blob) < _body_len(k) + 132 + 32 * npts:
        raise Exception("Proof invalid")
    ys = len(blob) - 32 * npts
    for j in range(npts):
        c, y = blob[at + 33 * j: at + 33 * j + 33], int.from_bytes(blob[ys + 32 * j: ys + 32 * j + 32], "big")
        x = int.from_bytes(c[1:], "big")
        if c[0] == 0:
            good = x == 0 and y == 0
        else:
            good = c[0] in (2, 3) and x < _P and y < _P and (y & 1) == (c[0] & 1) and (y * y - x * x * x - 7) % _P == 0
        if not good:
            raise Exception("Proof invalid")
    return MAGIC2 + bytes(blob[5:ys])


def wire_v2_to_v3(blobs, engine=None):
    """Format-2 proofs (a list) -> the same proofs in format 3; the y coordinates come from ONE batched decompression on the GPU."""
    comp = [compressed_points(b) for b in blobs]
    if any(b[:5] != MAGIC2 for b in blobs):
        raise Exception("Proof invalid")
    eng = engine or _engine.default_engine()
    total = sum(len(c) // 33 for c in comp)
    pts, ok = eng.ec_decompress_batch_bytes(b"".join(comp), total)
    if any(flag == 0 for flag in ok):
        raise Exception("Proof invalid")
    out, pos = [], 0
    for b, c in zip(blobs, comp):
        n = len(c) // 33
        # wire points are x | y, 32 bytes little-endian each (ec.Point.to_le64); the identity is 64 zero bytes
        ys = b"".join(pts[64 * (pos + j) + 32: 64 * (pos + j) + 64][::-1] for j in range(n))
        out.append(MAGIC3 + bytes(b[5:]) + ys)
        pos += n
    return out


def wire_v2_to_v1(blob) -> bytes:
    """The format-1 proof a format-2 (or format-3) proof stands for (pure Python; the bulk paths use bpmi_rp_wire_v2_to_v1 or the device)."""
    from base64 import b64encode
    if blob[:5] == MAGIC3:
        blob = wire_v3_to_v2(blob)
    if len(blob) < 6 or blob[:5] != MAGIC2:
        raise Exception("Proof invalid")
    k = blob[5]
    body = _body_len(k)
    o = body + 128
    if k > 16 or len(blob) < o + 4:
        raise Exception("Proof invalid")
    y, z, x, x_ip = (int.from_bytes(blob[body + 32 * j: body + 32 * j + 32], "big") for j in range(4))
    xs = [int.from_bytes(blob[6 + 32 * (5 + j): 6 + 32 * (6 + j)], "big") for j in range(k)]
    if any(v >= Q for v in (y, z, x, x_ip, *xs)):
        raise Exception("Proof invalid")
    seeds = []
    for _ in range(2):
        if len(blob) < o + 2:
            raise Exception("Proof invalid")
        (ln,) = struct.unpack(">H", blob[o: o + 2])
        if len(blob) < o + 2 + ln:
            raise Exception("Proof invalid")
        seeds.append(bytes(blob[o + 2: o + 2 + ln]))
        o += 2 + ln
    if o != len(blob):
        raise Exception("Proof invalid")
    pts = blob[6 + 32 * (5 + k): body]
    pt = [pts[33 * j: 33 * j + 33] for j in range(6 + 2 * k)]

    def item(c):
        return b64encode(b"\x00" if c == bytes(33) else c) + b"&"

    def num(v):
        return str(v).encode() + b"&"

    T1, T2, A, S = pt[0], pt[1], pt[2], pt[3]
    t_rp = b64encode(seeds[0]) + b"&" + item(A) + item(S) + num(y) + num(z) + item(T1) + item(T2) + num(x)
    t_1 = b64encode(seeds[1]) + b"&" + num(x_ip)
    t_2 = b"&" + t_1 + b"".join(item(pt[6 + j]) + item(pt[6 + k + j]) + num(xs[j]) for j in range(k))
    out = [MAGIC, blob[5:body], struct.pack(">H", 3)]
    for t in (t_rp, t_1, t_2):
        out += [struct.pack(">I", len(t)), t]
    return b"".join(out)


def proof_to_bytes(proof, version=1) -> bytes:
    if version == 2:
        return _proof_to_bytes_v2(proof)
    if version == 3:
        ip, p2 = proof.innerProof, proof.innerProof.proof2
        v2 = _proof_to_bytes_v2(proof)
        pts = [proof.T1, proof.T2, proof.A, proof.S, ip.u_new, ip.P_new] + list(p2.Ls) + list(p2.Rs)
        return MAGIC3 + v2[5:] + b"".join(bytes(32) if _pt33(P) == bytes(33) else int(P.y).to_bytes(32, "big") for P in pts)
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


def _proof_to_bytes_v2(proof) -> bytes:
    from base64 import b64decode
    v1 = proof_to_bytes(proof, 1)
    k = v1[5]
    body = _body_len(k)
    try:
        it_rp = proof.transcript.split(b"&")
        it_1 = proof.innerProof.transcript.split(b"&")
        y, z, x, x_ip = int(it_rp[3]), int(it_rp[4]), int(it_rp[7]), int(it_1[1])
        seed, seed1 = b64decode(it_rp[0], validate=True), b64decode(it_1[0], validate=True)
    except Exception:
        raise ValueError("the proof's transcripts are not canonical: it has no format-2 form")
    if max(y, z, x, x_ip) >= Q or min(y, z, x, x_ip) < 0 or max(len(seed), len(seed1)) > 0xFFFF:
        raise ValueError("the proof's transcripts are not canonical: it has no format-2 form")
    v2 = b"".join([MAGIC2, v1[5:body]] + [v.to_bytes(32, "big") for v in (y, z, x, x_ip)] +
                  [struct.pack(">H", len(seed)), seed, struct.pack(">H", len(seed1)), seed1])
    if wire_v2_to_v1(v2) != v1:
        raise ValueError("the proof's transcripts are not canonical: it has no format-2 form")
    return v2


def parse_blob(blob):
    """Structural checks and the scalar part of one serialised proof (either format) ->
    (k, ints, compressed_points, start_transcript, [3 transcripts]); no point is decoded."""
    if blob[:5] in (MAGIC2, MAGIC3):
        blob = wire_v2_to_v1(blob)
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
    """The 33-byte point encodings of a serialised proof of any format (cheap: no integer is parsed)."""
    if len(blob) < 6 or blob[:5] not in (MAGIC, MAGIC2, MAGIC3):
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
