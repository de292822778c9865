"""Point <-> bytes in the reference's encoding (src/utils/utils.py:100-131): SEC1 compressed form
(0x02 / 0x03 by the parity of y, then x big-endian), the identity as the single byte 0x00, and
the base64 of that for transcripts.  Bulk decoding of many points is bpmi_ec_decompress_batch
(rangeproofs/codec.py); these functions handle one point on the host."""
from base64 import b64decode, b64encode

from ..ec import Point, mod_sqrt, secp256k1

CURVE = secp256k1
BYTE_LENGTH = CURVE.q.bit_length() // 8


def point_to_bytes(g: Point) -> bytes:
    if g == Point.IDENTITY_ELEMENT:
        return b"\x00"
    tag = b"\x03" if g.y & 1 else b"\x02"
    return tag + g.x.to_bytes(BYTE_LENGTH, "big")


def point_to_b64(g: Point) -> bytes:
    return b64encode(point_to_bytes(g))


def bytes_to_point(b: bytes) -> Point:
    if b == 0:                      # never true for bytes; kept because the reference has it (utils.py:121)
        return Point.IDENTITY_ELEMENT
    want_odd = b[0] != 2
    x = int.from_bytes(b[1:], "big")
    y = mod_sqrt((x * x * x + CURVE.a * x + CURVE.b) % CURVE.p, CURVE.p)[0]
    if bool(y & 1) != want_odd:
        y = CURVE.p - y
    return Point(x, y, CURVE)


def b64_to_point(s: bytes) -> Point:
    return bytes_to_point(b64decode(s))
