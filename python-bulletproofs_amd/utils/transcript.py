"""Fiat-Shamir transcript, byte-exact with the reference (src/utils/transcript.py:6-33).

The transcript is ONE growing byte string: `base64(seed)&`, then for every absorbed item either
`base64(SEC1-compressed point)&` or `decimal digits&`.  A challenge is mod_hash of the whole
string so far.  It lives on the host on purpose: it is the serial edge between the rounds of the
inner-product argument (64 bytes per round come up from the GPU, 64 go down), and every proof's
`transcript` field must reproduce the reference's bytes for the verifiers' consistency checks."""
from base64 import b64encode

from .utils import mod_hash, point_to_b64

_SEP = b"&"


class Transcript:
    __slots__ = ("digest",)

    def __init__(self, seed=b""):
        self.digest = b64encode(seed) + _SEP

    def _absorb(self, chunk):
        self.digest += chunk + _SEP

    def add_point(self, g):
        self._absorb(point_to_b64(g))

    def add_list_points(self, gs):
        self.digest += b"".join(point_to_b64(g) + _SEP for g in gs)

    def add_number(self, x):
        self._absorb(str(x).encode())

    def get_modp(self, p):
        """The challenge for the current transcript, as a ModP in [1, p)."""
        return mod_hash(self.digest, p)
