"""Fiat-Shamir transcript (reference: src/utils/transcript.py:6-33): a growing byte
string `b64(seed)&` + `b64(compressed point)&` / `decimal&` items; challenges are
mod_hash of the whole string.  Host-side and byte-exact: it is the serial edge between
IPA rounds."""
import base64

from .utils import mod_hash, point_to_b64


class Transcript:
    def __init__(self, seed=b""):
        self.digest = base64.b64encode(seed) + b"&"

    def add_point(self, g):
        self.digest += point_to_b64(g) + b"&"

    def add_list_points(self, gs):
        for g in gs:
            self.add_point(g)

    def add_number(self, x):
        self.digest += str(x).encode() + b"&"

    def get_modp(self, p):
        return mod_hash(self.digest, p)
