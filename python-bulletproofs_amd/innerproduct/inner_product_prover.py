"""Inner-product argument provers (reference: src/innerproduct/inner_product_prover.py).

FastNIProver2 keeps g, h, a, b resident on the GPU for the whole log-n halving
(bpmi_ipa_*); per round only L and R (64 bytes each) come back for the host-side
Fiat-Shamir hash, and x, x^-1 (32 bytes each) go down for the fold."""
from typing import Optional

from .. import engine as _engine
from ..ec import pack_points, pack_scalars
from ..pippenger import PipSECP256k1
from ..utils.transcript import Transcript
from ..utils.utils import ModP
from ._rounds import run_rounds
from .inner_product_verifier import Proof1, Proof2


class NIProver:
    """Protocol 1 -> Protocol 2 reduction (reference :11-45)."""

    def __init__(self, g, h, u, P, c, a, b, group, seed=b"", h_scale=None):
        """h_scale (extension, not in the reference): integers c_i; the argument then runs
        over the generators c_i * h[i] without materialising them (see Verifier2)."""
        assert len(g) == len(h) == len(a) == len(b)
        self.g, self.h, self.u, self.P, self.c, self.a, self.b = g, h, u, P, c, a, b
        self.group = group
        self.h_scale = h_scale
        self.transcript = Transcript(seed)

    def prove(self) -> Proof1:
        x = self.transcript.get_modp(self.group.q)
        self.transcript.add_number(x)
        # P' = P + (x c) u and u' = x u, independent of each other: one round trip
        P_new, u_new = PipSECP256k1.multiexp2([self.P, self.u], [1, x * self.c], [self.u], [x])
        inner = FastNIProver2(self.g, self.h, u_new, P_new, self.a, self.b, self.group, self.transcript.digest,
                              h_scale=self.h_scale)
        return Proof1(u_new, P_new, inner.prove(), self.transcript.digest)


class FastNIProver2:
    """Protocol 2 prover (reference :48-110)."""

    def __init__(self, g, h, u, P, a, b, group, transcript: Optional[bytes] = None, h_scale=None):
        assert len(g) == len(h) == len(a) == len(b)
        self.h_scale = h_scale
        assert len(a) & (len(a) - 1) == 0
        self.n = len(a)
        self.log_n = self.n.bit_length() - 1
        self.g, self.h, self.u, self.P, self.a, self.b, self.group = g, h, u, P, a, b, group
        self.transcript = Transcript()
        if transcript:
            self.transcript.digest += transcript
            self.init_transcript_length = len(transcript.split(b"&"))
        else:
            self.init_transcript_length = 1

    def prove(self) -> Proof2:
        q = self.group.q
        eng = _engine.default_engine()
        state = eng.ipa_create(pack_points(self.g), pack_points(self.h), pack_scalars(self.a, q),
                               pack_scalars(self.b, q), self.n, self.u.to_le64(),
                               None if self.h_scale is None else pack_scalars(self.h_scale, q))
        xs, Ls, Rs = [], [], []
        try:
            run_rounds(state, self.transcript, q, xs, Ls, Rs)
            a, b = state.finish()
        finally:
            state.close()
        return Proof2(ModP(a, q), ModP(b, q), xs, Ls, Rs, self.transcript.digest, self.init_transcript_length)
