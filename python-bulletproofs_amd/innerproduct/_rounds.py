"""The halving loop of the inner-product argument, shared by the one-GPU prover and the sharded
one.  Everything heavy stays on the device inside `state` (engine.IpaState); per round 2 x 64
bytes come up (L, R), the Fiat-Shamir challenge is hashed here -- byte-exact with the reference
(src/innerproduct/inner_product_prover.py:96-110) -- and 2 x 32 bytes (x, 1/x) go down."""
from ..ec import Point, secp256k1
from ..utils.scalar import ModP
from ..utils.transcript import Transcript


def run_rounds(state, transcript, q, xs, Ls, Rs, combine=None):
    """Runs rounds until the state has length 1.  `combine(L_bytes, R_bytes) -> (L_bytes, R_bytes)`
    turns this rank's partial L, R into the global ones (sharded prover); None on one GPU."""
    if combine is None and type(transcript) is Transcript and hasattr(state, "prove_rounds") and q == secp256k1.q:
        # one GPU: the whole loop, hashing included, in one native call (bpmi_ipa_prove_rounds) -- the same bytes as below
        transcript.digest, x_ints, Lbs, Rbs = state.prove_rounds(transcript.digest)
        Ls.extend(Point.from_le64(b) for b in Lbs)
        Rs.extend(Point.from_le64(b) for b in Rbs)
        xs.extend(ModP(x, q) for x in x_ints)
        return
    while len(state) > 1:
        Lb, Rb = state.round_LR()                           # reference :96-99
        if combine is not None:
            Lb, Rb = combine(Lb, Rb)
        L, R = Point.from_le64(Lb), Point.from_le64(Rb)
        Ls.append(L)
        Rs.append(R)
        transcript.add_list_points([L, R])                  # :102
        x = transcript.get_modp(q)                          # :104
        xs.append(x)
        transcript.add_number(x)                            # :106
        state.fold(x.x, x.inv().x)                          # :107-110
