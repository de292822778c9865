"""Inner-product argument: proof containers and verifiers
(reference: src/innerproduct/inner_product_verifier.py)."""
from ..ec import Point, secp256k1
from ..pippenger import PipSECP256k1
from ..utils.utils import ModP, mod_hash, point_to_b64

SUPERCURVE = secp256k1


class Proof1:
    """Protocol 1 proof (reference :10-17)."""

    def __init__(self, u_new, P_new, proof2, transcript):
        self.u_new, self.P_new, self.proof2, self.transcript = u_new, P_new, proof2, transcript


class Proof2:
    """Protocol 2 proof (reference :61-73)."""

    def __init__(self, a, b, xs, Ls, Rs, transcript, start_transcript: int = 0):
        self.a, self.b, self.xs, self.Ls, self.Rs = a, b, xs, Ls, Rs
        self.transcript = transcript
        self.start_transcript = start_transcript


class _Checker:
    def assertThat(self, expr):
        if not expr:
            raise Exception("Proof invalid")


class Verifier2(_Checker):
    """Protocol 2 verifier (reference :76-147): transcript re-derivation, the s-vector,
    one MSM of size 2n+1 and one of size 2 log n -- both on the GPU."""

    def __init__(self, g, h, u, P, proof: Proof2):
        self.g, self.h, self.u, self.P, self.proof = g, h, u, P, proof

    def get_ss(self, xs):
        """s_i = prod_j xs[j]^(+1 if bit j (MSB first) of i is set else -1), i = 0..n-1
        (reference :91-102), built by doubling: n multiplications and log n inversions
        instead of n log n of each."""
        n = len(self.g)
        ss = [ModP(1, SUPERCURVE.q)]
        for x in reversed(xs[: n.bit_length() - 1]):
            xi = x.inv()
            ss = [s * xi for s in ss] + [s * x for s in ss]
        return ss

    def _scaled_ss(self, xs, a, b):
        """([a * s_i], [b * s_i^-1]) as plain integers mod q, by the same doubling."""
        q = SUPERCURVE.q
        n = len(self.g)
        sa, sb = [a % q], [b % q]
        for x in reversed(xs[: n.bit_length() - 1]):
            xv = x.x % q
            xi = pow(xv, -1, q)
            sa = [s * xi % q for s in sa] + [s * xv % q for s in sa]
            sb = [s * xv % q for s in sb] + [s * xi % q for s in sb]
        return sa, sb

    def verify_transcript(self):
        pr = self.proof
        log_n = len(self.g).bit_length() - 1
        items = pr.transcript.split(b"&")
        k = pr.start_transcript
        for i in range(log_n):
            self.assertThat(items[k + 3 * i] == point_to_b64(pr.Ls[i]))
            self.assertThat(items[k + 3 * i + 1] == point_to_b64(pr.Rs[i]))
            rehash = str(mod_hash(b"&".join(items[: k + 3 * i + 2]) + b"&", SUPERCURVE.q)).encode()
            self.assertThat(str(pr.xs[i]).encode() == items[k + 3 * i + 2] == rehash)

    def verify(self):
        self.verify_transcript()
        pr = self.proof
        sa, sb = self._scaled_ss(pr.xs, pr.a.x, pr.b.x)
        lhs = PipSECP256k1.multiexp(self.g + self.h + [self.u], sa + sb + [pr.a * pr.b])
        # RHS = P + sum x_i^2 L_i + x_i^-2 R_i  as one MSM with P at scalar 1
        rhs = PipSECP256k1.multiexp(
            pr.Ls + pr.Rs + [self.P],
            [x ** 2 for x in pr.xs] + [x.inv() ** 2 for x in pr.xs] + [1],
        )
        self.assertThat(lhs == rhs)
        return True


class Verifier1(_Checker):
    """Protocol 1 verifier (reference :20-58)."""

    def __init__(self, g, h, u, P, c, proof1):
        self.g, self.h, self.u, self.P, self.c, self.proof1 = g, h, u, P, c, proof1

    def verify_transcript(self):
        items = self.proof1.transcript.split(b"&")
        self.assertThat(items[1] == str(mod_hash(b"&".join(items[:1]) + b"&", SUPERCURVE.q)).encode())

    def verify(self):
        self.verify_transcript()
        x = ModP(int(self.proof1.transcript.split(b"&")[1]), SUPERCURVE.q)
        self.assertThat(self.proof1.P_new == self.P + (x * self.c) * self.u)
        self.assertThat(self.proof1.u_new == x * self.u)
        return Verifier2(self.g, self.h, self.proof1.u_new, self.proof1.P_new, self.proof1.proof2).verify()
