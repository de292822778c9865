"""Bulletproofs range proofs behind the reference's class names (src/rangeproofs), plus what the
reference does not have: a batch verifier and a wire format.

    NIRangeProver(v, n, g, h, gs, hs, gamma, u, group, seed).prove() -> Proof
    RangeVerifier(V, g, h, gs, hs, u, proof).verify()                -> True | Exception("Proof invalid")
    AggregNIRangeProver / AggregRangeVerifier                         m values in one proof
    BatchRangeVerifier, batch_verify                                  many proofs, one MSM
    BatchRangeProver                                                  many proofs, one device call
    proof_to_bytes, proofs_from_bytes                                 canonical bytes, GPU decompression
"""
from .batch import BatchRangeVerifier, batch_verify
from .batch_prover import BatchRangeProver
from .codec import proof_to_bytes, proofs_from_bytes
from .common import Proof
from .rangeproof_aggreg_prover import AggregNIRangeProver
from .rangeproof_aggreg_verifier import AggregRangeVerifier
from .rangeproof_prover import NIRangeProver
from .rangeproof_verifier import RangeVerifier

__all__ = [
    "AggregNIRangeProver", "AggregRangeVerifier", "BatchRangeProver", "BatchRangeVerifier", "NIRangeProver", "Proof", "RangeVerifier",
    "batch_verify", "proof_to_bytes", "proofs_from_bytes",
]
