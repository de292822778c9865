from .rangeproof_prover import NIRangeProver
from .rangeproof_verifier import RangeVerifier
from .rangeproof_aggreg_prover import AggregNIRangeProver
from .rangeproof_aggreg_verifier import AggregRangeVerifier
from .common import Proof
from .batch import BatchRangeVerifier, batch_verify
from .codec import proof_to_bytes, proofs_from_bytes

__all__ = ["NIRangeProver", "RangeVerifier", "AggregNIRangeProver", "AggregRangeVerifier", "Proof",
           "BatchRangeVerifier", "batch_verify", "proof_to_bytes", "proofs_from_bytes"]
