from .inner_product_prover import NIProver, FastNIProver2
from .inner_product_verifier import Proof1, Proof2, Verifier1, Verifier2

__all__ = ["NIProver", "FastNIProver2", "Proof1", "Proof2", "Verifier1", "Verifier2"]
