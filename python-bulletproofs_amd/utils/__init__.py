from .utils import ModP, mod_hash, point_to_bytes, point_to_b64, bytes_to_point, b64_to_point, inner_product
from .transcript import Transcript
from .commitments import commitment, vector_commitment
from .elliptic_curve_hash import elliptic_hash

__all__ = ["ModP", "mod_hash", "point_to_bytes", "point_to_b64", "bytes_to_point", "b64_to_point",
           "inner_product", "Transcript", "commitment", "vector_commitment", "elliptic_hash"]
