"""Import location of the reference (`from src.utils.utils import ModP, mod_hash, ...`); the
definitions live in scalar.py (field element, hash, inner product) and pointcodec.py."""
from .pointcodec import BYTE_LENGTH, CURVE, b64_to_point, bytes_to_point, point_to_b64, point_to_bytes
from .scalar import ModP, inner_product, mod_hash

__all__ = ["BYTE_LENGTH", "CURVE", "ModP", "b64_to_point", "bytes_to_point", "inner_product", "mod_hash",
           "point_to_b64", "point_to_bytes"]
