from ..ec import secp256k1
from .group import EC, Group, MultIntModP
from .pippenger import DevicePoints, Pippenger

PipSECP256k1 = Pippenger(EC(secp256k1))     # reference: src/pippenger/__init__.py:5

__all__ = ["Pippenger", "EC", "Group", "MultIntModP", "PipSECP256k1", "DevicePoints"]
