"""Multi-exponentiation entry point.  `PipSECP256k1.multiexp(points, scalars)` is the call every
prover and verifier makes (reference singleton: src/pippenger/__init__.py:5); on the secp256k1
group it is one call into the GPU engine (bpmi_msm)."""
from ..ec import secp256k1 as _curve
from .group import EC, Group, MultIntModP
from .pippenger import DevicePoints, Pippenger

#: stateless, shared by all callers -- like the reference's module-level instance
PipSECP256k1 = Pippenger(EC(_curve))

__all__ = ["DevicePoints", "EC", "Group", "MultIntModP", "PipSECP256k1", "Pippenger"]
