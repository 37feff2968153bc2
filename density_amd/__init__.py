"""density_amd — MI355X (gfx950) implementation of density's dictionary-hash encode/decode hot path.

`Chameleon`, `Cheetah`, `Lion` mirror the reference crate's codec types (see codec.py); `container` is the chunked,
data-parallel API; everything executes in libdensity_hip.so (HIP kernels), never on the CPU.
"""
from .codec import BY_NAME, Chameleon, Cheetah, DecodeError, EncodeError, Lion  # noqa: F401
from . import container  # noqa: F401

__all__ = ["Chameleon", "Cheetah", "Lion", "EncodeError", "DecodeError", "container", "BY_NAME"]
