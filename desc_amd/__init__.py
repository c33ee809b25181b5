"""desc_amd -- MI355X-native implementation of the DESC projected-gradient hot path
(ColeWyeth/DESC, Algorithms/DESC_PGD.m) behind the reference's call signatures.

The numerical work lives in libdesc_amd.so (hand-written HIP for gfx950, C ABI in
include/desc_amd.h).  Importing the package does not need a GPU; creating a solver
does, and there is no CPU fallback.
"""
from .stepsize import ConstantStepSize, HybridGradient, PiecewiseStepSize  # noqa: F401
from .algorithms import CEMP, DESC, DESC_PGD, GCW, Rotation_Alignment, Spectral  # noqa: F401
from .models import Nonuniform_Topology, Uniform_Topology  # noqa: F401

__all__ = ["DESC", "DESC_PGD", "CEMP", "Spectral", "GCW", "Rotation_Alignment", "ConstantStepSize", "PiecewiseStepSize", "HybridGradient",
           "Uniform_Topology", "Nonuniform_Topology"]
