"""polydeal_amd — MI355X-native SIP assembly for agglomerated polytopal DG (one hot path of polyDEAL).

Layout: ``csrc/`` HIP kernels + C ABI (include/polydeal_hip.h) + the C++ host mirror of the reference's
operator surface; ``_capi`` ctypes binding of the C ABI; ``handler`` Python face of the host mirror.
There is no CPU compute path: without libpolydeal_hip.so / a HIP device the compute calls raise.
"""
from ._capi import Context, PdhError, Problem, load_library, PDH_BASIS_AGGLODGP, PDH_BASIS_DGQ  # noqa: F401
from .handler import (AgglomerationHandler, BackgroundGrid, FE_AggloDGP, FE_DGQ, FiniteElement, HostError,  # noqa: F401
                      SipVariant, assemble_dg_matrix, fill_injection_matrix)

from .postprocess import compute_global_error, interpolate_to_points  # noqa: F401

__version__ = "0.1"
