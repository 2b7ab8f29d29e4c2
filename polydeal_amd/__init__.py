"""polydeal_amd — MI355X-native SIP assembly for agglomerated polytopal DG (one hot path of polyDEAL).

Layout: ``csrc/`` HIP kernels + C ABI (include/polydeal_hip.h), ``_capi`` ctypes binding.
"""
from ._capi import Context, PdhError, Problem, load_library, PDH_BASIS_AGGLODGP, PDH_BASIS_DGQ  # noqa: F401

__version__ = "0.1"
