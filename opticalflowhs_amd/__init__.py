"""opticalflowhs_amd -- MI355X-native Horn-Schunck optical flow (hot path of miczi/OpticalFlowHS).

Layout: csrc/ holds the HIP kernels and the C ABI (include/hsflow.h); solver.py is the host-side
mirror of the reference interface; slab.py / batch.py are the multi-GPU drivers.  The package has
no CPU fallback: importing it without libhsflow.so raises.
"""
from . import _lib
from ._lib import (HsflowError, KERNEL_AUTO, KERNEL_FUSED, KERNEL_SIMPLE, KERNEL_STRIP, KERNEL_FOLD, MODE_CLASSIC, MODE_CV,
                   TERM_EPS, TERM_ITER)
from .solver import HSFlow, TermCriteria, calc_optical_flow_hs, term_criteria

__all__ = ["HSFlow", "TermCriteria", "term_criteria", "calc_optical_flow_hs", "HsflowError",
           "TERM_ITER", "TERM_EPS", "MODE_CV", "MODE_CLASSIC", "KERNEL_AUTO", "KERNEL_SIMPLE",
           "KERNEL_FUSED", "KERNEL_STRIP", "KERNEL_FOLD"]

_lib.load()  # fail loudly at import time if the HIP library is absent
