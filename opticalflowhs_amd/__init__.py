"""opticalflowhs_amd -- MI355X-native Horn-Schunck optical flow (hot path of miczi/OpticalFlowHS).

Layout: csrc/ holds the HIP kernels and the C ABI (include/hsflow.h); solver.py is the host-side
mirror of the reference interface; pipeline.py streams pairs through one GPU with overlapped copies, slab.py is the multi-GPU row-slab driver.  The package has
no CPU fallback: importing it without libhsflow.so raises.
"""
from . import _lib
from ._lib import (HsflowError, KERNEL_AUTO, KERNEL_FUSED, KERNEL_SIMPLE, KERNEL_STRIP, KERNEL_FOLD, KERNEL_PERSIST, MODE_CLASSIC, MODE_CLASSIC_AS_SHIPPED, MODE_CV,
                   TERM_EPS, TERM_ITER)
from .solver import HSFlow, TermCriteria, calc_optical_flow_hs, make_params, plan_query, term_criteria
from .pipeline import PairPipeline, pinned_empty
from .multi import MultiPairs, SlabFrame

__all__ = ["HSFlow", "PairPipeline", "MultiPairs", "SlabFrame", "pinned_empty", "make_params", "plan_query", "TermCriteria", "term_criteria", "calc_optical_flow_hs", "HsflowError",
           "TERM_ITER", "TERM_EPS", "MODE_CV", "MODE_CLASSIC", "MODE_CLASSIC_AS_SHIPPED", "KERNEL_AUTO", "KERNEL_SIMPLE",
           "KERNEL_FUSED", "KERNEL_STRIP", "KERNEL_FOLD", "KERNEL_PERSIST", "OP_SLOTS_PER_PIXEL_SWEEP"]

# Wave64 VALU lane-operations one pixel costs per Jacobi sweep in the multi-sweep kernels' arithmetic (csrc/hs_kernels_strip.hip.h,
# cross_rows + strip_row_update): 2 + 2 additions for the two neighbour sums (one shared diagonal cross sum and one combining
# addition per plane), 4 fused multiply-adds for the update, 1 multiplication that carries the constant term to the next sweep's
# scale.  bench.py prices the VALU-issue roofline with it.  The straightforward form of the same update (3 + 3 additions, round 1)
# costs 11; bench.py reports the fraction at that accounting beside it so that rounds stay comparable.
OP_SLOTS_PER_PIXEL_SWEEP = 9
OP_SLOTS_PER_PIXEL_SWEEP_STRAIGHTFORWARD = 11

_lib.load()  # fail loudly at import time if the HIP library is absent
