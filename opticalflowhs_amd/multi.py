"""Several GPUs from ONE host process: mirrors of hsflow_multi_* (independent pairs, BASELINE config C4) and
hsflow_slab_* (one frame in row slabs with device-to-device halo exchange, config C5) of include/hsflow.h.

The one-process-per-GPU forms (torch.distributed, RCCL) are bench.py's pair sharding and slab.py; these are what a
C / C++ host such as the reference's main.cpp would call."""
import ctypes

import numpy as np

from . import _lib
from ._lib import HsflowError, TERM_ITER
from .solver import make_params


def _devs(devices):
    arr = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
    return arr, len(devices)


class MultiPairs(object):
    """Independent pairs over `devices` (a device may appear more than once): pair i -> devices[i % n]."""

    def __init__(self, width, height, devices=(0,), depth=4):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        self.width, self.height = int(width), int(height)
        arr, n = _devs(devices)
        st = self._lib.hsflow_multi_create(ctypes.byref(self._h), arr, n, self.width, self.height, int(depth))
        if st:
            self._h = None
            raise HsflowError(st, (self._lib.hsflow_multi_last_error(None) or b"").decode())
        self._held = {}

    def _check(self, st):
        if st:
            raise HsflowError(st, (self._lib.hsflow_multi_last_error(self._h) or b"").decode())

    def submit(self, prev, curr, u_out, v_out, params=None, **kw):
        for a, dt in ((prev, np.uint8), (curr, np.uint8), (u_out, np.float32), (v_out, np.float32)):
            if not isinstance(a, np.ndarray) or a.dtype != dt or a.shape != (self.height, self.width) or a.strides[1] != a.itemsize:
                raise ValueError("buffers must be (height, width) arrays: u8 frames, fp32 flow, unit column stride")
        if params is None:
            kw.setdefault("term_type", TERM_ITER)
            params = make_params(**kw)
        t = ctypes.c_uint64()
        self._check(self._lib.hsflow_multi_submit(
            self._h, _lib.FRAMES_GRAY8, ctypes.c_void_p(prev.ctypes.data), prev.strides[0], ctypes.c_void_p(curr.ctypes.data), curr.strides[0],
            ctypes.c_void_p(u_out.ctypes.data), u_out.strides[0], ctypes.c_void_p(v_out.ctypes.data), v_out.strides[0],
            ctypes.byref(params), ctypes.byref(t)))
        self._held[t.value] = (prev, curr, u_out, v_out)
        return t.value

    def wait(self, ticket):
        self._check(self._lib.hsflow_multi_wait(self._h, int(ticket)))
        self._held.pop(int(ticket), None)

    def drain(self):
        self._check(self._lib.hsflow_multi_drain(self._h))
        self._held.clear()

    def close(self):
        if self._h is not None:
            self._lib.hsflow_multi_destroy(self._h)
            self._h = None
            self._held.clear()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SlabFrame(object):
    """One width x height frame in len(devices) row slabs, slab k on devices[k]."""

    def __init__(self, width, height, devices=(0,), halo=16, overlapped=False):
        """overlapped: two sub-slabs per listed device (one's peer copies run under the other's sweeps)."""
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        self.width, self.height = int(width), int(height)
        arr, n = _devs(devices)
        create = self._lib.hsflow_slab_create_overlapped if overlapped else self._lib.hsflow_slab_create
        st = create(ctypes.byref(self._h), arr, n, self.width, self.height, int(halo))
        if st:
            self._h = None
            raise HsflowError(st, (self._lib.hsflow_slab_last_error(None) or b"").decode())

    def _check(self, st):
        if st:
            raise HsflowError(st, (self._lib.hsflow_slab_last_error(self._h) or b"").decode())

    def rows(self):
        out = []
        for k in range(self._lib.hsflow_slab_count(self._h)):
            lo, hi = ctypes.c_int(), ctypes.c_int()
            self._check(self._lib.hsflow_slab_rows(self._h, k, ctypes.byref(lo), ctypes.byref(hi)))
            out.append((lo.value, hi.value))
        return out

    def set_frames(self, prev, curr):
        prev = np.ascontiguousarray(prev, dtype=np.uint8)
        curr = np.ascontiguousarray(curr, dtype=np.uint8)
        if prev.shape != (self.height, self.width) or curr.shape != prev.shape:
            raise ValueError("frame shape must be (height, width)")
        self._check(self._lib.hsflow_slab_set_frames_u8(self._h, ctypes.c_void_p(prev.ctypes.data), prev.strides[0],
                                                        ctypes.c_void_p(curr.ctypes.data), curr.strides[0]))

    def solve(self, params=None, **kw):
        if params is None:
            kw.setdefault("term_type", TERM_ITER)
            params = make_params(**kw)
        self._check(self._lib.hsflow_slab_solve(self._h, ctypes.byref(params)))
        return self._lib.hsflow_slab_exchanges(self._h)

    def iterations_done(self):
        """Sweeps of the last solve (fewer than max_iter when the early stop of ITER|EPS fired)."""
        return self._lib.hsflow_slab_iterations_done(self._h)

    def eps_measured(self):
        """True if the last ITER|EPS solve had to measure Eps sweep by sweep (no slab's witness vouched for some chunk)."""
        return bool(self._lib.hsflow_slab_eps_measured(self._h))

    def flow(self):
        u = np.empty((self.height, self.width), np.float32)
        v = np.empty((self.height, self.width), np.float32)
        self._check(self._lib.hsflow_slab_get_flow(self._h, ctypes.c_void_p(u.ctypes.data), u.strides[0], ctypes.c_void_p(v.ctypes.data), v.strides[0]))
        return u, v

    def close(self):
        if self._h is not None:
            self._lib.hsflow_slab_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
