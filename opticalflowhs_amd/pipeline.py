"""Independent pairs streamed through one GPU: host frames in, host flow out (BASELINE config C4).

Mirror of `hsflow_pipeline_*` (include/hsflow.h).  The reference handles a pair as blocking write ->
derivatives -> iterations -> blocking read (HSOpticalFlowOpenCL.cpp:744-767); the pipeline keeps
`depth` pairs in flight on separate streams so that the PCIe copies of the neighbouring pairs run
beside the solve of the current one.  Host buffers should be page-locked (`pinned_empty`).
"""
import ctypes
import weakref

import numpy as np

from . import _lib
from ._lib import HsflowError, HsflowInfo, TERM_ITER
from .solver import make_params


def pinned_empty(shape, dtype):
    """numpy array over page-locked host memory (hsflow_host_alloc); freed with the array."""
    lib = _lib.load()
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    p = ctypes.c_void_p()
    st = lib.hsflow_host_alloc(ctypes.byref(p), max(n, 1))
    if st:
        raise HsflowError(st, (lib.hsflow_last_error(None) or b"").decode())
    buf = (ctypes.c_ubyte * max(n, 1)).from_address(p.value)
    weakref.finalize(buf, lib.hsflow_host_free, ctypes.c_void_p(p.value))  # runs when the last view is gone
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class _DeviceView(object):
    """fp32 device memory described for torch.as_tensor (zero copy)."""

    def __init__(self, ptr, shape, strides):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "strides": tuple(strides), "version": 2}


class PairPipeline(object):
    """`depth` single-pair contexts used round-robin.  Termination: ITER (default) or ITER|EPS."""

    def __init__(self, width, height, depth=4, device=0, lanes=None):
        """lanes: streams the slots are spread over (default: one per slot -- host-memory pairs; device-resident streams
        want 2 lanes with 4 - 8 slots, hsflow_pipeline_create_lanes)."""
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        self.width, self.height = int(width), int(height)
        st = self._lib.hsflow_pipeline_create_lanes(ctypes.byref(self._h), int(device), self.width, self.height, int(depth),
                                                    int(depth if lanes is None else lanes))
        if st:
            self._h = None
            raise HsflowError(st, (self._lib.hsflow_pipeline_last_error(None) or b"").decode())
        self.depth = self._lib.hsflow_pipeline_depth(self._h)
        self._held = {}  # ticket -> arrays kept alive while the device may touch them

    def _check(self, st):
        if st:
            raise HsflowError(st, (self._lib.hsflow_pipeline_last_error(self._h) or b"").decode())

    def submit(self, prev, curr, u_out, v_out, params=None, frames="gray", **kw):
        """Enqueues upload -> [pre-processing] -> solve -> download of one pair; returns its ticket.  The four
        arrays belong to the pipeline until `wait(ticket)` (or `drain()`) returned.
        frames: "gray" (u8, as is), "gray_blur" (3x3 box blur on the device), "bgr" / "bgr_blur" ((H, W, 3) u8)."""
        fmt = {"gray": _lib.FRAMES_GRAY8, "gray_blur": _lib.FRAMES_GRAY8_BLUR, "bgr": _lib.FRAMES_BGR8, "bgr_blur": _lib.FRAMES_BGR8_BLUR}[frames]
        colour = fmt >= _lib.FRAMES_BGR8
        for a in (prev, curr):
            want = (self.height, self.width, 3) if colour else (self.height, self.width)
            if not isinstance(a, np.ndarray) or a.dtype != np.uint8 or a.shape != want or a.strides[-1] != 1 or (colour and a.strides[1] != 3):
                raise ValueError("frames must be u8 arrays of shape %r with packed pixels" % (want,))
        for a, dt in ((u_out, np.float32), (v_out, np.float32)):
            if not isinstance(a, np.ndarray) or a.dtype != dt or a.shape != (self.height, self.width) or a.strides[1] != a.itemsize:
                raise ValueError("pipeline buffers must be (height, width) arrays: u8 frames, fp32 flow, unit column stride")
        if not (u_out.flags.writeable and v_out.flags.writeable):
            raise ValueError("flow outputs must be writeable")
        if params is None:
            kw.setdefault("term_type", TERM_ITER)
            params = make_params(**kw)
        t = ctypes.c_uint64()
        self._check(self._lib.hsflow_pipeline_submit_ex(
            self._h, fmt, ctypes.c_void_p(prev.ctypes.data), prev.strides[0], ctypes.c_void_p(curr.ctypes.data), curr.strides[0],
            ctypes.c_void_p(u_out.ctypes.data), u_out.strides[0], ctypes.c_void_p(v_out.ctypes.data), v_out.strides[0],
            ctypes.byref(params), ctypes.byref(t)))
        self._held.pop(t.value - self.depth, None)  # that slot was waited for inside submit
        self._held[t.value] = (prev, curr, u_out, v_out)
        return t.value

    def submit_device(self, prev, curr, params=None, **kw):
        """A pair that already lies in device memory (CUDA uint8 tensors of shape (height, width), unit column stride):
        device-to-device copy into the slot -> solve; the flow stays in the slot (`flow_device`).  The tensors are held
        until the ticket has been waited for."""
        for a in (prev, curr):
            if not (hasattr(a, "is_cuda") and a.is_cuda) or str(a.dtype) != "torch.uint8" or tuple(a.shape) != (self.height, self.width) or a.stride(1) != 1:
                raise ValueError("device frames must be CUDA uint8 tensors of shape (height, width) with unit column stride")
        if params is None:
            kw.setdefault("term_type", TERM_ITER)
            params = make_params(**kw)
        t = ctypes.c_uint64()
        self._check(self._lib.hsflow_pipeline_submit_device(self._h, ctypes.c_void_p(prev.data_ptr()), prev.stride(0), ctypes.c_void_p(curr.data_ptr()),
                                                            curr.stride(0), ctypes.byref(params), ctypes.byref(t)))
        self._held.pop(t.value - self.depth, None)
        self._held[t.value] = (prev, curr)
        return t.value

    def flow_device(self, ticket, copy=True):
        """wait(ticket) + that pair's flow as two CUDA tensors of shape (height, width).  copy=False: views of the slot's
        own planes (row stride = the context's pitch), valid until `depth` more pairs have been submitted."""
        import torch
        pu, pv, sb = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self._lib.hsflow_pipeline_flow_device(self._h, int(ticket), ctypes.byref(pu), ctypes.byref(pv), ctypes.byref(sb)))
        self._held.pop(int(ticket), None)
        out = []
        for p in (pu, pv):
            view = torch.as_tensor(_DeviceView(p.value, (self.height, self.width), (sb.value, 4)), device="cuda")
            out.append(view.clone() if copy else view)
        return tuple(out)

    def wait(self, ticket):
        self._check(self._lib.hsflow_pipeline_wait(self._h, int(ticket)))
        self._held.pop(int(ticket), None)

    def info(self, ticket):
        """wait(ticket) + the solver's report for that pair (iterations_done, last_eps, eps_rerun, ...)."""
        i = HsflowInfo()
        i.struct_size = ctypes.sizeof(HsflowInfo)
        self._check(self._lib.hsflow_pipeline_info(self._h, int(ticket), ctypes.byref(i)))
        self._held.pop(int(ticket), None)
        return {name: getattr(i, name) for name, _ in HsflowInfo._fields_}

    def drain(self):
        self._check(self._lib.hsflow_pipeline_drain(self._h))
        self._held.clear()

    def close(self):
        if self._h is not None:
            self._lib.hsflow_pipeline_destroy(self._h)
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
