"""Host-side mirror of the reference's solver interface, over the C ABI (include/hsflow.h).

`calc_optical_flow_hs` has the argument list of cvCalcOpticalFlowHS (OpenCV2.1/include/cv.h:481-483)
as the reference calls it (OpticalFlowHS/OpticalFlowOpenCV.cpp:29); `HSFlow` is the resident-context
form that the reference's HSOpticalFlowOpenCL class plays (setupCL once, then per pair
runDerivatives + iterations x runCLKernels; HSOpticalFlowOpenCL.cpp:744-751).
Everything here runs on the GPU through libhsflow.so; there is no CPU path in this package.
"""
import collections
import ctypes

import numpy as np

from . import _lib
from ._lib import (HsflowError, HsflowInfo, HsflowParams, KERNEL_AUTO, KERNEL_FUSED, KERNEL_SIMPLE, KERNEL_STRIP, KERNEL_FOLD, KERNEL_PERSIST,
                   MODE_CLASSIC, MODE_CV, TERM_EPS, TERM_ITER)

TermCriteria = collections.namedtuple("TermCriteria", "type max_iter epsilon")


def term_criteria(type_, max_iter, epsilon):
    """cvTermCriteria(): epsilon is rounded through fp32 (OpenCV2.1/include/cxtypes.h:904-915)."""
    return TermCriteria(int(type_), int(max_iter), float(np.float32(epsilon)))


def _ptr(x):
    """Device or host pointer of a numpy array / torch tensor / raw int."""
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if isinstance(x, np.ndarray):
        return ctypes.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        return ctypes.c_void_p(x.data_ptr())
    raise TypeError("expected numpy array, tensor or int pointer, got %r" % type(x))


def _is_device_tensor(x):
    return hasattr(x, "is_cuda") and bool(x.is_cuda)


def make_params(lam=1.0, max_iter=100, epsilon=1e-6, term_type=TERM_ITER | TERM_EPS,
                use_previous=False, mode=MODE_CV, alpha=1.0, kernel=KERNEL_AUTO, fuse_steps=0,
                tile_w=0, tile_h=0, threads=0, strip_rows=0, reuse_derivatives=False, use_graph=False,
                profile=False):
    p = HsflowParams()
    _lib.load().hsflow_default_params(ctypes.byref(p))
    p.mode = mode
    p.lambda_ = lam
    p.alpha = alpha
    p.term_type = term_type
    p.max_iter = max_iter
    p.epsilon = epsilon
    p.use_previous = 1 if use_previous else 0
    p.kernel = kernel
    p.fuse_steps = fuse_steps
    p.tile_w = tile_w
    p.tile_h = tile_h
    p.threads = threads
    p.strip_rows = strip_rows
    p.reuse_derivatives = 1 if reuse_derivatives else 0
    p.use_graph = 1 if use_graph else 0
    p.profile = 1 if profile else 0
    return p


class HSFlow(object):
    """A solver context holding `n_pairs` image pairs of one size resident on one GPU."""

    def __init__(self, width, height, n_pairs=1, device=0, stream=None, own_stream=False):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        self.width, self.height, self.n_pairs, self.device = int(width), int(height), int(n_pairs), int(device)
        sp = ctypes.c_void_p(int(stream)) if stream else ctypes.c_void_p()
        st = self._lib.hsflow_create(ctypes.byref(self._h), self.device, self.width, self.height,
                                     self.n_pairs, sp, 1 if own_stream else 0)
        if st != _lib.OK:
            msg = self._lib.hsflow_last_error(None).decode()
            self._h = ctypes.c_void_p()
            raise HsflowError(st, msg)

    # -- helpers -------------------------------------------------------------------------
    def _check(self, st):
        if st != _lib.OK:
            raise HsflowError(st, self._lib.hsflow_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.hsflow_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_row_origin(self, first_row):
        """The context holds rows [first_row, first_row + height) of a larger frame (row slabs): keeps the
        checkerboard phase of the update's summation order that of the whole frame (bit-identical results)."""
        self._check(self._lib.hsflow_set_row_origin(self._h, int(first_row)))

    # -- frames in -----------------------------------------------------------------------
    def set_frames(self, prev, curr, pair=0):
        """u8 single-channel frames: host numpy arrays (H, W) or CUDA tensors (H, W)."""
        if _is_device_tensor(prev):
            if str(prev.dtype) != "torch.uint8" or str(curr.dtype) != "torch.uint8":
                raise TypeError("Source images must have 8uC1 type")
            if tuple(prev.shape) != (self.height, self.width) or tuple(curr.shape) != (self.height, self.width):
                raise ValueError("frame shape must be (height, width)")
            self._check(self._lib.hsflow_set_frames_u8_device(
                self._h, pair, _ptr(prev), prev.stride(0), _ptr(curr), curr.stride(0)))
            return
        prev = np.asarray(prev)
        curr = np.asarray(curr)
        if prev.dtype != np.uint8 or curr.dtype != np.uint8:
            raise TypeError("Source images must have 8uC1 type")
        if prev.shape != (self.height, self.width) or curr.shape != (self.height, self.width):
            raise ValueError("frame shape must be (height, width)")
        if prev.strides[1] != 1:
            prev = np.ascontiguousarray(prev)
        if curr.strides[1] != 1:
            curr = np.ascontiguousarray(curr)
        self._check(self._lib.hsflow_set_frames_u8(self._h, pair, _ptr(prev), prev.strides[0],
                                                   _ptr(curr), curr.strides[0]))

    def set_frames_bgr(self, prev_bgr, curr_bgr, blur=True, pair=0):
        prev_bgr = np.ascontiguousarray(prev_bgr, dtype=np.uint8)
        curr_bgr = np.ascontiguousarray(curr_bgr, dtype=np.uint8)
        if prev_bgr.shape != (self.height, self.width, 3) or curr_bgr.shape != prev_bgr.shape:
            raise ValueError("colour frame shape must be (height, width, 3)")
        self._check(self._lib.hsflow_set_frames_bgr8(self._h, pair, _ptr(prev_bgr), prev_bgr.strides[0],
                                                     _ptr(curr_bgr), curr_bgr.strides[0], 1 if blur else 0))

    def set_frames_gray_blur(self, prev, curr, pair=0):
        """u8 gray frames, 3x3 box blur on the GPU first (the reference CPU route's cvSmooth)."""
        prev = np.ascontiguousarray(prev, dtype=np.uint8)
        curr = np.ascontiguousarray(curr, dtype=np.uint8)
        if prev.shape != (self.height, self.width) or curr.shape != prev.shape:
            raise ValueError("frame shape must be (height, width)")
        self._check(self._lib.hsflow_set_frames_gray8_blur(self._h, pair, _ptr(prev), prev.strides[0],
                                                           _ptr(curr), curr.strides[0]))

    def push_frame(self, nxt, pair=0):
        nxt = np.ascontiguousarray(nxt, dtype=np.uint8)
        if nxt.shape != (self.height, self.width):
            raise ValueError("frame shape must be (height, width)")
        self._check(self._lib.hsflow_push_frame_u8(self._h, pair, _ptr(nxt), nxt.strides[0]))

    # -- solve ---------------------------------------------------------------------------
    def make_params(self, **kw):
        """hsflow_params with the defaults of hsflow_default_params; see `make_params` (module)."""
        return make_params(**kw)

    def solve(self, params=None, **kw):
        p = params if params is not None else self.make_params(**kw)
        self._check(self._lib.hsflow_solve(self._h, ctypes.byref(p)))
        return self.info()

    def solve_async(self, params=None, **kw):
        p = params if params is not None else self.make_params(**kw)
        self._check(self._lib.hsflow_solve_async(self._h, ctypes.byref(p)))

    def set_eps_rows(self, first_row, rows):
        """Only the changes of rows [first_row, first_row + rows) count for Eps / the witness (rows <= 0: the whole frame)."""
        self._check(self._lib.hsflow_set_eps_rows(self._h, int(first_row), int(rows)))

    def solve_probe(self, params=None, **kw):
        """Exactly max_iter sweeps, nothing stops them; returns the Eps of every sweep (fp32 array)."""
        p = params if params is not None else self.make_params(**kw)
        out = np.empty(int(p.max_iter), np.float32)
        self._check(self._lib.hsflow_solve_probe(self._h, ctypes.byref(p), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return out

    def take_verdict(self):
        """The early-stop check an asynchronous ITER|EPS solve owes, without acting on it: True = "no early stop" proven."""
        v = ctypes.c_int(0)
        self._check(self._lib.hsflow_take_verdict(self._h, ctypes.byref(v)))
        return bool(v.value)

    def synchronize(self):
        self._check(self._lib.hsflow_synchronize(self._h))

    # -- results out ---------------------------------------------------------------------
    def flow(self, pair=0):
        u = np.empty((self.height, self.width), np.float32)
        v = np.empty((self.height, self.width), np.float32)
        self._check(self._lib.hsflow_get_flow(self._h, pair, _ptr(u), u.strides[0], _ptr(v), v.strides[0]))
        return u, v

    def flow_rows_to(self, u_dev, v_dev, row0, nrows, pair=0):
        """Copy flow rows [row0, row0+nrows) into CUDA tensors of shape (nrows, width)."""
        self._check(self._lib.hsflow_get_flow_device(self._h, pair, row0, nrows, _ptr(u_dev),
                                                     u_dev.stride(0) * 4, _ptr(v_dev), v_dev.stride(0) * 4))

    def set_flow_rows_from(self, u_dev, v_dev, row0, nrows, pair=0):
        self._check(self._lib.hsflow_set_flow_device(self._h, pair, row0, nrows, _ptr(u_dev),
                                                     u_dev.stride(0) * 4, _ptr(v_dev), v_dev.stride(0) * 4))

    def derivatives(self, pair=0):
        d = [np.empty((self.height, self.width), np.float32) for _ in range(3)]
        self._check(self._lib.hsflow_get_derivatives(self._h, pair, _ptr(d[0]), _ptr(d[1]), _ptr(d[2]),
                                                     d[0].strides[0]))
        return tuple(d)

    def frames(self, pair=0):
        a = np.empty((self.height, self.width), np.uint8)
        b = np.empty((self.height, self.width), np.uint8)
        self._check(self._lib.hsflow_get_frames_u8(self._h, pair, _ptr(a), a.strides[0], _ptr(b), b.strides[0]))
        return a, b

    def info(self):
        i = HsflowInfo()
        i.struct_size = ctypes.sizeof(HsflowInfo)
        self._check(self._lib.hsflow_get_info(self._h, ctypes.byref(i)))
        return {name: getattr(i, name) for name, _ in HsflowInfo._fields_}


def plan_query(width, height, n_pairs=1, params=None, **kw):
    """The launch plan hsflow_solve would use for this size and these parameters (no device needed)."""
    lib = _lib.load()
    p = params if params is not None else make_params(**kw)
    i = HsflowInfo()
    i.struct_size = ctypes.sizeof(HsflowInfo)
    st = lib.hsflow_plan_query(int(width), int(height), int(n_pairs), ctypes.byref(p), ctypes.byref(i))
    if st:
        raise HsflowError(st, (lib.hsflow_last_error(None) or b"").decode())
    return {name: getattr(i, name) for name, _ in HsflowInfo._fields_}


def calc_optical_flow_hs(prev, curr, use_previous, velx, vely, lam, criteria, device=0, **tuning):
    """cvCalcOpticalFlowHS(prev, curr, use_previous, velx, vely, lambda, criteria) on the GPU.

    prev/curr: (H, W) uint8; velx/vely: (H, W) float32, written in place (read first when
    use_previous).  criteria: TermCriteria / (type, max_iter, epsilon).  Raises TypeError /
    ValueError for what the original rejects with "Source images must have 8uC1 type and
    destination images must have 32fC1 type" and for mismatched sizes.
    """
    prev = np.asarray(prev)
    curr = np.asarray(curr)
    if prev.dtype != np.uint8 or curr.dtype != np.uint8 or prev.ndim != 2 or curr.ndim != 2:
        raise TypeError("Source images must have 8uC1 type and destination images must have 32fC1 type")
    if not (isinstance(velx, np.ndarray) and isinstance(vely, np.ndarray)) or \
            velx.dtype != np.float32 or vely.dtype != np.float32 or velx.ndim != 2 or vely.ndim != 2:
        raise TypeError("Source images must have 8uC1 type and destination images must have 32fC1 type")
    if prev.shape != curr.shape or velx.shape != prev.shape or vely.shape != prev.shape:
        raise ValueError("images and velocity fields must have equal sizes")
    ctype, max_iter, eps = criteria
    H, W = prev.shape
    with HSFlow(W, H, 1, device=device, own_stream=True) as ctx:
        ctx.set_frames(prev, curr)
        if use_previous:
            import torch  # device staging only
            ud = torch.from_numpy(np.ascontiguousarray(velx)).to("cuda:%d" % device)
            vd = torch.from_numpy(np.ascontiguousarray(vely)).to("cuda:%d" % device)
            torch.cuda.synchronize(device)
            ctx.set_flow_rows_from(ud, vd, 0, H)
            ctx.synchronize()
        info = ctx.solve(lam=lam, max_iter=max_iter, epsilon=eps, term_type=ctype,
                         use_previous=bool(use_previous), **tuning)
        u, v = ctx.flow()
    velx[...] = u
    vely[...] = v
    return info
