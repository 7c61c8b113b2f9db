"""ctypes front-end of the CPU oracle (oracle/libhs_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package `opticalflowhs_amd`.
The argument order mirrors cvCalcOpticalFlowHS (OpenCV2.1/include/cv.h:481-483) as the
reference calls it at OpticalFlowHS/OpticalFlowOpenCV.cpp:29.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhs_oracle.so")

TERMCRIT_ITER = 1  # CV_TERMCRIT_ITER (OpenCV2.1/include/cxtypes.h:894)
TERMCRIT_EPS = 2   # CV_TERMCRIT_EPS  (cxtypes.h:896)

_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        f32p = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        L.hs_oracle_cv_8u32f.argtypes = [u8p, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, f32p, f32p, ctypes.c_int, ctypes.c_float,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_double, ip, f32p]
        L.hs_oracle_cv_8u32f.restype = ctypes.c_int
        L.hs_oracle_cv_8u32f_mt.argtypes = [u8p, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, f32p, f32p, ctypes.c_int, ctypes.c_float,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                            ctypes.c_int, ip, f32p]
        L.hs_oracle_cv_8u32f_mt.restype = ctypes.c_int
        L.hs_oracle_cv_derivatives.argtypes = [u8p, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               f32p, f32p, f32p]
        L.hs_oracle_cv_derivatives.restype = ctypes.c_int
        L.hs_oracle_classic.argtypes = [u8p, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, f32p, f32p, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_int]
        L.hs_oracle_classic.restype = ctypes.c_int
        L.hs_oracle_classic_ex.argtypes = L.hs_oracle_classic.argtypes + [ctypes.c_int]
        L.hs_oracle_classic_ex.restype = ctypes.c_int
        L.hs_oracle_classic_derivatives.argtypes = L.hs_oracle_cv_derivatives.argtypes
        L.hs_oracle_classic_derivatives.restype = ctypes.c_int
        L.hs_oracle_bgr2gray_u8.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, u8p,
                                            ctypes.c_int]
        L.hs_oracle_bgr2gray_u8.restype = ctypes.c_int
        L.hs_oracle_box_blur3_u8.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, u8p,
                                             ctypes.c_int]
        L.hs_oracle_box_blur3_u8.restype = ctypes.c_int
        L.hs_oracle_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _f32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _check_frames(prev, curr):
    prev = np.ascontiguousarray(prev)
    curr = np.ascontiguousarray(curr)
    if prev.dtype != np.uint8 or curr.dtype != np.uint8 or prev.ndim != 2 or prev.shape != curr.shape:
        raise ValueError("Source images must have 8uC1 type and equal sizes")
    return prev, curr


def calc_optical_flow_hs(prev, curr, lam, max_iter, epsilon=1e-6, term_type=TERMCRIT_ITER | TERMCRIT_EPS,
                         use_previous=False, velx=None, vely=None, threads=1, return_info=False):
    """Oracle solve.  threads=1: faithful line-buffered form; threads!=1: OpenMP ping-pong form
    (threads<=0 means all cores).  epsilon is rounded through fp32 like cvTermCriteria does
    (cxtypes.h:912)."""
    prev, curr = _check_frames(prev, curr)
    H, W = prev.shape
    if use_previous:
        u = np.ascontiguousarray(velx, dtype=np.float32).copy()
        v = np.ascontiguousarray(vely, dtype=np.float32).copy()
    else:
        u = np.empty((H, W), np.float32)
        v = np.empty((H, W), np.float32)
    if (term_type & TERMCRIT_ITER) == 0 and (term_type & TERMCRIT_EPS) == 0:
        raise ValueError("term_type must include ITER and/or EPS")
    if (term_type & TERMCRIT_EPS) == 0 and max_iter <= 0:
        raise ValueError("ITER-only termination needs max_iter > 0 (the original would not stop)")
    eps = float(np.float32(epsilon))
    it = ctypes.c_int(0)
    le = ctypes.c_float(0)
    L = lib()
    if threads == 1:
        st = L.hs_oracle_cv_8u32f(_u8(prev), _u8(curr), W, W, H, int(bool(use_previous)), _f32(u),
                                  _f32(v), W * 4, float(lam), term_type, max_iter, eps,
                                  ctypes.byref(it), ctypes.byref(le))
    else:
        st = L.hs_oracle_cv_8u32f_mt(_u8(prev), _u8(curr), W, W, H, int(bool(use_previous)),
                                     _f32(u), _f32(v), W * 4, float(lam), term_type, max_iter, eps,
                                     int(threads), ctypes.byref(it), ctypes.byref(le))
    if st != 0:
        raise RuntimeError("oracle status %d" % st)
    if return_info:
        return u, v, it.value, le.value
    return u, v


def derivatives(prev, curr):
    prev, curr = _check_frames(prev, curr)
    H, W = prev.shape
    Ix, Iy, It = (np.empty((H, W), np.float32) for _ in range(3))
    st = lib().hs_oracle_cv_derivatives(_u8(prev), _u8(curr), W, W, H, _f32(Ix), _f32(Iy), _f32(It))
    if st != 0:
        raise RuntimeError("oracle status %d" % st)
    return Ix, Iy, It


def classic_flow(prev, curr, alpha, iterations, use_previous=False, u0=None, v0=None, update_v=True):
    """update_v=False: Kernels.cl as shipped (v never written, Kernels.cl:84-86)."""
    prev, curr = _check_frames(prev, curr)
    H, W = prev.shape
    if use_previous:
        u = np.ascontiguousarray(u0, dtype=np.float32).copy()
        v = np.ascontiguousarray(v0, dtype=np.float32).copy()
    else:
        u = np.empty((H, W), np.float32)
        v = np.empty((H, W), np.float32)
    st = lib().hs_oracle_classic_ex(_u8(prev), _u8(curr), W, W, H, int(bool(use_previous)), _f32(u),
                                    _f32(v), W * 4, float(alpha), int(iterations), int(bool(update_v)))
    if st != 0:
        raise RuntimeError("oracle status %d" % st)
    return u, v


def classic_derivatives(prev, curr):
    prev, curr = _check_frames(prev, curr)
    H, W = prev.shape
    Ex, Ey, Et = (np.empty((H, W), np.float32) for _ in range(3))
    st = lib().hs_oracle_classic_derivatives(_u8(prev), _u8(curr), W, W, H, _f32(Ex), _f32(Ey), _f32(Et))
    if st != 0:
        raise RuntimeError("oracle status %d" % st)
    return Ex, Ey, Et


def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    H, W, C = bgr.shape
    assert C == 3
    g = np.empty((H, W), np.uint8)
    st = lib().hs_oracle_bgr2gray_u8(_u8(bgr), 3 * W, W, H, _u8(g), W)
    if st != 0:
        raise RuntimeError("oracle status %d" % st)
    return g


def box_blur3(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    out = np.empty((H, W), np.uint8)
    st = lib().hs_oracle_box_blur3_u8(_u8(img), W, W, H, _u8(out), W)
    if st != 0:
        raise RuntimeError("oracle status %d" % st)
    return out


def num_threads():
    return lib().hs_oracle_num_threads()
