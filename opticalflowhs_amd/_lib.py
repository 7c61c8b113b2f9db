"""ctypes binding of libhsflow.so (the C ABI declared in include/hsflow.h).

There is NO fallback: if the HIP library is missing or a symbol is absent this raises.  The
product path never touches oracle/.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HSFLOW_LIB_PATH") or os.path.join(HERE, "libhsflow.so")  # override: diagnostic builds

# status codes / enums of include/hsflow.h
OK, E_ARG, E_SIZE, E_DEVICE, E_OOM, E_STATE, E_NOTERM = range(7)
TERM_ITER, TERM_EPS = 1, 2
MODE_CV, MODE_CLASSIC, MODE_CLASSIC_AS_SHIPPED = 0, 1, 2
KERNEL_AUTO, KERNEL_SIMPLE, KERNEL_FUSED, KERNEL_STRIP, KERNEL_FOLD, KERNEL_PERSIST = 0, 1, 2, 3, 4, 5
FRAMES_GRAY8, FRAMES_GRAY8_BLUR, FRAMES_BGR8, FRAMES_BGR8_BLUR = 0, 1, 2, 3


class HsflowParams(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("mode", ctypes.c_int32),
                ("lambda_", ctypes.c_float), ("alpha", ctypes.c_float),
                ("term_type", ctypes.c_int32), ("max_iter", ctypes.c_int32),
                ("epsilon", ctypes.c_double), ("use_previous", ctypes.c_int32),
                ("kernel", ctypes.c_int32), ("fuse_steps", ctypes.c_int32),
                ("tile_w", ctypes.c_int32), ("tile_h", ctypes.c_int32),
                ("threads", ctypes.c_int32), ("strip_rows", ctypes.c_int32),
                ("reuse_derivatives", ctypes.c_int32), ("use_graph", ctypes.c_int32),
                ("profile", ctypes.c_int32)]


class HsflowInfo(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("width", ctypes.c_int32),
                ("height", ctypes.c_int32), ("n_pairs", ctypes.c_int32), ("pitch", ctypes.c_int32),
                ("iterations_done", ctypes.c_int32), ("last_eps", ctypes.c_float),
                ("kernel", ctypes.c_int32), ("fuse_steps", ctypes.c_int32),
                ("tile_w", ctypes.c_int32), ("tile_h", ctypes.c_int32), ("threads", ctypes.c_int32),
                ("groups_per_thread", ctypes.c_int32), ("tiles", ctypes.c_int32),
                ("lds_bytes", ctypes.c_int32), ("jacobi_launches", ctypes.c_int32),
                ("deriv_ms", ctypes.c_float), ("jacobi_ms", ctypes.c_float),
                ("solve_ms", ctypes.c_float), ("eps_rerun", ctypes.c_int32),
                ("deriv_fused", ctypes.c_int32), ("persistent", ctypes.c_int32)]


_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
_pp = ctypes.POINTER(HsflowParams)

# name -> (restype, argtypes).  tests/test_abi.py checks this table against include/hsflow.h.
PROTOTYPES = {
    "hsflow_default_params": (None, [_pp]),
    "hsflow_create": (_i, [ctypes.POINTER(_vp), _i, _i, _i, _i, _vp, _i]),
    "hsflow_destroy": (_i, [_vp]),
    "hsflow_set_row_origin": (_i, [_vp, _i]),
    "hsflow_set_cu_share": (_i, [_vp, _i]),
    "hsflow_set_async_reduce": (_i, [_vp, _i]),
    "hsflow_set_eps_rows": (_i, [_vp, _i, _i]),
    "hsflow_solve_probe": (_i, [_vp, _pp, ctypes.POINTER(ctypes.c_float)]),
    "hsflow_take_verdict": (_i, [_vp, ctypes.POINTER(_i)]),
    "hsflow_set_frames_u8": (_i, [_vp, _i, _vp, _sz, _vp, _sz]),
    "hsflow_set_frames_u8_async": (_i, [_vp, _i, _vp, _sz, _vp, _sz]),
    "hsflow_set_frames_u8_device": (_i, [_vp, _i, _vp, _sz, _vp, _sz]),
    "hsflow_set_frames_bgr8": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _i]),
    "hsflow_set_frames_gray8_blur": (_i, [_vp, _i, _vp, _sz, _vp, _sz]),
    "hsflow_set_frames_bgr8_async": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _i]),
    "hsflow_set_frames_gray8_blur_async": (_i, [_vp, _i, _vp, _sz, _vp, _sz]),
    "hsflow_push_frame_u8": (_i, [_vp, _i, _vp, _sz]),
    "hsflow_solve": (_i, [_vp, _pp]),
    "hsflow_solve_async": (_i, [_vp, _pp]),
    "hsflow_synchronize": (_i, [_vp]),
    "hsflow_wait_solve": (_i, [_vp]),
    "hsflow_get_flow": (_i, [_vp, _i, _vp, _sz, _vp, _sz]),
    "hsflow_get_flow_async": (_i, [_vp, _i, _vp, _sz, _vp, _sz]),
    "hsflow_flow_view_device": (_i, [_vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_sz)]),
    "hsflow_get_flow_device": (_i, [_vp, _i, _i, _i, _vp, _sz, _vp, _sz]),
    "hsflow_set_flow_device": (_i, [_vp, _i, _i, _i, _vp, _sz, _vp, _sz]),
    "hsflow_get_derivatives": (_i, [_vp, _i, _vp, _vp, _vp, _sz]),
    "hsflow_get_frames_u8": (_i, [_vp, _i, _vp, _sz, _vp, _sz]),
    "hsflow_get_info": (_i, [_vp, ctypes.POINTER(HsflowInfo)]),
    "hsflow_get_info_ex": (_i, [_vp, ctypes.POINTER(HsflowInfo), _i]),
    "hsflow_last_error": (ctypes.c_char_p, [_vp]),
    "hsflow_status_string": (ctypes.c_char_p, [_i]),
    "hsflow_version": (_i, []),
    "hsflow_device_count": (_i, [ctypes.POINTER(_i)]),
    "hsflow_release_cached": (None, []),
    "hsflow_plan_query": (_i, [_i, _i, _i, _pp, ctypes.POINTER(HsflowInfo)]),
    "hsflow_host_alloc": (_i, [ctypes.POINTER(_vp), _sz]),
    "hsflow_host_free": (_i, [_vp]),
    "hsflow_host_register": (_i, [_vp, _sz]),
    "hsflow_host_unregister": (_i, [_vp]),
    "hsflow_pipeline_create": (_i, [ctypes.POINTER(_vp), _i, _i, _i, _i]),
    "hsflow_pipeline_create_lanes": (_i, [ctypes.POINTER(_vp), _i, _i, _i, _i, _i]),
    "hsflow_pipeline_destroy": (_i, [_vp]),
    "hsflow_pipeline_submit": (_i, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _pp, ctypes.POINTER(ctypes.c_uint64)]),
    "hsflow_pipeline_submit_ex": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _pp, ctypes.POINTER(ctypes.c_uint64)]),
    "hsflow_pipeline_submit_device": (_i, [_vp, _vp, _sz, _vp, _sz, _pp, ctypes.POINTER(ctypes.c_uint64)]),
    "hsflow_pipeline_flow_device": (_i, [_vp, ctypes.c_uint64, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_sz)]),
    "hsflow_pipeline_wait": (_i, [_vp, ctypes.c_uint64]),
    "hsflow_pipeline_info": (_i, [_vp, ctypes.c_uint64, ctypes.POINTER(HsflowInfo)]),
    "hsflow_pipeline_drain": (_i, [_vp]),
    "hsflow_pipeline_depth": (_i, [_vp]),
    "hsflow_pipeline_last_error": (ctypes.c_char_p, [_vp]),
    "hsflow_multi_create": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i, _i]),
    "hsflow_multi_destroy": (_i, [_vp]),
    "hsflow_multi_devices": (_i, [_vp]),
    "hsflow_multi_submit": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _pp, ctypes.POINTER(ctypes.c_uint64)]),
    "hsflow_multi_wait": (_i, [_vp, ctypes.c_uint64]),
    "hsflow_multi_drain": (_i, [_vp]),
    "hsflow_multi_last_error": (ctypes.c_char_p, [_vp]),
    "hsflow_slab_create": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i, _i]),
    "hsflow_slab_destroy": (_i, [_vp]),
    "hsflow_slab_count": (_i, [_vp]),
    "hsflow_slab_rows": (_i, [_vp, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "hsflow_slab_set_frames_u8": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "hsflow_slab_solve": (_i, [_vp, _pp]),
    "hsflow_slab_exchanges": (_i, [_vp]),
    "hsflow_slab_iterations_done": (_i, [_vp]),
    "hsflow_slab_eps_measured": (_i, [_vp]),
    "hsflow_slab_create_overlapped": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_i), _i, _i, _i, _i]),
    "hsflow_slab_get_flow": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "hsflow_slab_last_error": (ctypes.c_char_p, [_vp]),
    "hsflow_calc_optical_flow_hs_8u32f": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i,
                                               ctypes.c_float, _i, _i, ctypes.c_double]),
}

_lib = None


def load():
    """Returns the loaded library; raises ImportError if it cannot be loaded (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (same soname as
    # /opt/rocm's).  Loading torch first makes libhsflow.so bind to that copy; the other order
    # would put two runtimes in the process and the second one finds no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "opticalflowhs_amd: %s is missing -- build it with `python opticalflowhs_amd/build.py` "
            "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 not found
        raise ImportError("opticalflowhs_amd: cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ImportError("opticalflowhs_amd: %s does not export %s" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class HsflowError(RuntimeError):
    def __init__(self, status, message):
        RuntimeError.__init__(self, "hsflow status %d: %s" % (status, message))
        self.status = status
