"""CPU checks of the drop-in boundary: libhsflow.so loads and exports every symbol that
include/hsflow.h declares, struct layouts match, and argument errors are reported without a GPU.
No compute call is made here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "hsflow.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hsflow_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_tables_agree(hs):
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(hs._lib.PROTOTYPES) == names


def test_library_exports_every_declared_symbol(hs):
    lib = ctypes.CDLL(hs._lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_struct_sizes_and_defaults(hs):
    L = hs._lib.load()
    p = hs._lib.HsflowParams()
    L.hsflow_default_params(ctypes.byref(p))
    assert p.struct_size == ctypes.sizeof(hs._lib.HsflowParams)
    assert p.mode == hs.MODE_CV and p.term_type == (hs.TERM_ITER | hs.TERM_EPS)
    assert p.max_iter == 100 and abs(p.epsilon - 1e-6) < 1e-12 and p.lambda_ == 1.0
    assert L.hsflow_version() >= 1
    assert L.hsflow_status_string(0) == b"ok"
    assert L.hsflow_status_string(2) == b"invalid size or stride"


def test_argument_errors_without_gpu(hs):
    L = hs._lib.load()
    h = ctypes.c_void_p()
    assert L.hsflow_create(None, 0, 8, 8, 1, None, 1) == hs._lib.E_ARG
    assert L.hsflow_create(ctypes.byref(h), 0, 0, 8, 1, None, 1) == hs._lib.E_SIZE
    assert L.hsflow_create(ctypes.byref(h), 0, 8, -1, 1, None, 1) == hs._lib.E_SIZE
    assert L.hsflow_create(ctypes.byref(h), 0, 8, 8, 0, None, 1) == hs._lib.E_SIZE
    assert b"positive" in L.hsflow_last_error(None)
    assert L.hsflow_destroy(None) == 0
    assert L.hsflow_solve(None, None) == hs._lib.E_ARG
    assert L.hsflow_get_info(None, None) == hs._lib.E_ARG
    # one-shot entry: the original's pointer / size checks (cv210.dll VA 0x1012e089-0x1012e0c7)
    assert L.hsflow_calc_optical_flow_hs_8u32f(None, None, 8, 8, 8, 0, None, None, 32, 1.0, 1, 1, 0.0) == hs._lib.E_ARG
    buf = (ctypes.c_uint8 * 64)()
    vel = (ctypes.c_float * 64)()
    f = L.hsflow_calc_optical_flow_hs_8u32f
    assert f(buf, buf, 4, 8, 8, 0, vel, vel, 32, 1.0, 1, 1, 0.0) == hs._lib.E_SIZE   # width > img_step
    assert f(buf, buf, 8, 8, 8, 0, vel, vel, 30, 1.0, 1, 1, 0.0) == hs._lib.E_SIZE   # vel_step % 4
    assert f(buf, buf, 8, 8, 8, 0, vel, vel, 16, 1.0, 1, 1, 0.0) == hs._lib.E_SIZE   # vel_step < 4*width


def test_python_mirror_rejects_bad_types(hs):
    import numpy as np
    a = np.zeros((4, 4), np.uint8)
    f = np.zeros((4, 4), np.float32)
    with pytest.raises(TypeError):
        hs.calc_optical_flow_hs(a.astype(np.float32), a, 0, f, f, 1.0, hs.term_criteria(1, 1, 0))
    with pytest.raises(TypeError):
        hs.calc_optical_flow_hs(a, a, 0, f.astype(np.float64), f, 1.0, hs.term_criteria(1, 1, 0))
    with pytest.raises(ValueError):
        hs.calc_optical_flow_hs(a, a[:2], 0, f, f, 1.0, hs.term_criteria(1, 1, 0))
    c = hs.term_criteria(3, 50, 1e-6)
    assert c.epsilon == float(np.float32(1e-6)) and c.max_iter == 50


def test_no_oracle_in_product_package():
    """The product path must not import or call anything under oracle/."""
    pkg = os.path.join(ROOT, "opticalflowhs_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "hs_oracle" not in text and "libhs_oracle" not in text, fn
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn


def test_header_is_plain_c_and_a_c_host_links(hs, tmp_path):
    """include/hsflow.h must be usable from C (C99, no C++): a translation unit that takes the address
    of every declared entry point compiles with -Wall -Werror, links against libhsflow.so and runs the
    GPU-free calls."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    names = declared_functions()
    src = tmp_path / "host.c"
    src.write_text('#include "hsflow.h"\n#include <stdio.h>\n'
                   "typedef void (*fn)(void);\n"
                   "static fn table[] = {\n" + "".join("    (fn)%s,\n" % n for n in names) + "};\n"
                   "int main(void)\n{\n"
                   "    hsflow_params p;\n    hsflow_info i;\n    hsflow_ctx *c = 0;\n"
                   "    hsflow_default_params(&p);\n"
                   "    i.struct_size = sizeof i;\n"
                   "    if (p.struct_size != sizeof p || p.term_type != (HSFLOW_TERM_ITER | HSFLOW_TERM_EPS)) return 2;\n"
                   "    if (hsflow_create(&c, 0, 0, 8, 1, 0, 1) != HSFLOW_E_SIZE || c) return 3;\n"
                   "    if (hsflow_destroy(0) != HSFLOW_OK || hsflow_get_info(0, &i) != HSFLOW_E_ARG) return 4;\n"
                   '    printf("%d entry points, version %d, %s\\n", (int)(sizeof table / sizeof table[0]), hsflow_version(), hsflow_status_string(HSFLOW_E_NOTERM));\n'
                   "    return 0;\n}\n")
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(hs._lib.LIB_PATH)
    exe = str(tmp_path / "host")
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, str(src), "-o", exe, "-L", libdir, "-lhsflow",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert ("%d entry points" % len(names)) in r.stdout and "termination criteria never met" in r.stdout


def test_package_fails_loudly_without_the_hip_library(tmp_path):
    """No CPU fallback: importing the package when libhsflow.so is absent must raise, and nothing under
    opticalflowhs_amd/ may import or mention the oracle."""
    import subprocess
    import sys
    code = ("import os, sys\n"
            "os.environ['HSFLOW_LIB_PATH'] = %r\n"
            "sys.path.insert(0, %r)\n"
            "try:\n"
            "    import opticalflowhs_amd\n"
            "except ImportError as e:\n"
            "    print('IMPORT-ERROR', e)\n"
            "else:\n"
            "    print('IMPORTED')\n") % (str(tmp_path / "no_such_libhsflow.so"), ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "IMPORT-ERROR" in r.stdout and "no CPU fallback" in r.stdout.replace("There is no", "no"), (r.stdout, r.stderr[-500:])
    pkg = os.path.join(ROOT, "opticalflowhs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)
                assert "hs_oracle" not in text and "libhs_oracle" not in text, os.path.join(dirpath, f)
