"""Oracle vs the only outputs of the path that the reference holds: the four pictures its two routes
wrote (see tests/refpics.py).  CPU-only; the GPU twin is in tests/test_gpu_frontend.py."""
import numpy as np
import pytest

import refpics

ITER_EPS = 3
NAMES = ["city", "bunny"]


def cv_flow(oracle, A, B, lam=refpics.LAMBDA, it=refpics.ITERATIONS):
    return oracle.calc_optical_flow_hs(A, B, lam, it, epsilon=refpics.EPSILON, term_type=ITER_EPS)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_cpu_route_pictures(oracle, name):
    """cvCalcOpticalFlowHS as run by the reference's authors vs oracle/hs_cv_oracle.c."""
    pytest.importorskip("PIL")
    A0, B0 = refpics.gray_pair(name)
    A, B = oracle.box_blur3(A0), oracle.box_blur3(B0)
    u, v = cv_flow(oracle, A, B)
    wrong, quality = refpics.picture_difference(refpics.render(u, v), name)
    assert wrong == 0, (wrong, quality)            # every dot and every line pixel where the reference has it
    # how tight that is: sampled values close to a place where the drawing would have changed
    m = refpics.decision_margins(u, v)
    assert (m < 1e-3).sum() >= 5 and (m < 1e-2).sum() >= 50
    # and how specific: each of these loses dozens of dots / lines
    for label, (a, b, lam, it) in {"9 sweeps": (A, B, refpics.LAMBDA, 9), "11 sweeps": (A, B, refpics.LAMBDA, 11),
                                   "lambda 0.09": (A, B, 0.09, 10), "lambda 0.11": (A, B, 0.11, 10),
                                   "no blur": (A0, B0, refpics.LAMBDA, 10), "frames swapped": (B, A, refpics.LAMBDA, 10)}.items():
        u2, v2 = cv_flow(oracle, a, b, lam, it)
        wrong2, q2 = refpics.picture_difference(refpics.render(u2, v2), name)
        assert wrong2 > 200 and q2 < 40.0, (label, wrong2, q2)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_opencl_route_pictures(oracle, name):
    """Kernels.cl as shipped (v never written) vs oracle/hs_classic_oracle.c."""
    pytest.importorskip("PIL")
    A, B = refpics.gray_pair(name)
    u, v = oracle.classic_flow(A, B, refpics.ALPHA, refpics.ITERATIONS, update_v=False)
    assert not v.any()
    wrong, quality = refpics.picture_difference(refpics.render(u, v, "cl"), name, "cl")
    assert wrong == 0, (wrong, quality)
    for label, (alpha, it, upd) in {"9 sweeps": (refpics.ALPHA, 9, False), "11 sweeps": (refpics.ALPHA, 11, False),
                                    "alpha 14": (14.0, 10, False), "alpha 16": (16.0, 10, False),
                                    "v update restored": (refpics.ALPHA, 10, True)}.items():
        u2, v2 = oracle.classic_flow(A, B, alpha, it, update_v=upd)
        wrong2, q2 = refpics.picture_difference(refpics.render(u2, v2, "cl"), name, "cl")
        assert wrong2 > 200 and q2 < 40.0, (label, wrong2, q2)


def test_oracle_reproduces_the_fifth_picture_two_sweeps(oracle):
    """The reference holds a second OpenCL-route picture of the bunny pair, Release/bunny_cl_out.jpg (kept as
    tests/golden/ref_bunny_cl_out_release.jpg).  A scan (tools/scan_release_bunny.py, profiles/r02_release_bunny_scan.txt:
    42 228 parameter combinations) has one exact optimum: Kernels.cl as shipped, alpha 15 and TWO sweeps -- there the
    drawing saved as JPEG decodes to the picture with no pixel different.  A second, independent pin of the classic
    discretisation (derivative cube, 1/6-1/12 mean, alpha^2, the missing v update) at another sweep count."""
    pytest.importorskip("PIL")
    from PIL import Image
    import os
    ref = np.asarray(Image.open(os.path.join(refpics.GOLDEN, "ref_bunny_cl_out_release.jpg")).convert("RGB")).astype(np.int32)
    A, B = refpics.gray_pair("bunny")

    def wrong(alpha, it, upd):
        u, v = oracle.classic_flow(A, B, alpha, it, update_v=upd)
        ours = refpics.through_jpeg(refpics.render(u, v, "cl")).astype(np.int32)
        return int((np.abs(ours - ref).max(axis=2) > 4).sum())

    assert wrong(refpics.ALPHA, 2, False) == 0
    for alpha, it, upd in ((refpics.ALPHA, 1, False), (refpics.ALPHA, 3, False), (14.0, 2, False), (16.0, 2, False), (refpics.ALPHA, 2, True)):
        assert wrong(alpha, it, upd) > 200, (alpha, it, upd)


@pytest.mark.parametrize("name", NAMES)
def test_grid_decisions_without_a_jpeg_codec(oracle, name):
    """Same check from the committed blue / red levels only (no PIL): all clear decisions agree but
    the few points that long lines of neighbours overdraw."""
    A0, B0 = refpics.gray_pair(name)
    u, v = cv_flow(oracle, oracle.box_blur3(A0), oracle.box_blur3(B0))
    bad, clear = refpics.compare_decisions(u, v, name)
    assert clear > 0.99 * u[::4, ::4].size
    assert bad <= (1 if name == "city" else 8), (bad, clear)
    u9, v9 = cv_flow(oracle, oracle.box_blur3(A0), oracle.box_blur3(B0), it=9)
    assert refpics.compare_decisions(u9, v9, name)[0] >= bad + 15


def test_numpy_restatement_draws_the_same_picture(oracle):
    from oracle import hs_numpy
    A0, B0 = refpics.gray_pair("bunny")
    A, B = oracle.box_blur3(A0), oracle.box_blur3(B0)
    u, v = cv_flow(oracle, A, B)
    u2, v2 = hs_numpy.calc_optical_flow_hs(A, B, refpics.LAMBDA, refpics.ITERATIONS, refpics.EPSILON, ITER_EPS)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)


@pytest.mark.parametrize("name", NAMES)
def test_what_the_pictures_discriminate(oracle, name):
    """Plausible neighbouring discretisations (tests/variants.py) against the CPU-route pictures: the
    normative scheme is the oracle bit for bit and reproduces them; Gauss-Seidel ordering, derivatives
    taken on the other / both frames or by central differences, the 8-neighbour mean and lambda instead
    of 1/lambda each break thousands of pixels; the bunny picture also tells the replicate border of the
    Sobel stencil and of the blur from zero or mirrored borders.  What the pictures do NOT see is the
    border rule of the mean (zero padding draws the same picture after 10 sweeps): that one rests on the
    disassembly read."""
    pytest.importorskip("PIL")
    import variants
    A0, B0 = refpics.gray_pair(name)
    A, B = oracle.box_blur3(A0), oracle.box_blur3(B0)
    uo, vo = cv_flow(oracle, A, B)
    u, v = variants.flow(A, B, refpics.LAMBDA, refpics.ITERATIONS)
    assert np.array_equal(u, uo) and np.array_equal(v, vo)
    for kw in (dict(order="rows"), dict(derivative="sobelB"), dict(derivative="sobelAB"), dict(derivative="central"),
               dict(mask=8), dict(regulariser="direct")):
        u, v = variants.flow(A, B, refpics.LAMBDA, refpics.ITERATIONS, **kw)
        wrong, quality = refpics.picture_difference(refpics.render(u, v), name)
        assert wrong > 5000 and quality < 36.0, (kw, wrong, quality)
    u, v = variants.flow(A, B, refpics.LAMBDA, refpics.ITERATIONS, border="zero")
    assert refpics.picture_difference(refpics.render(u, v), name)[0] == 0   # not discriminated (see docstring)
    if name == "bunny":  # its flow reaches the frame border: the border rules of Sobel and of the blur show
        assert np.array_equal(variants.box_blur3(A0), A)
        for pad in ("constant", "reflect"):
            u, v = variants.flow(A, B, refpics.LAMBDA, refpics.ITERATIONS, derivative_pad=pad)
            assert refpics.picture_difference(refpics.render(u, v), name)[0] > 100, pad
            u, v = variants.flow(variants.box_blur3(A0, pad), variants.box_blur3(B0, pad), refpics.LAMBDA, refpics.ITERATIONS)
            assert refpics.picture_difference(refpics.render(u, v), name)[0] > 100, pad
