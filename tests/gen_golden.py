"""Generates tests/golden/* (run once in the build container; outputs are committed).

The reference holds NO golden vectors for this path, so these fixtures pin
the oracle against itself over time (regressions in oracle/ or in the fixtures' inputs show up as
diffs) and give the GPU tests fixed inputs.  They are produced by the C restatement
(oracle/hs_cv_oracle.c) and every one is cross-checked here against the independent NumPy
restatement (oracle/hs_numpy.py) bit for bit before it is written.

The bunny frames come from the reference's data files OpticalFlowHS/bunny_1.jpg / bunny_2.jpg
(424x240), decoded with PIL in this container, converted with the BGR2GRAY fixed-point formula
(oracle/hs_preproc_oracle.c); JPEG decoding is not bit-identical to OpenCV 2.1's libjpeg, which
is fine because parity is defined on identical u8 inputs to both solvers.

The only outputs of the path that the reference holds are pictures: its CPU route
(OpticalFlowOpenCV.cpp:31-47) wrote OpticalFlowHS/city_cv_out.jpg and bunny_cv_out.jpg, its OpenCL
route (HSOpticalFlowOpenCL.cpp:755-772) city_cl_out.jpg and bunny_cl_out.jpg -- a blue dot + red
line at every 4th pixel in x and y where the flow exceeds a threshold.  Those four data files are
copied to tests/golden/ref_*.jpg (31-44 KB each); tests/refpics.py re-draws the oracle's / the HIP
path's flow by the same rule and compares the pictures (tests/test_reference_pictures.py,
tests/test_gpu_frontend.py).  `reference_picture_grid` additionally reduces the two CPU-route
pictures to the mean blue / red level around every grid point, for a check that needs no JPEG codec.
"""
import hashlib
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hs_numpy, hs_oracle  # noqa: E402
from opticalflowhs_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference/OpticalFlowHS"
ITER, EPS = 1, 2


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, dtype=np.uint8).tobytes())


def reference_picture_grid(rgb, step=4):
    """3x3 mean of the blue and red channel around every grid point (y, x multiples of `step`)."""
    H, W = rgb.shape[:2]
    out = []
    for ch in (2, 0):
        pad = np.pad(rgb[:, :, ch].astype(np.int32), 1, mode="edge")
        acc = sum(pad[1 + dy:1 + dy + H, 1 + dx:1 + dx + W][::step, ::step] for dy in (-1, 0, 1) for dx in (-1, 0, 1))
        out.append(((acc + 4) // 9).astype(np.uint8))
    return out


def both(A, B, lam, it, tt=ITER, eps=1e-6):
    u, v, n, e = hs_oracle.calc_optical_flow_hs(A, B, lam, it, eps, tt, return_info=True)
    u2, v2, n2, e2 = hs_numpy.calc_optical_flow_hs(A, B, lam, it, eps, tt, return_info=True)
    assert np.array_equal(u, u2) and np.array_equal(v, v2) and n == n2, "restatements disagree"
    return u, v, n, e


def main():
    os.makedirs(OUT, exist_ok=True)
    # K5: random u8 pairs
    k5 = {}
    for (W, H) in ((37, 29), (64, 48)):
        A, B = synth.random_pair(W, H, seed=W * 1000 + H)
        k5["A_%dx%d" % (W, H)] = A
        k5["B_%dx%d" % (W, H)] = B
        for lam in (0.01, 0.1, 1.0, 10.0):
            for it in (1, 2, 10, 100):
                u, v, _, _ = both(A, B, lam, it)
                k5["u_%dx%d_l%g_i%d" % (W, H, lam, it)] = u
                k5["v_%dx%d_l%g_i%d" % (W, H, lam, it)] = v
    np.savez_compressed(os.path.join(OUT, "k5_random.npz"), **k5)

    # translating texture, small
    A, B = synth.translating_pair(256, 128, seed=1)
    u, v, _, _ = both(A, B, 1.0, 100)
    np.savez_compressed(os.path.join(OUT, "synth_256x128_s1_l1_i100.npz"), A=A, B=B, u=u, v=v)

    # EPS termination inside the iteration budget
    A, B = synth.smooth_random_pair(48, 40, seed=7, shift=(1, 0))
    u, v, n, e = both(A, B, 0.002, 500, ITER | EPS, 1e-3)
    assert 3 < n < 500, n
    np.savez_compressed(os.path.join(OUT, "eps_48x40_l0.002_e1e-3.npz"), A=A, B=B, u=u, v=v,
                        iters=np.int32(n), eps=np.float32(e))

    # K6: bunny pair (BASELINE config C1): gray -> 3x3 blur -> HS(lambda=1, ITER|EPS, 50, 1e-6)
    if os.path.isdir(REF):
        from PIL import Image
        frames = []
        for name in ("bunny_1.jpg", "bunny_2.jpg"):
            rgb = np.asarray(Image.open(os.path.join(REF, name)).convert("RGB"))
            gray = hs_oracle.bgr2gray(rgb[:, :, ::-1])
            write_pgm(os.path.join(OUT, name.replace(".jpg", "_gray.pgm")), gray)
            frames.append(hs_oracle.box_blur3(gray))
        u, v, n, e = both(frames[0], frames[1], 1.0, 50, ITER | EPS, 1e-6)
        np.savez_compressed(os.path.join(OUT, "bunny_flow_l1_i50.npz"), u=u, v=v, iters=np.int32(n))
        # the second pair the reference ships, and what its CPU route drew for both pairs
        grids = {}
        # the reference's input pictures as they are (data files, 87-146 KB each): what its command line is run on
        for name in ("city_1.jpg", "city_2.jpg", "bunny_1.jpg", "bunny_2.jpg"):
            shutil.copyfile(os.path.join(REF, name), os.path.join(OUT, "ref_" + name))
        for name in ("city_1.jpg", "city_2.jpg"):
            rgb = np.asarray(Image.open(os.path.join(REF, name)).convert("RGB"))
            write_pgm(os.path.join(OUT, name.replace(".jpg", "_gray.pgm")), hs_oracle.bgr2gray(rgb[:, :, ::-1]))
        for name in ("city", "bunny"):
            # the pictures themselves (data files) for the full-resolution comparison
            for route in ("cv", "cl"):
                shutil.copyfile(os.path.join(REF, "%s_%s_out.jpg" % (name, route)), os.path.join(OUT, "ref_%s_%s_out.jpg" % (name, route)))
            rgb = np.asarray(Image.open(os.path.join(REF, name + "_cv_out.jpg")).convert("RGB"))
            grids[name + "_blue"], grids[name + "_red"] = reference_picture_grid(rgb)
        np.savez_compressed(os.path.join(OUT, "ref_cv_out_grids.npz"), **grids)
    else:
        print("reference not present: bunny fixtures left as they are")

    with open(os.path.join(OUT, "SHA256SUMS"), "w") as f:
        for name in sorted(os.listdir(OUT)):
            if name == "SHA256SUMS":
                continue
            with open(os.path.join(OUT, name), "rb") as g:
                f.write("%s  %s\n" % (hashlib.sha256(g.read()).hexdigest(), name))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
