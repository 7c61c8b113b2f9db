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
"""
import hashlib
import os
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
