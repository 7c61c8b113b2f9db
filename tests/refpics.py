"""The reference's own output pictures as the parity check of both routes, end to end.

CPU route, OpticalFlowOpenCV::runFromImg (OpticalFlowOpenCV.cpp:7-52): both frames to gray, 3x3
blur, cvCalcOpticalFlowHS(ITER|EPS, it, 1e-6); then on a black image, for every y and x that are
multiples of 4: if u > 1 or v > 1 or u < -1 or v < -1, a filled blue circle of radius 2 and a red
line to cvPoint(x + u/2, y + v/2) (:33-46); saved as JPEG.
OpenCL route, HSOpticalFlowOpenCL::run (HSOpticalFlowOpenCL.cpp:712-772): both frames to gray (no
blur), Kernels.cl for `iterations` sweeps -- as shipped, i.e. v is never written (Kernels.cl:84-86)
-- then the same drawing with threshold 0.5 and a full-length line to cvPoint(x + u, y + v).

The four pictures the reference wrote from the data it ships (OpticalFlowHS/city_cv_out.jpg,
bunny_cv_out.jpg, city_cl_out.jpg, bunny_cl_out.jpg) are the only outputs of the path it holds;
they are kept as data under tests/golden/ref_*.jpg.  Their run parameters were not recorded.  A
scan with the oracle has ONE sharp optimum per route, the same for both image pairs:
lambda = 0.1 (main.cpp:8's LAMBDA) / alpha = 15 (main.cpp:5's ALPHA), and 10 iterations.  There,
re-drawing the oracle's flow by the rule above (circle and line rasterised as OpenCV's cvCircle /
cvLine do) and saving it as JPEG (quality 95, 4:2:0 -- cvSaveImage's defaults) decodes to the
reference's picture EXACTLY: no pixel differs, in any of the four -- in fact the encoded FILES are the
same bytes (tests/test_jpeg.py, tests/test_gpu_frontend.py).  One sweep more or less, or
another lambda / alpha, changes dozens of dots and lines (PSNR drops from identical to < 37 dB).

What that pins: 24 360 (CPU route) + 24 360 (OpenCL route) drawn / not-drawn decisions and the
integer end point of every drawn line, i.e. trunc(x + u/2), trunc(y + v/2) -- the pre-processing,
the derivative scaling, the meaning of lambda / alpha, the sweep count, the update itself and the
signs of u and v.  The tightest sampled points sit 1e-4 .. 6e-4 from a decision boundary, so the
oracle agrees with OpenCV 2.1's cvCalcOpticalFlowHS (as run by the reference's authors) to better
than 1e-3 there; it is not an fp32-level vector.  Neighbouring schemes are told apart
(tests/variants.py, test_what_the_pictures_discriminate): Gauss-Seidel instead of Jacobi ordering,
Sobel on frame B or on both frames, central differences, the 8-neighbour mean, lambda for 1/lambda
each break > 5 000 pixels; zero or mirrored borders for the Sobel stencil or the blur break the
bunny picture.  NOT told apart: the border rule of the mean (zero padding draws the same
pictures after 10 sweeps) -- replicate there rests on the disassembly read alone.
(Release/bunny_cl_out.jpg, a second OpenCL-route picture of the bunny pair, matches no scanned
parameter pair of the shipped Kernels.cl -- best 27 dB -- and is not used.)
"""
import io
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LAMBDA, ALPHA, ITERATIONS, EPSILON = 0.1, 15.0, 10, float(np.float32(1e-6))
STEP = 4
# Codec-free detector for "a dot was drawn here", calibrated on a picture drawn by `render` from a
# known flow and passed through JPEG: blue + red level around a drawn point >= 189, around an empty
# one <= 112 except where several long lines cross (up to 218 in the bunny's fast region).
DRAWN_ABOVE, EMPTY_BELOW = 130, 80


def read_pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = [int(t) for t in f.readline().split()]
        assert int(f.readline()) == 255
        return np.frombuffer(f.read(w * h), dtype=np.uint8).reshape(h, w).copy()


def gray_pair(name):
    return (read_pgm(os.path.join(GOLDEN, name + "_1_gray.pgm")), read_pgm(os.path.join(GOLDEN, name + "_2_gray.pgm")))


def reference_decisions(name):
    """(drawn, clear): per grid point of the CPU-route picture, whether the reference drew a dot,
    and whether that is clear (from tests/golden/ref_cv_out_grids.npz; no JPEG codec needed)."""
    g = np.load(os.path.join(GOLDEN, "ref_cv_out_grids.npz"))
    level = g[name + "_blue"].astype(np.int32) + g[name + "_red"].astype(np.int32)
    drawn = level >= DRAWN_ABOVE
    clear = drawn | (level < EMPTY_BELOW)
    return drawn, clear


def our_decisions(u, v, threshold=1.0):
    """The drawing condition of OpticalFlowOpenCV.cpp:40 on the 4-pixel grid."""
    uu, vv = u[::STEP, ::STEP], v[::STEP, ::STEP]
    return (uu > threshold) | (vv > threshold) | (uu < -threshold) | (vv < -threshold)


def compare_decisions(u, v, name):
    """(disagreements on clear points, number of clear points)."""
    drawn, clear = reference_decisions(name)
    ours = our_decisions(u, v)
    assert ours.shape == drawn.shape
    return int(((ours != drawn) & clear).sum()), int(clear.sum())


def cv_line(img, x0, y0, x1, y1, colour):
    """8-connected line as OpenCV 2.1's cvLine(thickness 1) rasterises it (its LineIterator): start
    at the LEFT end point, error term major - 2*minor, diagonal step while the error is negative."""
    H, W = img.shape[:2]
    dx, dy = x1 - x0, y1 - y0
    if dx < 0:
        x0, y0, dx, dy = x1, y1, -dx, -dy
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    steep = dy > dx
    major, minor = (dy, dx) if steep else (dx, dy)
    err = major - 2 * minor
    x, y = x0, y0
    for _ in range(major + 1):
        if 0 <= x < W and 0 <= y < H:
            img[y, x] = colour
        if err < 0:
            err += 2 * major - 2 * minor
            x += 1
            y += sy
        else:
            err -= 2 * minor
            if steep:
                y += sy
            else:
                x += 1


def render(u, v, route="cv"):
    """RGB picture of a flow: OpticalFlowOpenCV.cpp:33-46 (route "cv": threshold 1, half-length
    lines) or HSOpticalFlowOpenCL.cpp:759-769 (route "cl": threshold 0.5, full-length lines).  The
    dot is the 13 pixels with dx^2 + dy^2 <= 4; end points truncate like cvPoint(float, float)."""
    threshold, scale = (1.0, 0.5) if route == "cv" else (0.5, 1.0)
    H, W = u.shape
    img = np.zeros((H, W, 3), np.uint8)
    disc = [(dx, dy) for dy in range(-2, 3) for dx in range(-2, 3) if dx * dx + dy * dy <= 4]
    for y in range(0, H, STEP):
        for x in range(0, W, STEP):
            a, b = np.float32(u[y, x]), np.float32(v[y, x])
            if not (a > threshold or b > threshold or a < -threshold or b < -threshold):
                continue
            for dx, dy in disc:
                if 0 <= x + dx < W and 0 <= y + dy < H:
                    img[y + dy, x + dx] = (0, 0, 255)
            # int + float is evaluated in fp32, then truncated toward zero by the int conversion
            cv_line(img, x, y, int(np.float32(x) + a * np.float32(scale)), int(np.float32(y) + b * np.float32(scale)), (255, 0, 0))
    return img


def reference_picture(name, route="cv"):
    from PIL import Image
    return np.asarray(Image.open(os.path.join(GOLDEN, "ref_%s_%s_out.jpg" % (name, route))).convert("RGB"))


def through_jpeg(img):
    """What cvSaveImage(".jpg") does to the drawing: quality 95, 4:2:0 chroma."""
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG", quality=95, subsampling=2)
    return np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10.0 * np.log10(255.0 ** 2 / max(mse, 1e-12))


def picture_difference(drawing, name, route="cv"):
    """(pixels differing by more than 4 levels, PSNR) between a drawing saved as JPEG and the
    reference's picture.  A misplaced dot or line pixel differs by > 40 levels; the margin of 4 only
    absorbs a JPEG codec whose DCT rounds differently from the one in this image (where it is 0)."""
    ours, ref = through_jpeg(drawing), reference_picture(name, route)
    assert ours.shape == ref.shape
    d = np.abs(ours.astype(np.int32) - ref.astype(np.int32)).max(axis=2)
    return int((d > 4).sum()), psnr(ours, ref)


def decision_margins(u, v, route="cv"):
    """Distance of every sampled value from the nearest place where the drawing would change: the
    threshold (all grid points) and the next integer of the line's end point (drawn points)."""
    threshold, scale = (1.0, 0.5) if route == "cv" else (0.5, 1.0)
    uu, vv = u[::STEP, ::STEP].astype(np.float64), v[::STEP, ::STEP].astype(np.float64)
    m = np.minimum(np.abs(np.abs(uu) - threshold), np.abs(np.abs(vv) - threshold))
    drawn = our_decisions(u, v, threshold)
    ys, xs = np.meshgrid(np.arange(0, u.shape[0], STEP), np.arange(0, u.shape[1], STEP), indexing="ij")
    for base, comp in ((xs, uu), (ys, vv)):
        frac = (base + comp * scale) % 1.0
        m = np.where(drawn, np.minimum(m, np.minimum(frac, 1.0 - frac) / scale), m)
    return m
