"""Independent NumPy restatement of the CV-mode Horn-Schunck solver (second oracle).

TEST INFRASTRUCTURE ONLY.  Written separately from oracle/hs_cv_oracle.c (whole-array operations,
np.pad for the replicate border) so that the two restatements cross-check each other; both follow
SURVEY.md section 8c, i.e. cvCalcOpticalFlowHS as called at OpticalFlowHS/OpticalFlowOpenCV.cpp:29
(declaration OpenCV2.1/include/cv.h:481-483).  Parity status: see the header of hs_cv_oracle.c (pinned by the reference's output pictures at
drawing resolution, unpinned at the last fp32 bits).
"""
import numpy as np

TERMCRIT_ITER = 1
TERMCRIT_EPS = 2


def derivatives(A, B):
    """Sobel/8 of frame A with replicate border; It = B - A.  Exact in fp32."""
    a = np.pad(A.astype(np.int32), 1, mode="edge")
    sv = a[:-2, :] + 2 * a[1:-1, :] + a[2:, :]          # vertical [1 2 1], shape (H, W+2)
    sh = a[:, :-2] + 2 * a[:, 1:-1] + a[:, 2:]          # horizontal [1 2 1], shape (H+2, W)
    ix = (sv[:, 2:] - sv[:, :-2]).astype(np.float32) * np.float32(0.125)
    iy = (sh[2:, :] - sh[:-2, :]).astype(np.float32) * np.float32(0.125)
    it = (B.astype(np.int32) - A.astype(np.int32)).astype(np.float32)
    return ix, iy, it


def records(A, B, lam):
    ix, iy, it = derivatives(A, B)
    ilambda = np.float32(1.0) / np.float32(lam)
    xx, xy, yy, xt, yt = ix * ix, ix * iy, iy * iy, ix * it, iy * it   # all exact in fp32
    s = xx.astype(np.float64) + np.float64(ilambda)
    s = yy.astype(np.float64) + s
    alpha = (1.0 / s).astype(np.float32)
    return xx, xy, yy, xt, yt, alpha


def _mean4(p):
    """fl32(((L+R)+U+D) * 0.25) in float64, replicate border."""
    q = np.pad(p.astype(np.float64), 1, mode="edge")
    s = ((q[1:-1, :-2] + q[1:-1, 2:]) + q[:-2, 1:-1]) + q[2:, 1:-1]
    return (s * 0.25).astype(np.float32)


def calc_optical_flow_hs(A, B, lam, max_iter, epsilon=1e-6, term_type=TERMCRIT_ITER | TERMCRIT_EPS,
                         use_previous=False, velx=None, vely=None, return_info=False):
    xx, xy, yy, xt, yt, alpha = (r.astype(np.float64) for r in records(A, B, lam))
    H, W = A.shape
    if use_previous:
        u = np.array(velx, dtype=np.float32)
        v = np.array(vely, dtype=np.float32)
    else:
        u = np.zeros((H, W), np.float32)
        v = np.zeros((H, W), np.float32)
    eps_limit = float(np.float32(epsilon))
    it = 0
    eps = np.float32(0)
    while True:
        it += 1
        ax = _mean4(u).astype(np.float64)
        ay = _mean4(v).astype(np.float64)
        un = (ax - ((xy * ay + xx * ax) + xt) * alpha).astype(np.float32)
        vn = (ay - ((xy * ax + yy * ay) + yt) * alpha).astype(np.float32)
        if term_type & TERMCRIT_EPS:
            du = np.abs((u.astype(np.float64) - un.astype(np.float64)).astype(np.float32))
            dv = np.abs((v.astype(np.float64) - vn.astype(np.float64)).astype(np.float32))
            eps = np.float32(max(du.max(), dv.max()))
        u, v = un, vn
        if (term_type & TERMCRIT_ITER) and it == max_iter:
            break
        if (term_type & TERMCRIT_EPS) and float(eps) < eps_limit:
            break
    if return_info:
        return u, v, it, float(eps)
    return u, v
