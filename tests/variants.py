"""Plausible-but-different discretisations of Horn-Schunck, for showing what the reference's output
pictures discriminate (tests/test_reference_pictures.py).  NumPy, fp32 stores, written as small
deviations from the normative scheme of SURVEY.md 8c; none of this is used by the product or the oracle."""
import numpy as np


def _sobel8(A, pad="edge"):
    a = np.pad(A.astype(np.int32), 1, mode=pad)
    sv = a[:-2, :] + 2 * a[1:-1, :] + a[2:, :]
    sh = a[:, :-2] + 2 * a[:, 1:-1] + a[:, 2:]
    ix = (sv[:, 2:] - sv[:, :-2]).astype(np.float32) * np.float32(0.125)
    iy = (sh[2:, :] - sh[:-2, :]).astype(np.float32) * np.float32(0.125)
    return ix, iy


def _central(A):
    a = np.pad(A.astype(np.float32), 1, mode="edge")
    return (a[1:-1, 2:] - a[1:-1, :-2]) * np.float32(0.5), (a[2:, 1:-1] - a[:-2, 1:-1]) * np.float32(0.5)


def _mean(p, mask, border):
    q = np.pad(p.astype(np.float64), 1, mode="edge" if border == "replicate" else "constant")
    if mask == 4:
        s = ((q[1:-1, :-2] + q[1:-1, 2:]) + q[:-2, 1:-1]) + q[2:, 1:-1]
        return (s * 0.25).astype(np.float32)
    s4 = q[1:-1, :-2] + q[1:-1, 2:] + q[:-2, 1:-1] + q[2:, 1:-1]
    s8 = q[:-2, :-2] + q[:-2, 2:] + q[2:, :-2] + q[2:, 2:]
    return (s4 / 6.0 + s8 / 12.0).astype(np.float32)


def box_blur3(img, pad="edge"):
    """3x3 mean, round half to even; pad = edge (cvSmooth's replicate, normative) | reflect | constant."""
    a = np.pad(img.astype(np.int32), 1, mode=pad)
    H, W = img.shape
    s = sum(a[1 + dy:1 + dy + H, 1 + dx:1 + dx + W] for dy in (-1, 0, 1) for dx in (-1, 0, 1))
    return np.asarray(np.rint(s / 9.0), dtype=np.uint8)


def flow(A, B, lam, iters, derivative="sobelA", mask=4, border="replicate", order="jacobi", regulariser="inverse",
         derivative_pad="edge"):
    """derivative: sobelA (normative) | sobelB | sobelAB (mean of both frames) | central (on A);
    mask: 4 (normative) | 8 (1/6, 1/12);  border of the mean: replicate (normative) | zero;
    order: jacobi (normative) | rows (Gauss-Seidel over rows: a row sees the new row above);
    regulariser: inverse (1/lambda, normative) | direct (lambda)."""
    if derivative == "sobelA":
        ix, iy = _sobel8(A, derivative_pad)  # edge (replicate, normative) | constant (zero) | reflect
    elif derivative == "sobelB":
        ix, iy = _sobel8(B)
    elif derivative == "sobelAB":
        xa, ya = _sobel8(A)
        xb, yb = _sobel8(B)
        ix, iy = (xa + xb) * np.float32(0.5), (ya + yb) * np.float32(0.5)
    else:
        ix, iy = _central(A)
    it = (B.astype(np.int32) - A.astype(np.int32)).astype(np.float32)
    reg = np.float32(1.0) / np.float32(lam) if regulariser == "inverse" else np.float32(lam)
    alpha = (1.0 / (ix.astype(np.float64) ** 2 + iy.astype(np.float64) ** 2 + np.float64(reg))).astype(np.float32).astype(np.float64)
    ix, iy, it = ix.astype(np.float64), iy.astype(np.float64), it.astype(np.float64)
    H, W = A.shape
    u = np.zeros((H, W), np.float32)
    v = np.zeros((H, W), np.float32)
    for _ in range(iters):
        if order == "jacobi":
            ax, ay = _mean(u, mask, border).astype(np.float64), _mean(v, mask, border).astype(np.float64)
            t = (ix * ax + iy * ay + it) * alpha
            u, v = (ax - ix * t).astype(np.float32), (ay - iy * t).astype(np.float32)
        else:
            for y in range(H):  # in place, row by row
                ya, yb = max(y - 1, 0), min(y + 1, H - 1)
                def mean_row(p):
                    r = np.pad(p[y].astype(np.float64), 1, mode="edge")
                    return ((((r[:-2] + r[2:]) + p[ya].astype(np.float64)) + p[yb].astype(np.float64)) * 0.25).astype(np.float32).astype(np.float64)
                ax, ay = mean_row(u), mean_row(v)
                t = (ix[y] * ax + iy[y] * ay + it[y]) * alpha[y]
                u[y], v[y] = (ax - ix[y] * t).astype(np.float32), (ay - iy[y] * t).astype(np.float32)
    return u, v
