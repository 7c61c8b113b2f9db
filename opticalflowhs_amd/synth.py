"""Deterministic synthetic inputs for tests and bench.py.

"Translating texture": T(x,y) = 128 + sum_{k=1..12} A_k sin(2 pi (fx_k x + fy_k y) + phi_k), with
fx_k, fy_k ~ U(-0.08, 0.08) cycles/pixel, A_k ~ U(4, 10), phi_k ~ U(0, 2 pi) drawn from
SplitMix64(seed).  Frame A samples T(x, y), frame B samples T(x - dx, y - dy) analytically (a pure
translation by (dx, dy) = (0.75, -0.5) pixels), both rounded and clamped to u8.
"""
import numpy as np

_MASK = (1 << 64) - 1


class SplitMix64(object):
    def __init__(self, seed):
        self.s = seed & _MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def uniform(self, lo, hi):
        return lo + (hi - lo) * ((self.next() >> 11) * (1.0 / (1 << 53)))


def texture_params(seed, n=12):
    r = SplitMix64(seed)
    return [(r.uniform(-0.08, 0.08), r.uniform(-0.08, 0.08), r.uniform(4.0, 10.0),
             r.uniform(0.0, 2.0 * np.pi)) for _ in range(n)]


def _sample(params, xs, ys):
    t = np.full(np.broadcast(xs, ys).shape, 128.0)
    for fx, fy, amp, phi in params:
        t += amp * np.sin(2.0 * np.pi * (fx * xs + fy * ys) + phi)
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)


def translating_pair(width, height, seed=1, dx=0.75, dy=-0.5, row0=0, rows=None):
    """Returns (A, B) u8 frames of shape (rows or height, width); rows [row0, row0+rows) only."""
    rows = height - row0 if rows is None else rows
    params = texture_params(seed)
    x = np.arange(width, dtype=np.float64)[None, :]
    y = np.arange(row0, row0 + rows, dtype=np.float64)[:, None]
    return _sample(params, x, y), _sample(params, x - dx, y - dy)


def random_pair(width, height, seed=0):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (height, width), dtype=np.uint8),
            rng.integers(0, 256, (height, width), dtype=np.uint8))


def smooth_random_pair(width, height, seed=0, shift=(1, 0)):
    """A low-pass random texture and a copy shifted by whole pixels (replicate at the border)."""
    rng = np.random.default_rng(seed)
    base = rng.random((height + 16, width + 16))
    k = np.ones(7) / 7.0
    for ax in (0, 1):
        base = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), ax, base)
    base = (base - base.min()) / max(base.max() - base.min(), 1e-12) * 255.0
    a = base[8:8 + height, 8:8 + width]
    b = base[8 - shift[1]:8 - shift[1] + height, 8 - shift[0]:8 - shift[0] + width]
    return np.rint(a).astype(np.uint8), np.rint(b).astype(np.uint8)
