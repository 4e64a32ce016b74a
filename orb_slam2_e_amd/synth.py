"""Seeded synthetic inputs for tests and bench.

numpy-only; identical on the build container and the GPU box (same numpy).
"""
import numpy as np


def synth_frame(k, w=640, h=480):
    """Frame k (seed 1000+k): mid-grey 128 + 300 filled rectangles (side 8-80,
    intensity U[0,255]) + 200 filled discs (radius 4-30) + noise U[-6,6], clamped."""
    rng = np.random.Generator(np.random.PCG64(1000 + k))
    img = np.full((h, w), 128, dtype=np.int16)
    for _ in range(300):
        sw, sh = rng.integers(8, 81, size=2)
        x0 = rng.integers(-sw // 2, w - sw // 2)
        y0 = rng.integers(-sh // 2, h - sh // 2)
        img[max(y0, 0):max(y0 + sh, 0), max(x0, 0):max(x0 + sw, 0)] = rng.integers(0, 256)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(200):
        r = int(rng.integers(4, 31))
        cx = int(rng.integers(0, w)); cy = int(rng.integers(0, h))
        v = int(rng.integers(0, 256))
        y0, y1 = max(cy - r, 0), min(cy + r + 1, h)
        x0, x1 = max(cx - r, 0), min(cx + r + 1, w)
        m = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
        img[y0:y1, x0:x1][m] = v
    img += rng.integers(-6, 7, size=(h, w), dtype=np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_frames(n, w=640, h=480, start=0):
    return np.stack([synth_frame(start + k, w, h) for k in range(n)])


def synth_descriptors(n=2000, seed=7, flip_p=0.08, n_replace=200):
    """Match input of config 2: A random; B = A with bits flipped w.p. flip_p,
    rows shuffled, n_replace rows replaced by fresh random bytes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    A = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    flips = rng.random((n, 256)) < flip_p
    B = A ^ np.packbits(flips, axis=1, bitorder="little")
    perm = rng.permutation(n)
    B = B[perm]
    rep = rng.choice(n, size=min(n_replace, n), replace=False)
    B[rep] = rng.integers(0, 256, size=(len(rep), 32), dtype=np.uint8)
    return A, np.ascontiguousarray(B), perm
