"""Seeded synthetic rectified stereo pairs (no dataset exists offline).

left  = low-pass filtered uniform-random BGR texture (so that a true match exists and is unique),
right = left forward-warped by a piece-wise constant ground-truth disparity in [0, D),
        occlusion holes filled with fresh noise, plus uniform +-2 intensity noise.
"""
import numpy as np


def make_pair(H, W, D, seed=1234, block=48, noise=2):
    rng = np.random.default_rng(seed)
    tex = rng.integers(0, 256, size=(H + 2, W + 2, 3)).astype(np.float32)
    acc = np.zeros((H, W, 3), dtype=np.float32)
    for dy in range(3):
        for dx in range(3):
            acc += tex[dy:dy + H, dx:dx + W]
    # stretch contrast back after the 3x3 mean so that gray differences stay informative
    L = np.clip((acc / 9.0 - 128.0) * 2.5 + 128.0, 0, 255).astype(np.uint8)

    by, bx = (H + block - 1) // block, (W + block - 1) // block
    dmax = max(1, min(D, W // 2))
    gt_blocks = rng.integers(0, dmax, size=(by, bx))
    gt = np.repeat(np.repeat(gt_blocks, block, axis=0), block, axis=1)[:H, :W].astype(np.int32)

    R = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)  # occlusion filler
    ys, xs = np.mgrid[0:H, 0:W]
    # paint far-to-near so that nearer surfaces (larger d) overwrite farther ones
    order = np.argsort(gt, axis=None, kind="stable")
    yy, xx, dd = ys.ravel()[order], xs.ravel()[order], gt.ravel()[order]
    xr = xx - dd
    ok = xr >= 0
    R[yy[ok], xr[ok]] = L[yy[ok], xx[ok]]
    if noise:
        n = rng.integers(-noise, noise + 1, size=R.shape)
        R = np.clip(R.astype(np.int32) + n, 0, 255).astype(np.uint8)
    return L, R, gt


def shifted_pair(H, W, d0, seed=7):
    """R is L shifted by exactly d0 columns (K6): interior disparity must come out as d0."""
    rng = np.random.default_rng(seed)
    tex = rng.integers(0, 256, size=(H + 2, W + d0 + 2, 3)).astype(np.float32)
    acc = np.zeros((H, W + d0, 3), dtype=np.float32)
    for dy in range(3):
        for dx in range(3):
            acc += tex[dy:dy + H, dx:dx + W + d0]
    wide = np.clip((acc / 9.0 - 128.0) * 2.5 + 128.0, 0, 255).astype(np.uint8)
    L = np.ascontiguousarray(wide[:, :W])
    # R(x) = L(x + d0)  <=>  L(x) = R(x - d0)
    R = np.ascontiguousarray(wide[:, d0:d0 + W])
    return L, R
