"""A second, independent statement of the OpenCV 4.1.0 primitives the reference's hot path calls, on whole
numpy arrays -- one function per cv:: call, so that the literal restatements in tests/test_oracle_second.py can be written
Mat operation by Mat operation, in the order aswMethods.cpp performs them.  The C oracle (oracle/asw_oracle.c) fuses the same
semantics per pixel; bit-for-bit agreement of the two is the known-answer test the unpinned oracle can have (SURVEY section 4:
"two independent restatements must agree").  TEST INFRASTRUCTURE ONLY.

Conventions: images are numpy arrays [H][W] or [H][W][C]; CV_8U = uint8, CV_32F = float32; every f32 operation is performed
in float32 with separate multiply and add (OpenCV's portable, non-fused paths, App. A-8); f64 accumulations are explicit."""
import numpy as np

F32 = np.float32
F64 = np.float64

BORDER_REFLECT = "symmetric"      # fedcba|abcdefgh|hgfedcb   (App. A-2)
BORDER_REFLECT_101 = "reflect"    # gfedcb|abcdefgh|gfedcba   (App. A-3, default of boxFilter / filter2D)


def copyMakeBorder(src, top, bottom, left, right, mode):
    pad = ((top, bottom), (left, right)) + (((0, 0),) if src.ndim == 3 else ())
    if src.shape[0] == 1 or src.shape[1] == 1:
        # numpy's 'reflect' on a length-1 axis is undefined; OpenCV returns the only element
        out = src
        if src.shape[0] == 1 and (top or bottom):
            out = np.repeat(out, top + bottom + 1, axis=0)
            pad = ((0, 0),) + pad[1:]
        if src.shape[1] == 1 and (left or right):
            out = np.repeat(out, left + right + 1, axis=1)
            pad = (pad[0], (0, 0)) + pad[2:]
        return np.pad(out, pad, mode=mode)
    return np.pad(src, pad, mode=mode)


def cvtColor_BGR2GRAY(bgr, bits=14):
    """cvtColor(COLOR_BGR2GRAY) on 8U (App. A-1): fixed point, round to nearest.  bits = 14: R2Y 4899, G2Y 9617, B2Y 1868;
    bits = 15: 9798, 19235, 3735 (the later 4.x constants)."""
    cb, cg, cr = (1868, 9617, 4899) if bits == 14 else (3735, 19235, 9798)
    a = bgr.astype(np.int64)
    return ((a[..., 0] * cb + a[..., 1] * cg + a[..., 2] * cr + (1 << (bits - 1))) >> bits).astype(np.uint8)


def absdiff(a, b):
    if a.dtype == np.uint8:
        return np.abs(a.astype(np.int16) - np.asarray(b).astype(np.int16)).astype(np.uint8)
    return np.abs(a - b)  # f32: exact sign drop of the f32 difference


def add_u8(a, b):
    """cv::add on 8U: saturating."""
    return np.minimum(a.astype(np.int32) + b.astype(np.int32), 255).astype(np.uint8)


def cvRound_f32(v):
    return np.rint(v)  # round half to even, like cvRound / lrintf


def saturate_u8(v):
    return np.clip(v, 0, 255).astype(np.uint8)


def addWeighted_u8(a, alpha, b, beta, gamma=0.0):
    """addWeighted on 8U: float arithmetic with float-cast scalars, cvRound, saturate_cast<uchar>."""
    t = a.astype(F32) * F32(alpha) + b.astype(F32) * F32(beta)
    if gamma:
        t = t + F32(gamma)
    return saturate_u8(cvRound_f32(t))


def addWeighted_f32(a, alpha, b, beta):
    return a * F32(alpha) + b * F32(beta)


def compare_gt(src, s):
    """compare(src, scalar, dst, CMP_GT): 255 where src > s (the scalar is a double)."""
    return np.where(src.astype(F64) > float(s), 255, 0).astype(np.uint8)


def scale_u8(a, s):
    """a * s assigned to an 8U Mat (MatExpr scale -> convertTo): float multiply, cvRound, saturate."""
    return saturate_u8(cvRound_f32(a.astype(F32) * F32(s)))


def mul_u8(a, b, scale=1.0):
    """Mat::mul on 8U with a scale: saturate(round(scale * a * b)) in float."""
    return saturate_u8(cvRound_f32(F32(scale) * a.astype(F32) * b.astype(F32)))


def filter2D_f32(src_u8, kernel):
    """filter2D(src 8U, CV_32F, 3x3 kernel): correlation, anchor at the centre, BORDER_REFLECT_101 (App. A-7)."""
    p = copyMakeBorder(src_u8, 1, 1, 1, 1, BORDER_REFLECT_101).astype(np.int64)
    H, W = src_u8.shape[:2]
    acc = np.zeros(src_u8.shape, np.int64)
    for j in range(3):
        for i in range(3):
            if kernel[j][i]:
                acc += int(kernel[j][i]) * p[j:j + H, i:i + W]
    return acc.astype(F32)  # |values| <= 16 * 255: exact in f32


def boxFilter(src, k, mode=0):
    """boxFilter(src, dst, CV_32F, Size(k, k)) normalised, anchor k/2, BORDER_REFLECT_101 (App. A-9), any channel count.
    mode 0: f64 window sum, horizontal sums first, both in ascending order; * 1/(k*k) in f64; one rounding to f32.
    mode 1: OpenCV's RowSum / ColumnSum sliding sums, literally."""
    if src.ndim == 3:
        return np.stack([boxFilter(src[..., c], k, mode) for c in range(src.shape[2])], axis=2)
    H, W = src.shape
    h = k // 2
    p = copyMakeBorder(src.astype(F64), h, k - 1 - h, h, k - 1 - h, BORDER_REFLECT_101)
    scale = 1.0 / (float(k) * float(k))
    if mode == 0:
        rows = np.zeros((H + k - 1, W), F64)
        for i in range(k):
            rows = rows + p[:, i:i + W]
        s = np.zeros((H, W), F64)
        for j in range(k):
            s = s + rows[j:j + H]
        return (s * scale).astype(F32)
    rows = np.zeros((H + k - 1, W), F64)  # RowSum: s = first k; then s += S[x + k] - S[x]
    s = np.zeros(H + k - 1, F64)
    for i in range(k):
        s = s + p[:, i]
    rows[:, 0] = s
    for x in range(W - 1):
        s = s + (p[:, x + k] - p[:, x])
        rows[:, x + 1] = s
    out = np.zeros((H, W), F32)  # ColumnSum: SUM = first k-1 rows; per row: s0 = SUM + Sp; D = s0*scale; SUM = s0 - Sm
    SUM = np.zeros(W, F64)
    for j in range(k - 1):
        SUM = SUM + rows[j]
    for y in range(H):
        s0 = SUM + rows[y + k - 1]
        out[y] = (s0 * scale).astype(F32)
        SUM = s0 - rows[y]
    return out


def normalize_minmax(src):
    """normalize(src, dst, 0, 1, NORM_MINMAX, CV_32F) (App. A-10): extrema over ALL channels (NaN skipped, as minMaxIdx's ordered
    comparisons skip them), scale / shift formed in double, applied in float."""
    a = src.astype(F64)
    fin = a[~np.isnan(a)]
    smin, smax = (float(fin.min()), float(fin.max())) if fin.size else (float("inf"), float("-inf"))
    scale = 1.0 / (smax - smin) if (smax - smin) > np.finfo(F64).eps else 0.0
    shift = 0.0 - smin * scale
    return src.astype(F32) * F32(scale) + F32(shift)


def cv_exp_f32(x):
    """cv::exp on 32F (App. A-12): restated as the correctly rounded single-precision exponential."""
    return np.exp(x.astype(F64)).astype(F32) if False else _expf(x)


def _expf(x):
    # libm expf through ctypes, element by element: numpy's float32 exp is a different (SIMD) implementation
    import ctypes
    import ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    flat = np.asarray(x, F32).ravel()
    return np.array([libm.expf(float(v)) for v in flat], F32).reshape(np.shape(x))


def sum_f64_rowmajor(a):
    """cv::sum of a 32F Mat: f64 accumulation, row-major."""
    s = 0.0
    for v in np.asarray(a, F32).ravel():
        s += float(v)
    return s


# ---------------------------------------------------------------------------------------------------------------------------
# f3 primitives (reference main.cpp:30-31, 67-89, 97-98), whole-array forms
# ---------------------------------------------------------------------------------------------------------------------------
def resize_linear_u8(src, dsize):
    """resize(src 8UC3, dst, Size(dw, dh)) with INTER_LINEAR: an exact 2x downscale runs as the 2x2 area average; otherwise
    11-bit coefficient pairs per destination column/row, horizontal pass into int, the truncating 8U vertical pass."""
    dw, dh = dsize
    sh, sw = src.shape[:2]
    s = src.astype(np.int64)
    if sw == 2 * dw and sh == 2 * dh:
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    scale_x, scale_y = 1.0 / (float(dw) / sw), 1.0 / (float(dh) / sh)

    def taps(n_dst, n_src, scale, clamp_frac):
        f = ((np.arange(n_dst, dtype=F64) + 0.5) * scale - 0.5).astype(F32)
        i0 = np.floor(f).astype(np.int64)
        f = (f - i0.astype(F32)).astype(F32)
        if clamp_frac:  # x only: the fraction is dropped where the pair would leave the row
            lo = i0 < 0
            f[lo] = 0
            i0[lo] = 0
            hi = i0 >= n_src - 1
            f[hi] = 0
            i0[hi] = n_src - 1
        c0 = np.rint((F32(1) - f) * F32(2048)).astype(np.int64)
        c1 = np.rint(f * F32(2048)).astype(np.int64)
        return i0, c0, c1

    x0, a0, a1 = taps(dw, sw, scale_x, True)
    x1 = np.minimum(x0 + 1, sw - 1)  # where x0 + 1 == sw the coefficient a1 is 0
    hor = s[:, x0] * a0[None, :, None] + s[:, x1] * a1[None, :, None]  # [sh][dw][3], S * 2048 where the pair is degenerate
    y0, b0, b1 = taps(dh, sh, scale_y, False)
    r0 = hor[np.clip(y0, 0, sh - 1)]
    r1 = hor[np.clip(y0 + 1, 0, sh - 1)]
    b0 = b0[:, None, None]
    b1 = b1[:, None, None]
    return ((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2).astype(np.uint8)


def cvtColor_BGR2HSV(bgr):
    """cvtColor(COLOR_BGR2HSV) on 8U, H in [0, 180): 12-bit reciprocal tables for S and H."""
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    v = np.maximum(np.maximum(b, g), r)
    diff = v - np.minimum(np.minimum(b, g), r)
    idx = np.arange(256, dtype=F64)
    with np.errstate(divide="ignore"):
        sdiv = np.where(idx > 0, np.rint((255 << 12) / idx), 0).astype(np.int64)
        hdiv = np.where(idx > 0, np.rint((180 << 12) / (6.0 * idx)), 0).astype(np.int64)
    s = (diff * sdiv[v] + 2048) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + 2048) >> 12
    h = np.where(h < 0, h + 180, h)
    return np.stack([saturate_u8(h), s.astype(np.uint8), v.astype(np.uint8)], axis=-1)


def cvtColor_HSV2BGR(hsv):
    """cvtColor(COLOR_HSV2BGR) on 8U: to float (H * 6/180, S/255, V/255), the six-sector table, * 255, cvRound."""
    h = hsv[..., 0].astype(F32) * (F32(6) / F32(180))
    s = hsv[..., 1].astype(F32) * (F32(1) / F32(255))
    v = hsv[..., 2].astype(F32) * (F32(1) / F32(255))
    h = np.where(h >= 6, h - F32(6), h).astype(F32)  # H <= 255 * 6/180 = 8.5: at most one wrap
    sec = np.floor(h).astype(np.int64)
    f = (h - sec.astype(F32)).astype(F32)
    one = F32(1)
    t0 = v
    t1 = v * (one - s)
    t2 = v * (one - s * f)
    t3 = v * (one - s * (one - f))
    tab = np.stack([t0, t1, t2, t3], axis=0)
    order = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    pick = order[sec]  # [...][3] -> which of tab for B, G, R
    out = np.take_along_axis(tab, np.moveaxis(pick, -1, 0), axis=0)  # [3][...]
    out = np.where(s[None] == 0, v[None], out)
    return np.moveaxis(saturate_u8(cvRound_f32(out * F32(255))), 0, -1)


def bilateralFilter_u8(src, d, sigma_color, sigma_space):
    """bilateralFilter(src 8UC1, dst, d, sigmaColor, sigmaSpace, BORDER_REFLECT) (aswStereoMatch.cpp:75,84): circular support
    of radius d/2 visited in raster order; per tap w = space[k] * colour[|I - I0|] in float; float sums in tap order;
    cvRound(sum / wsum)."""
    import math
    radius = d // 2 if d > 0 else int(round(sigma_space * 1.5))
    radius = max(radius, 1)
    gs = -0.5 / (sigma_space * sigma_space)
    gc = -0.5 / (sigma_color * sigma_color)
    cw = np.array([math.exp(i * i * gc) for i in range(256)], F64).astype(F32)
    H, W = src.shape
    p = copyMakeBorder(src, radius, radius, radius, radius, BORDER_REFLECT).astype(np.int64)
    c = src.astype(np.int64)
    acc = np.zeros((H, W), F32)
    wacc = np.zeros((H, W), F32)
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            r = math.sqrt(float(i * i + j * j))
            if r > radius:
                continue
            ws = F32(math.exp(r * r * gs))
            n = p[radius + i:radius + i + H, radius + j:radius + j + W]
            w = ws * cw[np.abs(n - c)]
            acc = acc + n.astype(F32) * w
            wacc = wacc + w
    return cvRound_f32(acc / wacc).astype(np.uint8)


def detail_boost(bgr):
    """aswStereoMatch.cpp:67-89 on one image: split HSV, V' = V + 2 * (V - bilateral(V)) in saturating u8 Mat arithmetic."""
    hsv = cvtColor_BGR2HSV(bgr)
    v = hsv[..., 2]
    blur = bilateralFilter_u8(v, 7, 10.0, 3.0)
    detail = np.clip(v.astype(np.int32) - blur.astype(np.int32), 0, 255)  # u8 subtract saturates at 0
    out = hsv.copy()
    out[..., 2] = np.clip(v.astype(np.int32) + 2 * detail, 0, 255).astype(np.uint8)
    return cvtColor_HSV2BGR(out)


def disparity_to_u8(disp, normalize=True):
    """aswStereoMatch.cpp:97-98: convertTo(CV_8UC1) then normalize(0, 255, NORM_MINMAX) on the u8 image."""
    u = saturate_u8(cvRound_f32(np.asarray(disp, F32)))
    if not normalize:
        return u
    mn, mx = int(u.min()), int(u.max())
    scale = 255.0 * (1.0 / (mx - mn) if (mx - mn) > np.finfo(F64).eps else 0.0)
    shift = 0.0 - mn * scale
    return saturate_u8(cvRound_f32(u.astype(F32) * F32(scale) + F32(shift)))
