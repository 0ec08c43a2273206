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
