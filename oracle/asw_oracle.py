"""ctypes binding of the CPU parity oracle (oracle/asw_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under aswstereomatch_amd/ imports this module.
PARITY UNPINNED (see asw_oracle.c header): the reference ships no fixtures and needs OpenCV.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libasw_oracle.so")

DISPARITY_LEFT, DISPARITY_RIGHT = 0, 1
OK, ERR_SIZE_MISMATCH, ERR_EVEN_WINDOW, ERR_UNSUPPORTED_METHOD, ERR_UNSUPPORTED_LAYOUT = 0, 1, 2, 3, 4


def build(force=False):
    src = os.path.join(_HERE, "asw_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libasw_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_similarity_pixel.restype = C.c_float
        _lib.orc_similarity_pixel.argtypes = [C.c_int] * 3 + [C.c_float] * 3 + [C.c_double] * 3
        _lib.orc_wm_color_weight.restype = C.c_float
        _lib.orc_wm_color_weight.argtypes = [C.c_int] * 3 + [C.c_double]
        _lib.orc_wmedian_pick.restype = C.c_float
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.c_void_p)


def _out(shape, dtype):
    a = np.zeros(shape, dtype=dtype)
    return a, a.ctypes.data_as(C.c_void_p)


def set_threads(n):
    lib().orc_set_threads(int(n))


def max_threads():
    return int(lib().orc_get_max_threads())


def usable_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box shows every
    core of its host but grants a share of them; OpenMP teams larger than the share spin against each other), capped by
    what the OpenMP runtime offers."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, max_threads()))


def set_gray_bits(bits):
    """BGR2GRAY constant set of every gray conversion of the oracle: 14 (OpenCV 4.1.0, default) or 15 (later 4.x)."""
    lib().orc_set_gray_bits(int(bits))


def set_box_mode(mode):
    lib().orc_set_box_mode(int(mode))


def bgr2gray(img):
    img, p = _u8(img)
    H, W, _ = img.shape
    out, po = _out((H, W), np.uint8)
    lib().orc_bgr2gray(p, H, W, po)
    return out


def rgb2gray(img):
    img, p = _u8(img)
    H, W, _ = img.shape
    out, po = _out((H, W), np.uint8)
    lib().orc_rgb2gray(p, H, W, po)
    return out


def _hwc(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        return img, img.shape[0], img.shape[1], 1
    return img, img.shape[0], img.shape[1], img.shape[2]


def compute_ad(L, R, disp_type=0, minD=0, numD=30):
    L, H, W, Cn = _hwc(L)
    R = np.ascontiguousarray(R, dtype=np.uint8)
    if L.shape != R.shape:
        return ERR_SIZE_MISMATCH, None
    out, po = _out((numD, H, W), np.uint8)
    rc = lib().orc_compute_ad(L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), H, W, Cn, disp_type, minD, numD, po)
    return rc, out


def compute_tad(L, R, disp_type=0, T=30, minD=0, numD=30):
    L, H, W, Cn = _hwc(L)
    R = np.ascontiguousarray(R, dtype=np.uint8)
    if L.shape != R.shape:
        return ERR_SIZE_MISMATCH, None
    out, po = _out((numD, H, W), np.uint8)
    rc = lib().orc_compute_tad(L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), H, W, Cn, disp_type, T, minD, numD, po)
    return rc, out


def compute_sd(L, R, disp_type=0, minD=0, numD=30):
    L, H, W, Cn = _hwc(L)
    R = np.ascontiguousarray(R, dtype=np.uint8)
    if L.shape != R.shape:
        return ERR_SIZE_MISMATCH, None
    out, po = _out((numD, H, W), np.uint8)
    rc = lib().orc_compute_sd(L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), H, W, Cn, disp_type, minD, numD, po)
    return rc, out


def similarity_pixel(c, g, regularity=0.4, thresC=10.0, thresG=50.0):
    return float(lib().orc_similarity_pixel(int(c[0]), int(c[1]), int(c[2]), float(g[0]), float(g[1]), float(g[2]),
                                            regularity, thresC, thresG))


def compute_similarity(L, R, regularity=0.4, thresC=10.0, thresG=50.0, disp_type=0, minD=0, numD=30, win=None):
    L, H, W, Cn = _hwc(L)
    R = np.ascontiguousarray(R, dtype=np.uint8)
    if L.shape != R.shape:
        return ERR_SIZE_MISMATCH, None
    f = lib()
    if win is None:
        out, po = _out((numD, H, W), np.float32)
        rc = f.orc_compute_similarity(L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), H, W, Cn,
                                      C.c_double(regularity), C.c_double(thresC), C.c_double(thresG), disp_type, minD, numD, po)
    else:
        h = win // 2
        out, po = _out((numD, H + 2 * h, W + 2 * h), np.float32)
        rc = f.orc_compute_similarity_padded(L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), H, W, Cn,
                                             C.c_double(regularity), C.c_double(thresC), C.c_double(thresG), disp_type,
                                             win, minD, numD, po)
    return rc, out


def box_filter(src, k):
    src, p = _f32(src)
    out, po = _out(src.shape, np.float32)
    lib().orc_box_filter(p, po, src.shape[0], src.shape[1], k)
    return out


def cost_sad(L, R, disp_type=0, win=15, minD=0, numD=30):
    L, H, W, _ = _hwc(L)
    R, pr = _u8(R)
    out, po = _out((numD, H, W), np.float32)
    rc = lib().orc_cost_sad(L.ctypes.data_as(C.c_void_p), pr, H, W, disp_type, win, minD, numD, po)
    return rc, out


def wta(volume, minD=0):
    volume, p = _f32(volume)
    nD, H, W = volume.shape
    out, po = _out((H, W), np.float32)
    lib().orc_wta(p, nD, H, W, minD, po)
    return out


def lr_check(dl, dr, max_diff=1.0, invalid=-1.0):
    dl, pl = _f32(dl)
    dr, pr = _f32(dr)
    out, po = _out(dl.shape, np.float32)
    bad = lib().orc_lr_check(pl, pr, dl.shape[0], dl.shape[1], C.c_float(max_diff), C.c_float(invalid), po)
    return out, bad


def classic_taps(ks):
    nt = ks * ks - 1
    arrs = [np.zeros(nt, dtype=np.int32) for _ in range(4)]
    lib().orc_classic_taps(ks, *[a.ctypes.data_as(C.c_void_p) for a in arrs])
    return arrs  # dxw, dyw, dxs, dys


def _agg(fn, L, R, nvol, want_vol, *args):
    L, H, W, _ = _hwc(L)
    R, pr = _u8(R)
    disp, pd = _out((H, W), np.float32)
    vol, pv = (None, None)
    if want_vol:
        vol, pv = _out((nvol, H, W), np.float32)
    rc = fn(L.ctypes.data_as(C.c_void_p), pr, H, W, *args, pd, pv)
    return rc, disp, vol


def asw_classic(L, R, gamma_c=30.0, gamma_g=20.0, disp_type=0, win=15, minD=0, numD=64, want_vol=False, literal=False,
                rows=None):
    f = lib()
    if literal:
        return _agg(f.orc_asw_classic_literal, L, R, numD + 1, want_vol, C.c_double(gamma_c), C.c_double(gamma_g),
                    disp_type, win, minD, numD)
    if rows is not None:
        return _agg(f.orc_asw_classic_rows, L, R, numD + 1, want_vol, C.c_double(gamma_c), C.c_double(gamma_g),
                    disp_type, win, minD, numD, int(rows[0]), int(rows[1]))
    return _agg(f.orc_asw_classic, L, R, numD + 1, want_vol, C.c_double(gamma_c), C.c_double(gamma_g), disp_type, win,
                minD, numD)


def asw_direct8(L, R, disp_type=0, win=15, minD=0, numD=64, want_vol=False):
    return _agg(lib().orc_asw_direct8, L, R, numD + 1, want_vol, disp_type, win, minD, numD)


def asw_bilgrid(L, R, disp_type=0, sampleRateS=10.0, sampleRateR=10.0, minD=0, numD=64, want_vol=False):
    return _agg(lib().orc_asw_bilgrid, L, R, numD + 1, want_vol, disp_type, C.c_double(sampleRateS), C.c_double(sampleRateR), minD, numD)


def geodesic_dist(img, win=15, iters=3):
    img, H, W, _ = _hwc(img)
    out, po = _out((H, W, win, win), np.float32)
    rc = lib().orc_geodesic_dist(img.ctypes.data_as(C.c_void_p), H, W, win, iters, po)
    return rc, out


def asw_geodesic(L, R, disp_type=0, win=15, minD=0, numD=64, want_vol=False):
    return _agg(lib().orc_asw_geodesic, L, R, numD + 1, want_vol, disp_type, win, minD, numD)


def guided_filter(guide, P, r, eps):
    guide, H, W, Cn = _hwc(guide)
    P, pp = _f32(P)
    out, po = _out((H, W), np.float32)
    rc = lib().orc_guided_filter(guide.ctypes.data_as(C.c_void_p), Cn, pp, H, W, r, C.c_double(eps), po)
    return rc, out


def asw_guided2(L, R, disp_type=0, eps=1e-6, win=15, minD=0, numD=64, want_vol=False):
    return _agg(lib().orc_asw_guided2, L, R, numD, want_vol, disp_type, C.c_double(eps), win, minD, numD)


def asw_guided(L, R, disp_type=0, eps=1e-6, win=15, minD=0, numD=64, want_vol=False):
    return _agg(lib().orc_asw_guided, L, R, numD, want_vol, disp_type, C.c_double(eps), win, minD, numD)


def cost_ncc(L, R, disp_type=0, win=15, minD=0, numD=30, raw=False):
    L, H, W, _ = _hwc(L)
    R, pr = _u8(R)
    out, po = _out((numD, H, W), np.float32)
    rc = lib().orc_cost_ncc(L.ctypes.data_as(C.c_void_p), pr, H, W, disp_type, win, minD, numD, int(bool(raw)), po)
    return rc, out


def ncc_disparity(L, R, disp_type=0, win=15, minD=0, numD=30):
    L, H, W, _ = _hwc(L)
    R, pr = _u8(R)
    out, po = _out((H, W), np.float32)
    rc = lib().orc_ncc_disparity(L.ctypes.data_as(C.c_void_p), pr, H, W, disp_type, win, minD, numD, po)
    return rc, out


def asw_guided3(L, R, disp_type=0, eps=1e-6, win=15, minD=0, numD=64, want_vol=False):
    return _agg(lib().orc_asw_guided3, L, R, numD, want_vol, disp_type, C.c_double(eps), win, minD, numD)


def asw_blo1(L, R, disp_type=0, sampleRateR=0.015, win=15, minD=0, numD=64, want_vol=False):
    return _agg(lib().orc_asw_blo1, L, R, numD, want_vol, disp_type, C.c_double(sampleRateR), win, minD, numD)


def wm_color_weight(d0, d1, d2, rateR=10.0):
    return float(lib().orc_wm_color_weight(int(d0), int(d1), int(d2), C.c_double(rateR)))


def wm_space_kernel(win, rateS=10.0):
    out, po = _out((win, win), np.float32)
    lib().orc_wm_space_kernel(win, C.c_double(rateS), po)
    return out


def wmedian_pick(cost, weight):
    cost, pc = _f32(cost)
    weight, pw = _f32(weight)
    return float(lib().orc_wmedian_pick(pc, pw, cost.size))


def asw_wmedian(L, R, disp_type=0, win=15, rateS=10.0, rateR=10.0, minD=0, numD=64, want_vol=False):
    return _agg(lib().orc_asw_wmedian, L, R, numD, want_vol, disp_type, win, C.c_double(rateS), C.c_double(rateR), minD,
                numD)


def stereo_matching(L, R, disparity_type, algorithm, win=15, minD=0, numD=64):
    L, H, W, _ = _hwc(L)
    R, pr = _u8(R)
    disp, pd = _out((H, W), np.float32)
    rc = lib().orc_stereo_matching(L.ctypes.data_as(C.c_void_p), pr, H, W, disparity_type, algorithm, win, minD, numD, pd)
    return rc, disp


# ---- driver-side pre/post-processing ----
def resize_linear(img, dsize):
    """cv::resize(img, Size(w, h)) with INTER_LINEAR on 8UC3; dsize = (w, h)."""
    img, p = _u8(img)
    sh, sw, _ = img.shape
    dw, dh = dsize
    out, po = _out((dh, dw, 3), np.uint8)
    lib().orc_resize_linear_u8c3(p, sh, sw, po, dh, dw)
    return out


def bgr2hsv(img):
    img, p = _u8(img)
    out, po = _out(img.shape, np.uint8)
    lib().orc_bgr2hsv_u8(p, C.c_size_t(img.shape[0] * img.shape[1]), po)
    return out


def hsv2bgr(img):
    img, p = _u8(img)
    out, po = _out(img.shape, np.uint8)
    lib().orc_hsv2bgr_u8(p, C.c_size_t(img.shape[0] * img.shape[1]), po)
    return out


def bilateral_u8(plane, d=7, sigma_color=10.0, sigma_space=3.0):
    plane, p = _u8(plane)
    out, po = _out(plane.shape, np.uint8)
    lib().orc_bilateral_u8c1(p, plane.shape[0], plane.shape[1], d, C.c_double(sigma_color), C.c_double(sigma_space), po)
    return out


def bilateral_tables(d=7, sigma_color=10.0, sigma_space=3.0):
    dy = np.zeros(1024, np.int32); dx = np.zeros(1024, np.int32); w = np.zeros(1024, np.float32)
    n = lib().orc_bilateral_taps(d, C.c_double(sigma_space), dy.ctypes.data_as(C.c_void_p), dx.ctypes.data_as(C.c_void_p),
                                 w.ctypes.data_as(C.c_void_p))
    lut = np.zeros(256, np.float32)
    lib().orc_bilateral_color_lut(C.c_double(sigma_color), lut.ctypes.data_as(C.c_void_p))
    return dy[:n].copy(), dx[:n].copy(), w[:n].copy(), lut


def detail_boost(img):
    img, p = _u8(img)
    out, po = _out(img.shape, np.uint8)
    lib().orc_detail_boost(p, img.shape[0], img.shape[1], po)
    return out


def preprocess(img, dsize, boost=True):
    img, p = _u8(img)
    sh, sw, _ = img.shape
    dw, dh = dsize
    out, po = _out((dh, dw, 3), np.uint8)
    lib().orc_preprocess(p, sh, sw, dh, dw, int(bool(boost)), po)
    return out


def disparity_to_u8(disp, normalize=True):
    disp, p = _f32(disp)
    out, po = _out(disp.shape, np.uint8)
    lib().orc_disparity_to_u8(p, C.c_size_t(disp.size), int(bool(normalize)), po)
    return out
