"""aswstereomatch_amd -- MI355X-native adaptive-support-weight stereo matching.

Host-side mirror (Python) of the reference's interface for the hot path
(aswStereoMatch/methods/aswMethods.h, "M.h"): same function names, argument order, defaults
and enum values; numpy arrays stand where cv::Mat stands (as in OpenCV's own Python binding).
Everything is computed by hand-written HIP kernels behind the C-ABI of include/asw_mi355x.h;
there is no CPU fallback -- a missing libasw_mi355x.so raises ImportError at first use.

Error behaviour follows the reference: where the reference returns silently or
hands back an empty cv::Mat (size mismatch, even window) the mirror returns None / an empty
list and records the status in ``last_status()``; anything else raises AswError.
"""
import ctypes as C
import enum
import os
import threading

import numpy as np

from . import _lib
from ._lib import AswError, AswImage, AswTiming

__all__ = [
    "DisparityType", "StereoMatchingAlgorithms", "DISPARITY_LEFT", "DISPARITY_RIGHT", "Context",
    "stereoMatching", "computeAD", "computeTAD", "computeSimilarity", "getCostSAD", "getCostSAD_d", "computeAdaptiveWeight",
    "computeAdaptiveWeight_geodesic", "getGeodesicDist", "getGuidedFilter", "computeAdaptiveWeight_GuidedF",
    "computeAdaptiveWeight_GuidedF_2", "computeAdaptiveWeight_WeightedMedian", "winnerTakeAll", "last_status",
    "stereoMatchingBatch", "computeAdaptiveWeight_BLO1", "computeAdaptiveWeight_direct8", "computeNCC", "computeNCC_costs",
    "computeAdaptiveWeight_GuidedF_3",
    "AswError",
]


class DisparityType(enum.IntEnum):  # parametersStereo.h:4-8
    DISPARITY_LEFT = 0
    DISPARITY_RIGHT = 1


class StereoMatchingAlgorithms(enum.IntEnum):  # parametersStereo.h:10-24
    BM = 0
    SGBM = 1
    ADAPTIVE_WEIGHT = 2
    ADAPTIVE_WEIGHT_8DIRECT = 3
    ADAPTIVE_WEIGHT_GEODESIC = 4
    ADAPTIVE_WEIGHT_BILATERAL_GRID = 5
    ADAPTIVE_WEIGHT_BLO1 = 6
    ADAPTIVE_WEIGHT_GUIDED_FILTER = 7
    ADAPTIVE_WEIGHT_GUIDED_FILTER_2 = 8
    ADAPTIVE_WEIGHT_GUIDED_FILTER_3 = 9
    ADAPTIVE_WEIGHT_MEDIAN = 10
    NCC = 11


DISPARITY_LEFT = DisparityType.DISPARITY_LEFT
DISPARITY_RIGHT = DisparityType.DISPARITY_RIGHT

OK, ERR_SIZE_MISMATCH, ERR_EVEN_WINDOW, ERR_UNSUPPORTED_METHOD, ERR_UNSUPPORTED_LAYOUT = 0, 1, 2, 3, 4
ERR_HIP, ERR_ALLOC, ERR_BAD_ARGUMENT, ERR_NO_FRAME = 5, 6, 7, 8
# statuses for which the reference returns silently / an empty Mat
_SILENT = (ERR_SIZE_MISMATCH, ERR_EVEN_WINDOW)

_state = threading.local()


def last_status():
    """Status of the last call made from this thread (0 = ok)."""
    return getattr(_state, "status", 0)


def _image(arr, depth=0):
    """numpy array -> asw_image (keeps `arr` alive through the returned tuple)."""
    a = np.asarray(arr)
    want = np.uint8 if depth == 0 else np.float32
    if a.dtype != want:
        raise TypeError("expected %s image, got %s" % (np.dtype(want).name, a.dtype))
    if a.ndim == 2:
        ch = 1
    elif a.ndim == 3:
        ch = a.shape[2]
    else:
        raise ValueError("image must be HxW or HxWxC")
    if a.strides[-1] != a.itemsize or (a.ndim == 3 and a.strides[1] != ch * a.itemsize):
        a = np.ascontiguousarray(a)
    img = AswImage(a.ctypes.data, a.shape[0], a.shape[1], ch, depth, a.strides[0])
    return img, a


class Context:
    """One device context (asw_create / asw_destroy).  Not shared between threads."""

    def __init__(self, device_id=0, env=None):
        """env: measurement / test switches (ASW_BILATERAL_XQ, ASW_BAND_Q, ...) for THIS context: the library reads its
        switches once, inside asw_create, so they are set only around that call."""
        self._h = C.c_void_p()
        self._lib = _lib.lib()
        saved = {k: os.environ.get(k) for k in (env or {})}
        try:
            for k, v in (env or {}).items():
                os.environ[k] = str(v)
            rc = self._lib.asw_create(int(device_id), C.byref(self._h))
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if rc != 0:
            raise AswError(rc, "asw_create(device %d)" % device_id)
        self.device_id = device_id

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.asw_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_gray_bits(self, bits):
        """cvtColor(BGR2GRAY) constant set: 14 (OpenCV 4.1.0, default) or 15 (later 4.x) -- asw_set_gray_bits."""
        rc = self._lib.asw_set_gray_bits(self._h, int(bits))
        if rc != 0:
            raise AswError(rc, "asw_set_gray_bits")

    # ---- helpers ----
    def _finish(self, rc, where):
        _state.status = rc
        if rc == 0:
            return True
        if rc in _SILENT:
            return False
        raise AswError(rc, where)

    @staticmethod
    def _candidates(algorithm, numDisparity):
        n = _lib.lib().asw_volume_planes(int(algorithm), int(numDisparity))  # the library knows which ranges are inclusive
        return n if n > 0 else int(numDisparity)

    # ---- whole-method entry point (M.h:91-92) ----
    def stereoMatching(self, srcLeft, srcRight, disparityType, algorithmType, winSize=15, minDisparity=0,
                       numDisparity=64, return_cost_volume=False):
        li, la = _image(srcLeft)
        ri, ra = _image(srcRight)
        disp = np.zeros((la.shape[0], la.shape[1]), np.float32)
        di, _ = _image(disp, 5)
        vol = None
        pv = None
        if return_cost_volume:
            vol = np.zeros((self._candidates(int(algorithmType), numDisparity), la.shape[0], la.shape[1]), np.float32)
            pv = vol.ctypes.data_as(C.c_void_p)
        rc = self._lib.asw_stereo_match(self._h, C.byref(li), C.byref(ri), C.byref(di), int(disparityType),
                                        int(algorithmType), winSize, minDisparity, numDisparity, pv, 0 if vol is None else vol.size)
        if not self._finish(rc, "asw_stereo_match"):
            return (None, None) if return_cost_volume else None
        return (disp, vol) if return_cost_volume else disp

    def _aggregate(self, fn, name, n_vol, leftImg, rightImg, args, return_cost_volume):
        li, la = _image(leftImg)
        ri, ra = _image(rightImg)
        disp = np.zeros((la.shape[0], la.shape[1]), np.float32)
        di, _ = _image(disp, 5)
        vol, pv = None, None
        if return_cost_volume:
            vol = np.zeros((n_vol, la.shape[0], la.shape[1]), np.float32)
            pv = vol.ctypes.data_as(C.c_void_p)
        rc = fn(self._h, C.byref(li), C.byref(ri), C.byref(di), *args, pv, 0 if vol is None else vol.size)
        if not self._finish(rc, name):
            return (None, None) if return_cost_volume else None
        return (disp, vol) if return_cost_volume else disp

    # ---- per-method entry points (M.h:133-182) ----
    def computeAdaptiveWeight(self, leftImg, rightImg, gamma_c=30, gamma_g=2, dispType=DISPARITY_LEFT, winSize=7,
                              minDisparity=186, numDisparity=144, return_cost_volume=False):
        return self._aggregate(self._lib.asw_aggregate_bilateral, "asw_aggregate_bilateral", numDisparity + 1, leftImg,
                               rightImg, (float(gamma_c), float(gamma_g), int(dispType), winSize, minDisparity, numDisparity),
                               return_cost_volume)

    def computeAdaptiveWeight_direct8(self, leftImg, rightImg, dispType=DISPARITY_LEFT, winSize=7, minDisparity=186,
                                      numDisparity=144, return_cost_volume=False):
        """M.h:135-136, M.cpp:1167-1319 (DISPARITY_LEFT only: the reference's RIGHT branch is undefined behaviour)."""
        return self._aggregate(self._lib.asw_aggregate_direct8, "asw_aggregate_direct8", numDisparity + 1, leftImg,
                               rightImg, (int(dispType), winSize, minDisparity, numDisparity), return_cost_volume)

    def computeAdaptiveWeight_geodesic(self, leftImg, rightImg, dispType=DISPARITY_LEFT, winSize=7, minDisparity=186,
                                       numDisparity=144, return_cost_volume=False):
        return self._aggregate(self._lib.asw_aggregate_geodesic, "asw_aggregate_geodesic", numDisparity + 1, leftImg,
                               rightImg, (int(dispType), winSize, minDisparity, numDisparity), return_cost_volume)

    def computeAdaptiveWeight_GuidedF(self, leftImg, rightImg, dispType=DISPARITY_LEFT, eps=1e-8, winSize=35,
                                      minDisparity=186, numDisparity=144, return_cost_volume=False):
        return self._aggregate(self._lib.asw_aggregate_guided, "asw_aggregate_guided", numDisparity, leftImg, rightImg,
                               (int(dispType), float(eps), winSize, minDisparity, numDisparity), return_cost_volume)

    def computeAdaptiveWeight_GuidedF_2(self, leftImg, rightImg, dispType=DISPARITY_LEFT, eps=1e-8, winSize=35,
                                        minDisparity=186, numDisparity=144, return_cost_volume=False):
        return self._aggregate(self._lib.asw_aggregate_guided2, "asw_aggregate_guided2", numDisparity, leftImg, rightImg,
                               (int(dispType), float(eps), winSize, minDisparity, numDisparity), return_cost_volume)

    def computeAdaptiveWeight_GuidedF_3(self, leftImg, rightImg, dispType=DISPARITY_LEFT, eps=1e-6, winSize=35,
                                        minDisparity=186, numDisparity=144, return_cost_volume=False):
        """M.h:174-176, M.cpp:3063-3137: NCC costs + guided filter."""
        return self._aggregate(self._lib.asw_aggregate_guided3, "asw_aggregate_guided3", numDisparity, leftImg, rightImg,
                               (int(dispType), float(eps), winSize, minDisparity, numDisparity), return_cost_volume)

    def computeNCC(self, leftImg, rightImg, dispType=DISPARITY_LEFT, winSize=7, minDisparity=0, numDisparity=30):
        """computeNCC -> disparity (M.h:122-123, M.cpp:812-913)."""
        li, la = _image(leftImg)
        ri, ra = _image(rightImg)
        disp = np.zeros((la.shape[0], la.shape[1]), np.float32)
        di, _ = _image(disp, 5)
        rc = self._lib.asw_ncc_disparity(self._h, C.byref(li), C.byref(ri), C.byref(di), int(dispType), winSize, minDisparity,
                                         numDisparity)
        return disp if self._finish(rc, "asw_ncc_disparity") else None

    def computeNCC_costs(self, leftImg, rightImg, dispType=DISPARITY_LEFT, winSize=7, minDisparity=0, numDisparity=30,
                         normalized=True):
        """computeNCC -> cost_ds (M.h:124-126, M.cpp:924-1013); normalized=False returns the planes before normalize()."""
        a = np.asarray(leftImg)
        return self._cost(self._lib.asw_cost_ncc, "asw_cost_ncc", leftImg, rightImg, np.float32, a.shape[:2], numDisparity,
                          (int(dispType), winSize, minDisparity, numDisparity, int(bool(normalized))))

    def computeAdaptiveWeight_BLO1(self, leftImg, rightImg, dispType=DISPARITY_LEFT, sampleRateR=10, winSize=35,
                                   minDisparity=186, numDisparity=144, return_cost_volume=False):
        return self._aggregate(self._lib.asw_aggregate_blo1, "asw_aggregate_blo1", numDisparity, leftImg, rightImg,
                               (int(dispType), float(sampleRateR), winSize, minDisparity, numDisparity), return_cost_volume)

    def computeAdaptiveWeight_bilateralGrid(self, leftImg, rightImg, dispType=DISPARITY_LEFT, sampleRateS=10, sampleRateR=10,
                                            minDisparity=186, numDisparity=144, return_cost_volume=False):
        """M.h:155-157, M.cpp:2253-2430 (DISPARITY_LEFT only: the reference's RIGHT branch reads past the image row)."""
        return self._aggregate(self._lib.asw_aggregate_bilgrid, "asw_aggregate_bilgrid", numDisparity + 1, leftImg, rightImg,
                               (int(dispType), float(sampleRateS), float(sampleRateR), minDisparity, numDisparity),
                               return_cost_volume)

    def computeAdaptiveWeight_WeightedMedian(self, leftImg, rightImg, dispType=DISPARITY_LEFT, winSize=35,
                                             sampleRateS=10, sampleRateR=10, minDisparity=186, numDisparity=144,
                                             return_cost_volume=False):
        return self._aggregate(self._lib.asw_aggregate_wmedian, "asw_aggregate_wmedian", numDisparity, leftImg, rightImg,
                               (int(dispType), winSize, float(sampleRateS), float(sampleRateR), minDisparity, numDisparity),
                               return_cost_volume)

    # ---- cost builders (M.h:101-113): return a list of planes like std::vector<cv::Mat> ----
    def _cost(self, fn, name, leftImg, rightImg, dtype, plane_shape, numDisparity, args):
        li, la = _image(leftImg)
        ri, ra = _image(rightImg)
        out = np.zeros((numDisparity,) + plane_shape, dtype)
        rc = fn(self._h, C.byref(li), C.byref(ri), out.ctypes.data_as(C.c_void_p), *args)
        if not self._finish(rc, name):
            return []  # cost_ds left empty / untouched by the reference
        return list(out)

    def computeAD(self, leftImg, rightImg, dispType=DISPARITY_LEFT, minDisparity=0, numDisparity=30):
        a = np.asarray(leftImg)
        return self._cost(self._lib.asw_cost_ad, "asw_cost_ad", leftImg, rightImg, np.uint8, a.shape[:2], numDisparity,
                          (int(dispType), minDisparity, numDisparity))

    def computeTAD(self, leftImg, rightImg, dispType=DISPARITY_LEFT, threshold_T=30, minDisparity=0, numDisparity=30):
        a = np.asarray(leftImg)
        return self._cost(self._lib.asw_cost_tad, "asw_cost_tad", leftImg, rightImg, np.uint8, a.shape[:2], numDisparity,
                          (int(dispType), threshold_T, minDisparity, numDisparity))

    def computeSD(self, leftImg, rightImg, dispType=DISPARITY_LEFT, minDisparity=0, numDisparity=30):
        """Squared differences (M.cpp:670-759): the AD plane multiplied by itself in u8, saturating at 255."""
        a = np.asarray(leftImg)
        return self._cost(self._lib.asw_cost_sd, "asw_cost_sd", leftImg, rightImg, np.uint8, a.shape[:2], numDisparity,
                          (int(dispType), minDisparity, numDisparity))

    def computeSimilarity(self, leftImg, rightImg, regularity, thresC, thresG, dispType, minDisparity, numDisparity,
                          winSize=None):
        """Both overloads of computeSimilarity (M.cpp:415, 651); winSize selects the padded one."""
        a = np.asarray(leftImg)
        h = 0 if winSize is None else winSize // 2
        shape = (a.shape[0] + 2 * h, a.shape[1] + 2 * h)
        return self._cost(self._lib.asw_cost_similarity, "asw_cost_similarity", leftImg, rightImg, np.float32, shape,
                          numDisparity, (float(regularity), float(thresC), float(thresG), int(dispType),
                                         0 if winSize is None else winSize, minDisparity, numDisparity))

    def getCostSAD(self, leftImg, rightImg, dispType=DISPARITY_LEFT, winSize=35, minDisparity=0, numDisparity=30):
        """getCostSAD_d (M.cpp:2442) for every disparity, as M.cpp:2884-2889 calls it."""
        a = np.asarray(leftImg)
        return self._cost(self._lib.asw_cost_sad, "asw_cost_sad", leftImg, rightImg, np.float32, a.shape[:2], numDisparity,
                          (int(dispType), winSize, minDisparity, numDisparity))

    def getCostSAD_d(self, leftImg, rightImg, disparity, dispType=DISPARITY_LEFT, winSize=35):
        """getCostSAD_d as declared (M.h:156, M.cpp:2442-2503): one disparity, the non-reference view already bordered by
        the caller (M.cpp:2877-2878).  None where the reference returns Mat()."""
        li, la = _image(leftImg)
        ri, ra = _image(rightImg)
        ref = la if int(dispType) == 0 else ra
        out = np.zeros(ref.shape[:2], np.float32)
        rc = self._lib.asw_cost_sad_d(self._h, C.byref(li), C.byref(ri), out.ctypes.data_as(C.c_void_p), int(disparity),
                                      int(dispType), winSize)
        return out if self._finish(rc, "asw_cost_sad_d") else None

    # ---- public building blocks ----
    def getGuidedFilter(self, guidedImg, inputP, r, eps):
        gi, ga = _image(guidedImg)
        p = np.ascontiguousarray(inputP, dtype=np.float32)
        if p.shape != ga.shape[:2]:
            _state.status = ERR_SIZE_MISMATCH
            return None  # M.cpp:2768-2769
        q = np.zeros_like(p)
        rc = self._lib.asw_guided_filter(self._h, C.byref(gi), p.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p),
                                         r, float(eps))
        return q if self._finish(rc, "asw_guided_filter") else None

    def getGeodesicDist(self, originImg, winSize=15, iterTime=3):
        ii, ia = _image(originImg)
        out = np.zeros((ia.shape[0], ia.shape[1], winSize, winSize), np.float32)
        rc = self._lib.asw_geodesic_dist(self._h, C.byref(ii), out.ctypes.data_as(C.c_void_p), winSize, iterTime)
        return out if self._finish(rc, "asw_geodesic_dist") else None

    def winnerTakeAll(self, costVolume, minDisparity=0):
        v = np.ascontiguousarray(costVolume, dtype=np.float32)
        disp = np.zeros(v.shape[1:], np.float32)
        rc = self._lib.asw_wta(self._h, v.ctypes.data_as(C.c_void_p), v.shape[0], v.shape[1], v.shape[2], minDisparity,
                               disp.ctypes.data_as(C.c_void_p))
        self._finish(rc, "asw_wta")
        return disp

    def leftRightCheck(self, dispLeft, dispRight, maxDiff=1.0, invalid=-1.0):
        """asw_lr_check: (checked left disparity, number of rejected pixels)."""
        a = np.ascontiguousarray(dispLeft, dtype=np.float32)
        b = np.ascontiguousarray(dispRight, dtype=np.float32)
        if a.ndim != 2 or a.shape != b.shape:
            raise ValueError("leftRightCheck: two float32 maps of equal (H, W) shape expected")
        out = np.zeros(a.shape, np.float32)
        n = C.c_int(0)
        rc = self._lib.asw_lr_check(self._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1],
                                    float(maxDiff), float(invalid), out.ctypes.data_as(C.c_void_p), C.byref(n))
        self._finish(rc, "asw_lr_check")
        return out, n.value

    def bgr2gray(self, img):
        ii, ia = _image(img)
        out = np.zeros(ia.shape[:2], np.uint8)
        rc = self._lib.asw_bgr2gray(self._h, C.byref(ii), out.ctypes.data_as(C.c_void_p))
        self._finish(rc, "asw_bgr2gray")
        return out

    # ---- resident (HBM) API used by bench.py.  The reference has no equivalent whose silent returns could be mimicked:
    # every non-zero status raises AswError (a rejected upload or a failed match also empties the slot's results, so a
    # later download cannot hand back another frame's disparity) ----
    def _strict(self, rc, where):
        _state.status = rc
        if rc != 0:
            raise AswError(rc, where)

    def upload_pair(self, slot, left, right):
        li, la = _image(left)
        ri, ra = _image(right)
        self._strict(self._lib.asw_upload_pair(self._h, slot, C.byref(li), C.byref(ri)), "asw_upload_pair")

    def match_resident(self, slot, disparityType, algorithmType, winSize, minDisparity, numDisparity, keep_volume=False):
        rc = self._lib.asw_match_resident(self._h, slot, int(disparityType), int(algorithmType), winSize, minDisparity,
                                          numDisparity, 1 if keep_volume else 0)
        self._strict(rc, "asw_match_resident")

    def download_disparity(self, slot, shape):
        disp = np.zeros(shape, np.float32)
        di, _ = _image(disp, 5)
        self._strict(self._lib.asw_download_disparity(self._h, slot, C.byref(di)), "asw_download_disparity")
        return disp

    def download_volume(self, slot, shape):
        vol = np.zeros(shape, np.float32)
        self._strict(self._lib.asw_download_volume(self._h, slot, vol.ctypes.data_as(C.c_void_p), vol.size), "asw_download_volume")
        return vol

    # ---- driver-side pre/post-processing on the device (aswStereoMatch.cpp:30-31, 67-89, 97-98) ----
    def preprocess_pair(self, slot, left_full, right_full, dsize=(640, 360), detail_boost=True):
        """resize(img, Size(w, h)) + the HSV-V bilateral detail boost of the reference's main(); the result becomes the
        resident pair of `slot`."""
        li, la = _image(left_full)
        ri, ra = _image(right_full)
        rc = self._lib.asw_preprocess_pair(self._h, slot, C.byref(li), C.byref(ri), int(dsize[0]), int(dsize[1]),
                                           1 if detail_boost else 0)
        self._strict(rc, "asw_preprocess_pair")
        return True

    def download_pair(self, slot, shape):
        """The resident 8U pair of `slot` (e.g. after preprocess_pair); shape = (rows, cols, channels)."""
        left = np.zeros(shape, np.uint8)
        right = np.zeros(shape, np.uint8)
        li, _ = _image(left)
        ri, _ = _image(right)
        self._strict(self._lib.asw_download_pair(self._h, slot, C.byref(li), C.byref(ri)), "asw_download_pair")
        return left, right

    def download_disparity_u8(self, slot, shape, normalize=True):
        """disparityMap.convertTo(CV_8UC1) [+ normalize(0, 255, NORM_MINMAX)] of the last match of `slot` (main.cpp:97-98)."""
        out = np.zeros(shape, np.uint8)
        oi, _ = _image(out)
        self._strict(self._lib.asw_download_disparity_u8(self._h, slot, C.byref(oi), 1 if normalize else 0),
                     "asw_download_disparity_u8")
        return out

    def timing(self):
        t = AswTiming()
        self._lib.asw_get_timing(self._h, C.byref(t))
        return {"total_ms": t.total_ms, "aggregate_ms": t.aggregate_ms, "cost_ms": t.cost_ms,
                "aggregate_launches": t.aggregate_launches}


def stereoMatchingBatch(lefts, rights, disparityType, algorithmType, winSize=15, minDisparity=0, numDisparity=64,
                        device_ids=None, out=None):
    """asw_stereo_match_batch: frame i -> device device_ids[i % len(device_ids)], one host thread per device.

    out: optional list of C-contiguous float32 (H, W) arrays to receive the disparities (a frame loop that reuses its
    output buffers avoids first-touch page faults on 8 MB per 1080p frame)."""
    lib = _lib.lib()
    n = len(lefts)
    if device_ids is None:
        device_ids = list(range(max(1, lib.asw_device_count())))
    keep = []
    L = (AswImage * n)()
    R = (AswImage * n)()
    D = (AswImage * n)()
    outs = []
    for i in range(n):
        li, la = _image(lefts[i])
        ri, ra = _image(rights[i])
        if out is not None:
            o = out[i]
            if not (isinstance(o, np.ndarray) and o.dtype == np.float32 and o.flags.c_contiguous and o.shape == la.shape[:2]):
                raise ValueError("out[%d] must be a C-contiguous float32 array of shape %s" % (i, (la.shape[:2],)))
        else:
            o = np.zeros((la.shape[0], la.shape[1]), np.float32)
        di, _ = _image(o, 5)
        L[i], R[i], D[i] = li, ri, di
        keep.append((la, ra))
        outs.append(o)
    devs = (C.c_int * len(device_ids))(*device_ids)
    rc = lib.asw_stereo_match_batch(n, L, R, D, int(disparityType), int(algorithmType), winSize, minDisparity, numDisparity,
                                    len(device_ids), devs)
    _state.status = rc
    if rc in _SILENT:
        return None
    if rc != 0:
        raise AswError(rc, "asw_stereo_match_batch")
    return outs


# ---- module-level functions with the reference's names, bound to a lazily created default context ----
_default = {}
_default_lock = threading.Lock()


def default_context(device_id=None):
    if device_id is None:
        device_id = int(os.environ.get("ASW_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _default_lock:
        if device_id not in _default:
            _default[device_id] = Context(device_id)
        return _default[device_id]


def _bind(name):
    def f(*args, **kwargs):
        return getattr(default_context(), name)(*args, **kwargs)
    f.__name__ = name
    f.__doc__ = getattr(Context, name).__doc__
    return f


stereoMatching = _bind("stereoMatching")
computeAD = _bind("computeAD")
computeTAD = _bind("computeTAD")
computeSimilarity = _bind("computeSimilarity")
getCostSAD = _bind("getCostSAD")
getCostSAD_d = _bind("getCostSAD_d")
computeAdaptiveWeight = _bind("computeAdaptiveWeight")
computeAdaptiveWeight_direct8 = _bind("computeAdaptiveWeight_direct8")
computeAdaptiveWeight_GuidedF_3 = _bind("computeAdaptiveWeight_GuidedF_3")
computeNCC = _bind("computeNCC")
computeNCC_costs = _bind("computeNCC_costs")
computeAdaptiveWeight_geodesic = _bind("computeAdaptiveWeight_geodesic")
getGeodesicDist = _bind("getGeodesicDist")
getGuidedFilter = _bind("getGuidedFilter")
computeAdaptiveWeight_GuidedF = _bind("computeAdaptiveWeight_GuidedF")
computeAdaptiveWeight_GuidedF_2 = _bind("computeAdaptiveWeight_GuidedF_2")
computeAdaptiveWeight_WeightedMedian = _bind("computeAdaptiveWeight_WeightedMedian")
computeAdaptiveWeight_BLO1 = _bind("computeAdaptiveWeight_BLO1")
winnerTakeAll = _bind("winnerTakeAll")
