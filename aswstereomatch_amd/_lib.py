"""ctypes loader for libasw_mi355x.so -- the C-ABI of include/asw_mi355x.h.

The product path has NO CPU fallback: if the HIP library is missing or fails to load, every
entry point raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported
from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libasw_mi355x.so")

# every symbol include/asw_mi355x.h declares
ABI_SYMBOLS = [
    "asw_create", "asw_destroy", "asw_status_string", "asw_device_count", "asw_set_gray_bits",
    "asw_stereo_match", "asw_upload_pair", "asw_match_resident", "asw_download_disparity",
    "asw_download_volume", "asw_synchronize", "asw_get_timing",
    "asw_aggregate_bilateral", "asw_aggregate_geodesic", "asw_aggregate_guided", "asw_aggregate_guided2",
    "asw_aggregate_wmedian", "asw_aggregate_blo1", "asw_aggregate_bilgrid", "asw_aggregate_direct8", "asw_aggregate_guided3",
    "asw_cost_ncc", "asw_ncc_disparity",
    "asw_preprocess_pair", "asw_download_pair", "asw_download_disparity_u8",
    "asw_cost_ad", "asw_cost_tad", "asw_cost_sd", "asw_cost_similarity", "asw_cost_sad", "asw_cost_sad_d",
    "asw_guided_filter", "asw_geodesic_dist", "asw_wta", "asw_bgr2gray", "asw_lr_check", "asw_volume_planes",
    "asw_stereo_match_batch",
]


class AswImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("channels", C.c_int),
                ("depth", C.c_int), ("step", C.c_size_t)]


class AswTiming(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("aggregate_ms", C.c_float), ("cost_ms", C.c_float),
                ("aggregate_launches", C.c_int)]


class AswError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        try:
            msg = lib().asw_status_string(status).decode()
        except Exception:  # pragma: no cover
            msg = "?"
        super().__init__("%s failed: status %d (%s)" % (where, status, msg))


_lib = None


def lib():
    """Load the HIP library; raise loudly when it is absent (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libasw_mi355x.so is not built (%s). Run `python -m aswstereomatch_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        l.asw_status_string.restype = C.c_char_p
        l.asw_status_string.argtypes = [C.c_int]
        l.asw_destroy.restype = None
        l.asw_destroy.argtypes = [C.c_void_p]
        l.asw_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        P = C.c_void_p
        I = C.c_int
        D = C.c_double
        IMG = C.POINTER(AswImage)
        l.asw_stereo_match.argtypes = [P, IMG, IMG, IMG, I, I, I, I, I, P, C.c_size_t]
        l.asw_upload_pair.argtypes = [P, I, IMG, IMG]
        l.asw_match_resident.argtypes = [P, I, I, I, I, I, I, I]
        l.asw_download_disparity.argtypes = [P, I, IMG]
        l.asw_download_volume.argtypes = [P, I, P, C.c_size_t]
        l.asw_synchronize.argtypes = [P]
        l.asw_set_gray_bits.argtypes = [P, I]
        l.asw_get_timing.argtypes = [P, C.POINTER(AswTiming)]
        l.asw_aggregate_bilateral.argtypes = [P, IMG, IMG, IMG, D, D, I, I, I, I, P, C.c_size_t]
        l.asw_aggregate_geodesic.argtypes = [P, IMG, IMG, IMG, I, I, I, I, P, C.c_size_t]
        l.asw_aggregate_direct8.argtypes = [P, IMG, IMG, IMG, I, I, I, I, P, C.c_size_t]
        l.asw_aggregate_guided3.argtypes = [P, IMG, IMG, IMG, I, D, I, I, I, P, C.c_size_t]
        l.asw_cost_ncc.argtypes = [P, IMG, IMG, P, I, I, I, I, I]
        l.asw_ncc_disparity.argtypes = [P, IMG, IMG, IMG, I, I, I, I]
        l.asw_preprocess_pair.argtypes = [P, I, IMG, IMG, I, I, I]
        l.asw_download_pair.argtypes = [P, I, IMG, IMG]
        l.asw_download_disparity_u8.argtypes = [P, I, IMG, I]
        l.asw_aggregate_guided.argtypes = [P, IMG, IMG, IMG, I, D, I, I, I, P, C.c_size_t]
        l.asw_aggregate_guided2.argtypes = [P, IMG, IMG, IMG, I, D, I, I, I, P, C.c_size_t]
        l.asw_aggregate_wmedian.argtypes = [P, IMG, IMG, IMG, I, I, D, D, I, I, P, C.c_size_t]
        l.asw_aggregate_blo1.argtypes = [P, IMG, IMG, IMG, I, D, I, I, I, P, C.c_size_t]
        l.asw_aggregate_bilgrid.argtypes = [P, IMG, IMG, IMG, I, D, D, I, I, P, C.c_size_t]
        l.asw_cost_ad.argtypes = [P, IMG, IMG, P, I, I, I]
        l.asw_cost_sd.argtypes = [P, IMG, IMG, P, I, I, I]
        l.asw_cost_tad.argtypes = [P, IMG, IMG, P, I, I, I, I]
        l.asw_cost_similarity.argtypes = [P, IMG, IMG, P, D, D, D, I, I, I, I]
        l.asw_cost_sad.argtypes = [P, IMG, IMG, P, I, I, I, I]
        l.asw_cost_sad_d.argtypes = [P, IMG, IMG, P, I, I, I]
        l.asw_guided_filter.argtypes = [P, IMG, P, P, I, D]
        l.asw_geodesic_dist.argtypes = [P, IMG, P, I, I]
        l.asw_wta.argtypes = [P, P, I, I, I, I, P]
        l.asw_volume_planes.argtypes = [I, I]
        l.asw_lr_check.argtypes = [P, P, P, I, I, C.c_float, C.c_float, P, P]
        l.asw_bgr2gray.argtypes = [P, IMG, P]
        l.asw_stereo_match_batch.argtypes = [I, IMG, IMG, IMG, I, I, I, I, I, I, C.POINTER(C.c_int)]
        _lib = l
    return _lib
