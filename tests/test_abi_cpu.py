"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/asw_mi355x.h declares; enum values equal the reference's.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

import aswstereomatch_amd as asw
from aswstereomatch_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    build.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_symbols_are_exported(so):
    hdr = open(os.path.join(ROOT, "include", "asw_mi355x.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|const char\*)\s+(asw_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(so, name), name


def test_only_the_abi_is_exported(so):
    # built with -fvisibility=hidden: the dynamic symbol table holds the C-ABI and nothing else of ours
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    funcs = {l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T"}
    assert funcs == set(_lib.ABI_SYMBOLS), funcs ^ set(_lib.ABI_SYMBOLS)


def test_enum_values_match_reference():
    # parametersStereo.h:4-24
    A = asw.StereoMatchingAlgorithms
    assert [int(A.BM), int(A.SGBM), int(A.ADAPTIVE_WEIGHT), int(A.ADAPTIVE_WEIGHT_8DIRECT), int(A.ADAPTIVE_WEIGHT_GEODESIC),
            int(A.ADAPTIVE_WEIGHT_BILATERAL_GRID), int(A.ADAPTIVE_WEIGHT_BLO1), int(A.ADAPTIVE_WEIGHT_GUIDED_FILTER),
            int(A.ADAPTIVE_WEIGHT_GUIDED_FILTER_2), int(A.ADAPTIVE_WEIGHT_GUIDED_FILTER_3), int(A.ADAPTIVE_WEIGHT_MEDIAN),
            int(A.NCC)] == list(range(12))
    assert int(asw.DISPARITY_LEFT) == 0 and int(asw.DISPARITY_RIGHT) == 1
    hdr = open(os.path.join(ROOT, "include", "asw_mi355x.h")).read()
    for name, val in [("ASW_ALG_ADAPTIVE_WEIGHT", 2), ("ASW_ALG_ADAPTIVE_WEIGHT_GEODESIC", 4),
                      ("ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER", 7), ("ASW_ALG_ADAPTIVE_WEIGHT_GUIDED_FILTER_2", 8),
                      ("ASW_ALG_ADAPTIVE_WEIGHT_BLO1", 6), ("ASW_ALG_ADAPTIVE_WEIGHT_MEDIAN", 10), ("ASW_ALG_NCC", 11)]:
        assert re.search(r"%s\s*=\s*%d\b" % (name, val), hdr), name


def test_status_strings(so):
    so.asw_status_string.restype = ctypes.c_char_p
    assert so.asw_status_string(0) == b"ok"
    assert b"odd" in so.asw_status_string(2)


def test_product_does_not_import_oracle():
    # the product package must never route through the CPU oracle
    pkg = os.path.join(ROOT, "aswstereomatch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                for needle in ("libasw_oracle", "asw_oracle", "from oracle", "import oracle", "oracle/"):
                    assert needle not in src.replace("CPU oracle under oracle/", ""), (f, needle)


def test_volume_planes_is_a_pure_host_function():
    # sizing rule of cost_volume_out (no GPU involved): inclusive candidate ranges have one more plane
    from aswstereomatch_amd import _lib

    l = _lib.lib()
    assert [l.asw_volume_planes(a, 64) for a in range(12)] == [0, 0, 65, 65, 65, 65, 64, 64, 64, 64, 64, 64]
    assert l.asw_volume_planes(99, 64) == 0


def test_algorithmic_bytes_is_one_function_for_bench_and_tools():
    """B_alg = 2*W*H*3 + W*H*N_d*4 + W*H*4 with N_d = D+1 for the inclusive ranges; bench.py and tools/pmc_traffic.py
    import the same function, and the committed counter records carry the value it gives for their shape."""
    import json

    from aswstereomatch_amd.roofline import algorithmic_bytes, candidates

    assert algorithmic_bytes(1920, 1080, candidates(8, 128)) == 1082419200      # C3
    assert algorithmic_bytes(1920, 1080, candidates(2, 128)) == 1090713600      # C5 frame
    assert algorithmic_bytes(1242, 375, candidates(4, 192)) == 364216500        # C4
    for name, alg in (("bilateral", 2), ("guided2", 8), ("geodesic", 4)):
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_%s.json" % name)))
        W, H, D, _ = rec["shape"]
        assert rec["algorithmic_bytes_per_frame"] == algorithmic_bytes(W, H, candidates(alg, D))
        assert rec["hbm_bytes_per_frame"] == sum(k["fetch_bytes_corrected"] + k["write_bytes"] for k in rec["per_kernel"].values()) or \
            abs(rec["hbm_bytes_per_frame"] - sum(k["fetch_bytes_corrected"] + k["write_bytes"] for k in rec["per_kernel"].values())) < 64
    import bench
    assert bench.algorithmic_bytes is algorithmic_bytes
