"""The C++ surface of the reference (include/aswMethods_mi355x.hpp) builds with plain g++ against the C-ABI
library, and -- on the GPU -- gives the same disparity as the ctypes path."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "shim_demo")
CEXE = os.path.join(ROOT, "tests", "cpp", "abi_demo")
EXE_CV = os.path.join(ROOT, "tests", "cpp", "shim_demo_cv")


def _build():
    from aswstereomatch_amd import build

    build.build()
    cmd = ["g++", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_demo.cpp"),
           "-L" + os.path.join(ROOT, "aswstereomatch_amd"), "-lasw_mi355x", "-Wl,-rpath," + os.path.join(ROOT, "aswstereomatch_amd"),
           "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)


def _build_cv():
    """The header's cv::Mat branch (-DASW_WITH_OPENCV) against tests/cpp/cv_stub/opencv2/core.hpp, a compile-only stand-in for the
    few OpenCV declarations the header touches (OpenCV itself is absent here)."""
    from aswstereomatch_amd import build

    build.build()
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-DASW_WITH_OPENCV", "-I" + os.path.join(ROOT, "tests", "cpp", "cv_stub"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_demo.cpp"),
           "-L" + os.path.join(ROOT, "aswstereomatch_amd"), "-lasw_mi355x", "-Wl,-rpath," + os.path.join(ROOT, "aswstereomatch_amd"),
           "-Wl,-rpath,/opt/rocm/lib", "-o", EXE_CV]
    subprocess.check_call(cmd)


def _build_c():
    from aswstereomatch_amd import build

    build.build()
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "abi_demo.c"), "-L" + os.path.join(ROOT, "aswstereomatch_amd"), "-lasw_mi355x",
           "-Wl,-rpath," + os.path.join(ROOT, "aswstereomatch_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", CEXE]
    subprocess.check_call(cmd)


def test_abi_header_is_plain_c99():
    # the drop-in boundary: no C++, no torch, no OpenCV types -- a C99 translation unit includes the header and links
    _build_c()
    assert os.path.exists(CEXE)


@pytest.mark.gpu
def test_c_program_matches_ctypes_path(tmp_path):
    import aswstereomatch_amd as asw
    from aswstereomatch_amd.synth import make_pair
    from oracle import asw_oracle as O

    if not os.path.exists(CEXE):
        _build_c()
    L, R, _ = make_pair(36, 64, 10, seed=6, block=12)
    L.tofile(tmp_path / "l.raw")
    R.tofile(tmp_path / "r.raw")
    ctx = asw.Context(0)
    for alg in (2, 3, 10):
        r = subprocess.run([CEXE, "36", "64", str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), str(alg), "7", "0", "10",
                            str(tmp_path / "d.raw"), str(tmp_path / "d8.raw")], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.startswith("ok 36 64"), (r.stdout, r.stderr)
        want = ctx.stereoMatching(L, R, asw.DISPARITY_LEFT, alg, 7, 0, 10)
        assert np.array_equal(np.fromfile(tmp_path / "d.raw", np.float32).reshape(36, 64), want)
        assert np.array_equal(np.fromfile(tmp_path / "d8.raw", np.uint8).reshape(36, 64), O.disparity_to_u8(want, True))
    ctx.close()


def test_shim_compiles_without_opencv():
    _build()
    assert os.path.exists(EXE)


def test_shim_cv_mat_branch_compiles():
    """Boundary hygiene, NOT parity evidence: the `#ifdef ASW_WITH_OPENCV` half of include/aswMethods_mi355x.hpp (view(), make(),
    AswPoint = cv::Point, every function on cv::Mat: the half a maintainer of the reference includes, M.h:91-184) is seen by a
    compiler, against a stand-in for the OpenCV declarations it names.  A typo there would otherwise ship unnoticed."""
    _build_cv()
    assert os.path.exists(EXE_CV)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["asw_mat", "cv_mat_stub"])
def test_shim_matches_ctypes_path(tmp_path, which):
    """Both Mat branches of the shim give the ctypes path's results on the GPU (the cv::Mat one on the stand-in Mat: it checks
    the header's plumbing -- strides, depth / channel mapping, empty-Mat returns --, not OpenCV and not the reference)."""
    import aswstereomatch_amd as asw
    from aswstereomatch_amd.synth import make_pair

    if which == "cv_mat_stub":
        _build_cv()
    elif not os.path.exists(EXE):
        _build()
    exe = EXE_CV if which == "cv_mat_stub" else EXE
    L, R, _ = make_pair(40, 72, 10, seed=5, block=12)
    L.tofile(tmp_path / "l.raw")
    R.tofile(tmp_path / "r.raw")
    ctx = asw.Context(0)
    for alg in (2, 8, 5):
        out = tmp_path / ("d%d.raw" % alg)
        r = subprocess.run([exe, "40", "72", str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), str(alg), "7", "0", "10", str(out)],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.startswith("ok 40 72 planes=4 sd=3 same=%d" % (1 if alg == 5 else -1)), (r.stdout, r.stderr)
        got = np.fromfile(out, np.float32).reshape(40, 72)
        assert np.array_equal(got, ctx.stereoMatching(L, R, asw.DISPARITY_LEFT, alg, 7, 0, 10))
    # the two header functions the round-1 shim lacked (M.h:141, 156): getGeodesicDist's std::map form and getCostSAD_d on a
    # caller-bordered view
    pre = tmp_path / "extra"
    r = subprocess.run([exe, "40", "72", str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), "2", "7", "0", "10", str(tmp_path / "y.raw"), str(pre)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "geo=%d sadd=1" % (40 * 72) in r.stdout, (r.stdout, r.stderr)
    sad = np.fromfile(str(pre) + ".sad", np.float32).reshape(40, 72)
    assert np.array_equal(sad, ctx.getCostSAD(L, R, asw.DISPARITY_LEFT, 7, 0, 10)[1])
    Rb = np.concatenate([R[:, :9][:, ::-1], R], axis=1)            # REFLECT border of max_offset = 9 columns on the left
    assert np.array_equal(sad, ctx.getCostSAD_d(L, Rb, 1, asw.DISPARITY_LEFT, 7))
    assert ctx.getCostSAD_d(L, R, 1, asw.DISPARITY_LEFT, 7) is None     # not bordered: Mat() in the reference
    geo = np.fromfile(str(pre) + ".geo", np.float32).reshape(7, 7)
    assert np.array_equal(geo, ctx.getGeodesicDist(L, 7, 3)[2, 3])
    # even window: the reference returns an empty Mat (M.cpp:1440-1443) -> so does the shim
    r = subprocess.run([exe, "40", "72", str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), "4", "6", "0", "10", str(tmp_path / "x.raw")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "empty"
    ctx.close()
