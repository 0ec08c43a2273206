"""In-tree build of libasw_mi355x.so (HIP, gfx950 only).  hipcc cross-compiles without a GPU.

    python -m aswstereomatch_amd.build          # incremental
    python -m aswstereomatch_amd.build --force
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libasw_mi355x.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the reference is MSVC/SSE2 code that never fuses a*b+c, and the parity
# oracle is built the same way; fused forms are written explicitly where they are provably exact.
CXXFLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-fno-fast-math",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function",
] + os.environ.get("HIPCC_EXTRA", "").split()  # measurement builds, e.g. -DASW_XQ_ABLATION (tools/prof_bilateral.sh)


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _compile(src, obj):
    cmd = [HIPCC] + CXXFLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return src, r.returncode, r.stdout


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(s, o) or hdr_time > os.path.getmtime(o):
            jobs.append((s, o))
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for src, rc, out in ex.map(lambda j: _compile(*j), jobs):
                if verbose or rc != 0:
                    sys.stderr.write(out)
                if rc != 0:
                    raise RuntimeError("hipcc failed on %s" % src)
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
