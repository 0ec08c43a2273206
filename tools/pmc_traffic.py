"""HBM traffic per frame from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) of tools/run_one.py --reps N.

    python tools/pmc_traffic.py <fetch_csv> <write_csv> <reps> <out_json> <workload text> <alg> <W> <H> <D> <win> [aggregate regex]

Per MI355X_MICROARCH.md (HBM): FETCH_SIZE (KB) reports half the bytes of wide coalesced reads on gfx950 -> doubled; WRITE_SIZE (KB)
is exact for streaming stores.  Sums every kernel dispatch of the run and divides by the number of frames (reps).  `aggregate
regex`: the kernels between the aggregation events of the library (what bench.py's roofline.achieved is timed over); the others
(cost build, gray conversion, ...) are listed but not summed into hbm_bytes_per_frame.  The algorithmic bytes come from the one
function bench.py uses (aswstereomatch_amd/roofline.py)."""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aswstereomatch_amd.roofline import algorithmic_bytes, candidates  # noqa: E402


def kernel_key(name):
    if "box_walk" in name:
        m = re.search(r"(\w+Src\w*)", name)
        tag = {"StatsSrc": "stats", "ABSrc": "ab", "QSrc": "q", "QSrcP": "q", "SadSrc": "sad", "U8Src": "u8"}.get(m.group(1), m.group(1)) if m else "?"
        return "k_box_walk[%s]" % tag
    m = re.search(r"(k_\w+(?:<[\w, ]+>)?)", name)
    return m.group(1) if m else name[:40]


def load(path, counter):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        key = kernel_key(r["Kernel_Name"])
        per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
    return per


def main(fetch_csv, write_csv, reps, out_json, workload, alg, W, H, D, win, agg_regex=None):
    reps, alg, W, H, D, win = int(reps), int(alg), int(W), int(H), int(D), int(win)
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    kernels, outside = {}, {}
    total = 0.0
    for k in list(f.keys()) + [k for k in w if k not in f]:
        if k.startswith("__amd") or "copyBuffer" in k:
            continue
        fb, wb = 2 * f.get(k, 0.0) * 1024 / reps, w.get(k, 0.0) * 1024 / reps
        rec = {"fetch_bytes_corrected": int(fb), "write_bytes": int(wb)}
        if agg_regex is None or re.search(agg_regex, k):
            kernels[k] = rec
            total += fb + wb
        else:
            outside[k] = rec
    balg = algorithmic_bytes(W, H, candidates(alg, D))
    out = {"workload": workload, "shape": [W, H, D, win],
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of tools/run_one.py, "
           "%d frames; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (HBM section)" % reps,
           "per_kernel": kernels, "outside_the_aggregation_events": outside,
           "hbm_bytes_per_frame": int(total), "algorithmic_bytes_per_frame": int(balg),
           "ratio_to_algorithmic": round(total / float(balg), 3)}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:12])
