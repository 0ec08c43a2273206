"""HBM traffic per frame from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) of tools/run_one.py --reps N.

    python tools/pmc_traffic.py <fetch_csv> <write_csv> <reps> <out_json> <workload text> <algorithmic bytes>

Per MI355X_MICROARCH.md (HBM): FETCH_SIZE (KB) reports half the bytes of wide coalesced reads on gfx950 -> doubled; WRITE_SIZE (KB)
is exact for streaming stores.  Sums every kernel dispatch of the run and divides by the number of frames (reps)."""
import collections
import csv
import json
import re
import sys


def load(path, counter):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(k_\w+(?:<[\w, ]+>)?)", r["Kernel_Name"])
        key = m.group(1) if m else r["Kernel_Name"][:40]
        per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
    return per


def main(fetch_csv, write_csv, reps, out_json, workload, balg):
    reps = int(reps)
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    kernels = {}
    total = 0.0
    for k in list(f.keys()) + [k for k in w if k not in f]:
        if k.startswith("__amd") or "copyBuffer" in k:
            continue
        fb, wb = 2 * f.get(k, 0.0) * 1024 / reps, w.get(k, 0.0) * 1024 / reps
        kernels[k] = {"fetch_bytes_corrected": int(fb), "write_bytes": int(wb)}
        total += fb + wb
    out = {"workload": workload, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of tools/run_one.py, "
           "%d frames; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (HBM section)" % reps,
           "per_kernel": kernels, "hbm_bytes_per_launch": int(total), "hbm_bytes_per_frame": int(total),
           "algorithmic_bytes_per_launch": int(balg), "ratio_to_algorithmic": round(total / float(balg), 3)}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:7])
