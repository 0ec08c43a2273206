"""Rebuilds profiles/pmc_bilateral.json and profiles/pmc_guided2.json from the counter CSVs of a refresh_profiles.sh run.

    python tools/update_pmc_json.py gpurun_out/<tag>
"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "box_walk" in n:
            key = {"StatsSrc": "stats", "ABSrc": "ab", "QSrc": "q"}.get(re.search(r"(\w+Src)", n).group(1), "other")
        else:
            m = re.search(r"(k_\w+|__amd_\w+)", n)
            key = m.group(1) if m else n
        out.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return out


def main(d):
    bf, bw = load(os.path.join(d, "pmc_bilateral_FETCH_SIZE.csv")), load(os.path.join(d, "pmc_bilateral_WRITE_SIZE.csv"))
    f, w = bf["k_asw_bilateral"]["FETCH_SIZE"], bw["k_asw_bilateral"]["WRITE_SIZE"]
    p = os.path.join(ROOT, "profiles", "pmc_bilateral.json")
    j = json.load(open(p))
    j.update({"FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024)})
    json.dump(j, open(p, "w"), indent=1)
    gf, gw, gl = (load(os.path.join(d, "pmc_guided2_%s.csv" % n)) for n in ("FETCH_SIZE", "WRITE_SIZE", "L2"))
    ks = ["stats", "ab", "q", "k_wta"]
    F = {k: gf[k]["FETCH_SIZE"] / 1024 for k in ks}
    W = {k: gw[k]["WRITE_SIZE"] / 1024 for k in ks}
    tot = sum(2 * F[k] + W[k] for k in ks) * 1024 * 1024
    p = os.path.join(ROOT, "profiles", "pmc_guided2.json")
    j = json.load(open(p))
    j["FETCH_SIZE_MB_raw"] = {k.replace("k_", ""): round(F[k], 1) for k in ks}
    j["WRITE_SIZE_MB"] = {k.replace("k_", ""): round(W[k], 1) for k in ks}
    j["L2_hit_rate"] = {k.replace("k_", ""): round(gl[k]["TCC_HIT_sum"] / (gl[k]["TCC_HIT_sum"] + gl[k]["TCC_MISS_sum"]), 3) for k in ks}
    j["hbm_bytes_per_launch"] = int(tot / 4)
    j["hbm_bytes_per_frame"] = int(tot)
    j["cost_build_bytes_per_frame"] = int((2 * gf["k_similarity"]["FETCH_SIZE"] + gw["k_similarity"]["WRITE_SIZE"]) * 1024)
    json.dump(j, open(p, "w"), indent=1)
    print("bilateral bytes/launch", json.load(open(os.path.join(ROOT, "profiles", "pmc_bilateral.json")))["hbm_bytes_per_launch"])
    print("guided2 bytes/frame", int(tot), j["L2_hit_rate"])


if __name__ == "__main__":
    main(sys.argv[1])
