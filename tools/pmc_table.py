"""Per-kernel table of the counter passes written by tools/pmc_run.sh: python tools/pmc_table.py <dir> <name>
(FETCH_SIZE is doubled: gfx950 tallies a 128-byte request as 64, MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import os
import re
import sys


def key(n):
    if "box_walk" in n:
        m = re.search(r"(\w+Src)", n)
        return {"StatsSrc": "stats", "ABSrc": "ab", "QSrc": "q"}.get(m.group(1), m.group(1))
    m = re.search(r"(k_\w+|__amd_\w+)", n)
    return m.group(1) if m else n[:30]


def load(d, name):
    acc = collections.OrderedDict()
    for c in ("FETCH_SIZE", "WRITE_SIZE", "L2", "SQ"):
        p = os.path.join(d, "pmc_%s_%s.csv" % (name, c))
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            k = key(r["Kernel_Name"])
            a = acc.setdefault(k, collections.defaultdict(float))
            a[r["Counter_Name"]] += float(r["Counter_Value"])
            a["vgpr"] = float(r["VGPR_Count"])
    return acc


def main(d, name):
    acc = load(d, name)
    tot = 0.0
    for k, a in acc.items():
        f, w = 2 * a.get("FETCH_SIZE", 0) * 1024, a.get("WRITE_SIZE", 0) * 1024
        if f + w < 1e6:
            continue
        tot += f + w
        hit = a.get("TCC_HIT_sum", 0) / max(1.0, a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0))
        wc = max(1.0, a.get("SQ_WAVE_CYCLES", 0))
        print("%-14s fetch %7.3f GB  write %7.3f GB  L2hit %.3f  vgpr %3d | waves %8d  VALU inst %.3g  valu_active/busy*4/1024 n/a  wait_any/wave_cycles %.2f  valu_active/wave_cycles %.3f  lds_active/wave_cycles %.3f"
              % (k, f / 1e9, w / 1e9, hit, a["vgpr"], a.get("SQ_WAVES", 0), a.get("SQ_INSTS_VALU", 0), a.get("SQ_WAIT_INST_ANY", 0) / wc,
                 a.get("SQ_ACTIVE_INST_VALU", 0) / wc, a.get("SQ_ACTIVE_INST_LDS", 0) / wc))
    print("total %.3f GB" % (tot / 1e9))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
