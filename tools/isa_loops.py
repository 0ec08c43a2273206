"""Instruction mix of the loops of one kernel in an `hipcc -S --cuda-device-only` listing (VALU count, LDS instructions by
kind, scratch traffic): python tools/isa_loops.py listing.s <substring of the mangled kernel name> [min VALU per loop]."""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(lines) if key in l and l.rstrip().endswith(":") or (key in l and re.match(r"^_Z\S+:", l)))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if not (m and m.group(1) in labels and labels[m.group(1)] < i):
        continue
    a, c = labels[m.group(1)], Counter()
    for t in (x.strip().split() for x in body[a:i + 1]):
        if not t or t[0][0] in ".;":
            continue
        op = t[0]
        if op.startswith("ds_") or op.startswith("scratch_") or op.startswith("global_") or op.startswith("buffer_"):
            c[op] += 1
        elif op.startswith("v_"):
            c["VALU"] += 1
    if c["VALU"] >= minv:
        print(a, i, dict(c))
