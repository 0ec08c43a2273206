"""Print a rocprofv3 kernel-stats CSV compactly: python tools/kstats.py file.csv [max rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f'{name[:110]:110s} calls {int(r["Calls"]):4d}  avg {float(r["AverageNs"]) / 1e6:8.4f} ms  min {float(r["MinNs"]) / 1e6:8.4f}  {float(r["Percentage"]):5.1f} %')
