"""Per-kernel sums of rocprofv3 counter CSV lines (as cut by tools/prof_bilateral.sh): counter value per dispatch."""
import collections
import csv
import re
import sys

HDR = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
       "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
       "Counter_Value", "Start_Timestamp", "End_Timestamp"]
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in (r for r in csv.DictReader(open(path), fieldnames=HDR) if r["Counter_Value"] != "Counter_Value"):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        acc[(m.group(1) if m else r["Kernel_Name"][:40], r["VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(path)
    for k, v in acc.items():
        print("  ", k, {c: "%.4g" % (sum(x) / len(x)) for c, x in sorted(v.items())})
