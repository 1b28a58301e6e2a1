"""Reduce a rocprofv3 *_counter_collection.csv to one row per (kernel, counter): launches, mean value (after dropping the
first quarter of the launches as warm-up) -- what tools/collect_profiles.py needs, at a size that travels back."""
import collections
import csv
import sys

acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"], r["Counter_Name"], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"],
         r["LDS_Block_Size"])].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Counter_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                "Launches", "Launches_Averaged", "Mean_Value"])
    for k, v in sorted(acc.items()):
        u = v[len(v) // 4:]
        w.writerow(list(k) + [len(v), len(u), sum(u) / max(1, len(u))])
