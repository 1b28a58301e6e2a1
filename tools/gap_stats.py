"""Diagnostic: from a rocprofv3 --kernel-trace CSV, the idle gaps between consecutive kernels of the closed loop (end of one
dispatch to the start of the next) -- a loop the HOST cannot feed fast enough shows gaps well above the ~1.5-2 us of a
dependent kernel boundary.   python tools/gap_stats.py <kernel_trace.csv>"""
import csv
import sys

import numpy as np

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_rollout_fused" in r["Kernel_Name"] or "k_finalize" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
gap = (st[1:] - en[:-1]) * 1e-3
dur = (en - st) * 1e-3
is_roll = np.array(["k_rollout_fused" in r["Kernel_Name"] for r in rows])
sel = slice(len(gap) // 4, None)  # drop the warm-up quarter
g = gap[sel]
print("kernels %d; gap us: median %.2f mean %.2f p90 %.2f p99 %.2f max %.1f" % (len(rows), np.median(g), g.mean(), np.percentile(g, 90), np.percentile(g, 99), g.max()))
print("duration us: rollout median %.2f, finalize median %.2f" % (np.median(dur[is_roll]), np.median(dur[~is_roll])))
period = (st[2:] - st[:-2]) * 1e-3
print("period of two launches us: median %.2f" % np.median(period[len(period) // 4:]))
