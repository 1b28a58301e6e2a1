"""Summarise rocprofv3 --pmc passes into profiles/rNN_pmc_traffic.json (HBM bytes per launch).

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 64-byte requests... per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for wide
coalesced streams on gfx950 (FETCH_SIZE counts 128-B read requests as 64 B); we store both the raw and the
corrected value and say which correction was applied.
usage: pmc_summary.py <fetch_counter_csv> <write_counter_csv> <kernel substring> <out.json>
"""
import csv
import json
import sys


def per_launch(path, counter, kern):
    vals = []
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    vals = vals[len(vals) // 4:]  # skip warm-up launches
    return sum(vals) / max(1, len(vals)), len(vals)


fetch, nf = per_launch(sys.argv[1], "FETCH_SIZE", sys.argv[3])
write, nw = per_launch(sys.argv[2], "WRITE_SIZE", sys.argv[3])
out = {"kernel": sys.argv[3], "launches_averaged": [nf, nw],
       "FETCH_SIZE_raw_KB": fetch, "WRITE_SIZE_raw_KB": write,
       "correction": "gfx950: FETCH_SIZE reads half of a wide coalesced stream -> x2 (MI355X_MICROARCH.md, HBM); "
                     "WRITE_SIZE exact",
       "k_rollout_fused_hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(out)
