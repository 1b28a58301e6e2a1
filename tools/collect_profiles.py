"""Summaries of one `tools/make_profiles.sh <round>` run (gpurun_out/prof_<round>/), each stamped with the identity of the
kernel sources it was taken with (`build_id` = dnn_mppi_mpc_amd.source_id(); the bench lines of the same run carry it in
config.build_id and must agree).

  --on-box   (called by make_profiles.sh on the GPU box) the reduced counter rows of every PMC pass ->
             gpurun_out/prof_<round>/pmc.json AND profiles/<round>_pmc.json, so that the bench lines taken afterwards quote
             the counter figures of their own build
  (default)  in the build container: copy the run's files into profiles/<round>_*

<round>_pmc.json: {"build_id", "kernels": [{"kernel" (as rocprofv3 prints it), "grid", "wg" (work-items), "vgprs", "sgprs",
"lds", "launches_averaged", "counters": {name: mean per launch}}]} -- one entry per kernel instantiation AND launch size;
bench.py looks an entry up by the instantiation its timed launches took (mppi_get_rollout_kernel) and the launch's workgroup
count, and quotes nothing when there is none.  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts a
wide coalesced read at half its bytes, /opt/skills/guides/MI355X_MICROARCH.md, HBM section), applied where it is quoted.
usage: python tools/collect_profiles.py r03 [--on-box]
"""
import csv
import glob
import importlib.util
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r03"
ON_BOX = "--on-box" in sys.argv
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
DST = os.path.join(ROOT, "profiles")


def build_id():
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "dnn-mppi-mpc_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.source_id()


def pmc_json():
    entries = {}
    for path in sorted(glob.glob(os.path.join(SRC, "pmc_*.reduced.csv"))):
        for r in csv.DictReader(open(path)):
            if "mppi::" not in r["Kernel_Name"]:
                continue
            key = (r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
            e = entries.setdefault(key, {"kernel": r["Kernel_Name"], "grid": key[1], "wg": key[2], "vgprs": int(r["VGPR_Count"]),
                                         "sgprs": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]), "launches_averaged": 0,
                                         "counters": {}, "counter_launches": {}, "passes": []})
            n = int(r["Launches_Averaged"])
            # (the same counter from two passes -- SQ_WAVES rides along in several --: the pass with more launches)
            if n >= e["counter_launches"].get(r["Counter_Name"], -1):
                e["counters"][r["Counter_Name"]] = float(r["Mean_Value"])
                e["counter_launches"][r["Counter_Name"]] = n
            e["launches_averaged"] = max(e["launches_averaged"], n)
            tag = os.path.basename(path)[:-len(".reduced.csv")]
            if tag not in e["passes"]:
                e["passes"].append(tag)
    return {"build_id": build_id(),
            "_note": "rocprofv3 --pmc passes of the bench commands (tools/make_profiles.sh), one run per counter set, --kernel-trace "
                     "only; per kernel instantiation and launch size: mean per launch after dropping the first quarter of the "
                     "launches (tools/pmc_reduce.py).  FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (uncorrected)",
            "kernels": sorted(entries.values(), key=lambda e: (e["kernel"], e["grid"]))}


if ON_BOX:
    d = pmc_json()
    json.dump(d, open(os.path.join(SRC, "pmc.json"), "w"), indent=1)
    shutil.copy(os.path.join(SRC, "pmc.json"), os.path.join(DST, f"{R}_pmc.json"))
    print(f"{len(d['kernels'])} (kernel, launch size) entries, build {d['build_id']}")
    sys.exit(0)

bid = build_id()
pmc = json.load(open(os.path.join(SRC, "pmc.json")))
if pmc["build_id"] != bid:
    raise SystemExit(f"the profiles were taken with build {pmc['build_id']}, the tree is {bid}: run make_profiles.sh again")
for f in sorted(glob.glob(os.path.join(SRC, "bench*.json"))):
    line = open(f).read().strip().splitlines()
    if not line:
        print("empty", f)
        continue
    d = json.loads(line[-1])
    got = d.get("config", {}).get("build_id")
    if got != bid:
        raise SystemExit(f"{f} is of build {got}, the tree is {bid}")
    shutil.copy(f, os.path.join(DST, f"{R}_{os.path.basename(f)}"))
shutil.copy(os.path.join(SRC, "pmc.json"), os.path.join(DST, f"{R}_pmc.json"))
for f in sorted(glob.glob(os.path.join(SRC, "kernel_stats_*.csv"))) + sorted(glob.glob(os.path.join(SRC, "pmc_*.reduced.csv"))):
    shutil.copy(f, os.path.join(DST, f"{R}_{os.path.basename(f).replace('.reduced', '')}"))
if os.path.exists(os.path.join(SRC, "parity_margins.txt")):
    shutil.copy(os.path.join(SRC, "parity_margins.txt"), os.path.join(DST, f"{R}_parity_margins.txt"))
print("copied; build", bid)
for e in pmc["kernels"]:
    c = e["counters"]
    if "SQ_INSTS_VALU" in c and c.get("SQ_WAVES"):
        print("%-100s wgs %6d  VALU/wave %7.1f  launches %d" % (e["kernel"][:100], e["grid"] // e["wg"], c["SQ_INSTS_VALU"] / c["SQ_WAVES"], e["launches_averaged"]))
