"""Summarise rocprofv3 --pmc passes (their *_counter_collection.csv) into profiles/rNN_pmc_*.json, stamped with the
identity of the kernel sources (`build_id` = dnn_mppi_mpc_amd.source_id()): bench.py quotes a counter-derived figure
only from a profile of the build it runs.

  traffic: HBM bytes per launch.  FETCH_SIZE / WRITE_SIZE per /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
           hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for wide coalesced streams on gfx950 (FETCH_SIZE counts a
           128-byte read request as 64 B); raw and corrected values are both stored.
      pmc_summary.py traffic <fetch_pass.csv> <write_pass.csv> <kernel substring> <out.json>
  valu:    instructions per wave (SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_LDS over SQ_WAVES) and the VALU issue time they
           imply: per wave x waves per SIMD x 4 clocks (a wave64 VALU instruction occupies the SIMD16 for 4 clocks;
           quarter-rate instructions take longer, so this is a lower bound on the VALU pipe's busy time).
      pmc_summary.py valu <pass.csv> <kernel substring> <label> <out.json> [shader clock GHz, default 2.07]
           (an existing <out.json> of the same build is extended by the new label)
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_id():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "dnn-mppi-mpc_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.source_id()


def per_launch(path, counter, kern, skip_quarter=True):
    vals = []
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    if skip_quarter:
        vals = vals[len(vals) // 4:]  # skip warm-up launches
    return sum(vals) / max(1, len(vals)), len(vals)


def main():
    mode = sys.argv[1]
    if mode == "traffic":
        fetch_csv, write_csv, kern, out_path = sys.argv[2:6]
        fetch, nf = per_launch(fetch_csv, "FETCH_SIZE", kern)
        write, nw = per_launch(write_csv, "WRITE_SIZE", kern)
        out = {"build_id": build_id(), "kernel": kern, "launches_averaged": [nf, nw],
               "FETCH_SIZE_raw_KB": fetch, "WRITE_SIZE_raw_KB": write,
               "correction": "gfx950: FETCH_SIZE reads half of a wide coalesced stream -> x2 (MI355X_MICROARCH.md, HBM); "
                             "WRITE_SIZE exact",
               "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0}
    elif mode == "valu":
        pass_csv, kern, label, out_path = sys.argv[2:6]
        ghz = float(sys.argv[6]) if len(sys.argv) > 6 else 2.07
        raw = {c: per_launch(pass_csv, c, kern)[0] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES")}
        n = per_launch(pass_csv, "SQ_WAVES", kern)[1]
        waves = raw["SQ_WAVES"]
        per_wave = {"VALU": raw["SQ_INSTS_VALU"] / waves, "SALU": raw["SQ_INSTS_SALU"] / waves, "LDS": raw["SQ_INSTS_LDS"] / waves}
        wps = max(1, round(waves / 1024.0))  # 256 compute units x 4 SIMDs
        clocks = per_wave["VALU"] * wps * 4.0
        out = {}
        if os.path.exists(out_path):
            old = json.load(open(out_path))
            if old.get("build_id") == build_id():
                out = old
        out["build_id"] = build_id()
        out[label] = {"kernel": kern, "launches_averaged": n, "waves": waves, "per_wave": per_wave, "waves_per_simd": wps,
                      "valu_issue_clocks_per_simd": clocks, "shader_clock_GHz_in_kernel": ghz,
                      "valu_issue_us": clocks / ghz * 1e-3, "raw_per_launch": raw}
        out["_note"] = ("rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace (its own run); "
                        "valu_issue = instructions per wave x waves per SIMD x 4 clocks: a LOWER bound on the VALU pipe's "
                        "busy time (the 64-bit multiplies of Philox and the transcendentals are quarter-rate)")
    else:
        raise SystemExit(__doc__)
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out))


main()
