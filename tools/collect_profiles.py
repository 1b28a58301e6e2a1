"""Copy the measurements of one `tools/make_profiles.sh <round>` run (gpurun_out/prof_<round>/) into profiles/ and derive
the counter summaries bench.py quotes -- each stamped with the identity of the kernel sources it was taken with
(`build_id` = dnn_mppi_mpc_amd.source_id(); the bench line of the same run carries it in config.build_id and must agree).

  <round>_pmc_valu.json     instructions per wave (SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_LDS over SQ_WAVES) and the VALU
                            issue time they imply: per wave x waves per SIMD x 4 clocks (a wave64 VALU instruction
                            occupies the SIMD16 for 4 clocks; quarter-rate instructions take longer: a LOWER bound)
  <round>_pmc_traffic.json  HBM bytes per launch per /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
                            (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for wide coalesced streams on gfx950 (FETCH_SIZE counts
                            a 128-byte read request as 64 B); raw and corrected values both stored
usage: python tools/collect_profiles.py r02
"""
import csv
import glob
import importlib.util
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
DST = os.path.join(ROOT, "profiles")
GHZ = 2.07  # shader clock while these kernels run (clock64 against the wall clock, round 1 stamps build)


def build_id():
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "dnn-mppi-mpc_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.source_id()


def reduced(name):
    p = os.path.join(SRC, name + ".reduced.csv")
    return list(csv.DictReader(open(p))) if os.path.exists(p) else []


def counters(rows, kern):
    out, n = {}, 0
    for r in rows:
        if kern in r["Kernel_Name"]:
            out[r["Counter_Name"]] = float(r["Mean_Value"])
            n = int(r["Launches_Averaged"])
            out["_kernel"] = r["Kernel_Name"]
            out["_vgpr"], out["_sgpr"], out["_lds"] = int(r["VGPR_Count"]), int(r["SGPR_Count"]), int(r["LDS_Block_Size"])
    return out, n


def valu_entry(rows, kern):
    c, n = counters(rows, kern)
    if not c:
        return None
    waves = c["SQ_WAVES"]
    per_wave = {"VALU": c["SQ_INSTS_VALU"] / waves, "SALU": c["SQ_INSTS_SALU"] / waves, "LDS": c["SQ_INSTS_LDS"] / waves}
    wps = max(1, round(waves / 1024.0))  # 256 compute units x 4 SIMDs
    clocks = per_wave["VALU"] * wps * 4.0
    return {"kernel": c["_kernel"], "launches_averaged": n, "waves": waves, "per_wave": per_wave, "waves_per_simd": wps,
            "vgprs": c["_vgpr"], "sgprs": c["_sgpr"], "lds_bytes_per_workgroup": c["_lds"],
            "valu_issue_clocks_per_simd": clocks, "shader_clock_GHz_in_kernel": GHZ, "valu_issue_us": clocks / GHZ * 1e-3}


def copy(src_glob, dst_name):
    hits = sorted(glob.glob(os.path.join(SRC, src_glob), recursive=True), key=os.path.getmtime)
    if hits:  # (gpurun_out/ accumulates over calls: the newest)
        shutil.copy(hits[-1], os.path.join(DST, f"{R}_{dst_name}"))
        return True
    print("missing", src_glob)
    return False


bid = build_id()
bench = json.load(open(os.path.join(SRC, "bench.json")))
if bench["config"]["build_id"] != bid:
    raise SystemExit(f"the profiles were taken with build {bench['config']['build_id']}, the tree is {bid}: run make_profiles.sh again")
for f in ("bench.json", "bench_steps20.json", "bench_c3.json", "bench_c4.json", "bench_c5.json", "configs.jsonl"):
    copy(f, f)
if os.path.exists(os.path.join(ROOT, "gpurun_out", "issue_cost.txt")):  # tools/issue_cost.hip, when it was run this round
    shutil.copy(os.path.join(ROOT, "gpurun_out", "issue_cost.txt"), os.path.join(DST, f"{R}_issue_cost.txt"))
copy("kt/**/*kernel_stats.csv", "kernel_stats.csv")
copy("kt_trav/**/*kernel_stats.csv", "traversal_kernel_stats.csv")
copy("kt_cfg/**/*kernel_stats.csv", "configs34_kernel_stats.csv")
for d in ("pmc_inst", "pmc_fetch", "pmc_write", "pmc_trav", "pmc_cfg4"):
    copy(d + ".reduced.csv", d + ".csv")

valu = {"build_id": bid,
        "_note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace (its own run each: "
                 "tools/make_profiles.sh); valu_issue = instructions per wave x waves per SIMD x 4 clocks: an instruction-count "
                 "MODEL of the VALU pipe's busy time, not a bound -- measured issue costs at four waves per SIMD "
                 "(tools/issue_cost.hip, profiles/*_issue_cost.txt): 2.5 cycles for plain f32 / integer / logic operations on "
                 "VGPR or literal operands, 4.3 for DPP, compares, min/max, conversions, integer multiplies (v_mad_u64_u32 "
                 "included), packed f32 and ANY instruction with an SGPR operand, 8.3 for transcendentals"}
for label, rows, kern in (
        ("config 2: lean rollout kernel (the hold phase and every launch of the frozen index)", reduced("pmc_inst"), "k_rollout_fused<float, 0, 1, false, 2, false>"),
        ("config 2 traversal: rollout kernel that resolves the sequential index in one launch", reduced("pmc_trav"), "k_rollout_fused<float, 0, 1, false, 2, true>"),
        ("config 2 traversal: k_finalize with the map composition", reduced("pmc_trav"), "k_finalize<float, 0, 1, false, true, true>"),
        ("config 2: k_finalize (lean)", reduced("pmc_inst"), "k_finalize<float, 0, 1, false, true, false>"),
        ("config 4 shard: k_rollout_dual<float, racecar, 1 sample per wave, 2 in sequence>, K=8192 T=75", reduced("pmc_cfg4"), "k_rollout_dual<float, 1, 1, false, 2")):
    e = valu_entry(rows, kern)
    if e:
        valu[label] = e
json.dump(valu, open(os.path.join(DST, f"{R}_pmc_valu.json"), "w"), indent=1)

fetch, nf = counters(reduced("pmc_fetch"), "k_rollout_fused<float, 0, 1, false, 2, false>")
write, nw = counters(reduced("pmc_write"), "k_rollout_fused<float, 0, 1, false, 2, false>")
if fetch and write:
    traffic = {"build_id": bid, "kernel": fetch["_kernel"], "launches_averaged": [nf, nw],
               "FETCH_SIZE_raw_KB": fetch["FETCH_SIZE"], "WRITE_SIZE_raw_KB": write["WRITE_SIZE"],
               "correction": "gfx950: FETCH_SIZE reads half of a wide coalesced stream -> x2 (MI355X_MICROARCH.md, HBM); "
                             "WRITE_SIZE exact",
               "hbm_bytes_per_launch": (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0}
    json.dump(traffic, open(os.path.join(DST, f"{R}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else {"VALU": v["per_wave"]["VALU"], "issue_us": v["valu_issue_us"]})
                  for k, v in valu.items() if k != "_note"}, indent=1))
