"""Diagnostic: where does one iteration spend its time?  Uses lib/libmppi_hip_stamps.so (make stamps),
a build with s_memrealtime stamps of block 0 at phase boundaries.  Not part of the product or the tests."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import dnn_mppi_mpc_amd as pkg  # noqa: E402
from dnn_mppi_mpc_amd import _capi  # noqa: E402

_capi.LIB_PATH = os.environ.get("MPPI_STAMPS_LIB") or os.path.join(ROOT, "dnn-mppi-mpc_amd", "lib", "libmppi_hip_stamps.so")
from bench import config2_kwargs  # noqa: E402

which = sys.argv[2] if len(sys.argv) > 2 else "2"   # "2" (bench.py's workload) or "4s" (race car, K=8192, T=75)
if which == "4s":
    from oracle import mppi_oracle as mo  # noqa: E402  (path generator only)
    lem = mo.generate_lemniscate_racecar(100, 10.0)
    ctrl = pkg.MPPIRacecarController(ref_path=lem, horizon_step_T=75, number_of_samples_K=8192,
                                     obstacle_circles=np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]]),
                                     visualize_optimal_traj=False, visualze_sampled_trajs=False)
    eng = ctrl._engine
    eng.set_state(lem[0].astype(np.float64))
    WARM = 20
else:
    ctrl = pkg.MPPIAlgorithms(**config2_kwargs(), precision=sys.argv[1] if len(sys.argv) > 1 else "f32", seed=1)
    eng = ctrl._engine
    eng.set_state(np.zeros(3))
    WARM = 300
eng.run_closed_loop(WARM)
lib = eng.lib
names = {0: "roll:start", 1: "roll:state loaded", 8: "roll:chunk start", 9: "roll:eps ready", 10: "roll:dynamics done",
         11: "roll:index done", 12: "roll:stage cost 0", 13: "roll:stage cost 1", 14: "roll:terminal", 2: "roll:S done", 3: "roll:block sync", 4: "roll:end", 16: "fin:start",
         17: "fin:prefetch issued", 24: "fin:merge loads issued", 25: "fin:min done", 26: "fin:eta done",
         18: "fin:merge done", 19: "fin:filter done", 20: "fin:shift done", 21: "fin:end"}
acc = {}
N = 50 if which == "2" else 30
for _ in range(N):
    eng.run_closed_loop(1)
    buf = (C.c_ulonglong * 64)()
    lib.mppi_debug_stamps(buf, 64)
    for grp in ((0, 1, 8, 9, 10, 11) + ((12, 13, 14) if which != "2" else ()) + (2, 3, 4), (16, 17, 24, 25, 26, 18, 19, 20, 21)):
        base = buf[grp[0]]
        for g in grp:
            acc.setdefault(g, []).append((buf[g] - base) * 10.0)  # ns (100 MHz)
for grp in ((0, 1, 8, 9, 10, 11) + ((12, 13, 14) if which != "2" else ()) + (2, 3, 4), (16, 17, 24, 25, 26, 18, 19, 20, 21)):
    for g in grp:
        print(f"{names[g]:28s} +{np.median(acc[g]):8.0f} ns")
    print()
# shader clock during the rollout kernel: s_memtime ticks per s_memrealtime tick (100 MHz)
buf = (C.c_ulonglong * 64)()
eng.run_closed_loop(1)
lib.mppi_debug_stamps(buf, 64)
dt_wall = (buf[4] - buf[0]) * 10e-9
print(f"shader clock during k_rollout: {(buf[44] - buf[40]) / dt_wall / 1e6:.0f} MHz over {dt_wall*1e6:.2f} us")
dt_fin = (buf[21] - buf[16]) * 10e-9
print(f"shader clock during k_finalize: {(buf[53] - buf[48]) / dt_fin / 1e6:.0f} MHz over {dt_fin*1e6:.2f} us")
