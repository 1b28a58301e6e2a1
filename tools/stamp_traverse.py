"""Diagnostic: stamps of the traversal-phase kernels (the waypoint index moving) at BASELINE config 2 (or `c3`), first and
LAST workgroup of the rollout launch.  Uses lib/libmppi_hip_stamps.so (make stamps).  Not part of the product or the tests."""
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
from bench import config2_kwargs, config3_kwargs  # noqa: E402

C3 = len(sys.argv) > 1 and sys.argv[1] == "c3"  # config 3: two samples per wave (k_rollout_dual<..., LB>)
ctrl = pkg.MPPIAlgorithms(**(config3_kwargs() if C3 else config2_kwargs()), precision="f32", seed=1)
eng = ctrl._engine
lib = eng.lib
names = {0: "roll:start", 1: "roll:state loaded", 9: "roll:eps ready", 10: "roll:dynamics done", 11: "roll:pass A done",
         12: "roll:barrier 1", 13: "roll:look-back done (wave 0)", 14: "roll:costs by offset", 15: "roll:barrier 2",
         2: "roll:S done", 3: "roll:block sync", 4: "roll:end"}
order = (0, 1, 9, 10, 11, 12, 13, 14, 15, 2, 3, 4)
if C3:  # (k_rollout_dual's own stamps 11 .. 14 sit in its cost section: the look-back path has 27 .. 31)
    names.update({27: "roll:pass A done", 28: "roll:barrier 1", 29: "roll:look-back done (wave 0)", 30: "roll:costs by offset",
                  31: "roll:barrier 2"})
    order = (0, 1, 9, 10, 27, 28, 29, 30, 31, 2, 3, 4)
acc = {}
for ep in range(6):
    ctrl.restart_episode(np.zeros(3))
    eng.run_closed_loop(4)
    for _ in range(14):
        eng.run_closed_loop(1)
        buf = (C.c_ulonglong * 128)()
        lib.mppi_debug_stamps(buf, 128)
        t0 = min(buf[0], buf[64])
        for g in order:
            acc.setdefault(g, []).append(((buf[g] - t0) * 10.0, (buf[64 + g] - t0) * 10.0))
print("%-32s %10s %10s" % ("stamp", "first wg", "last wg"))
for g in order:
    a = np.array(acc[g])
    print("%-32s %+9.0f ns %+9.0f ns" % (names[g], np.median(a[:, 0]), np.median(a[:, 1])))
