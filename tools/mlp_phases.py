"""Diagnostic (stamps build, `make -C dnn-mppi-mpc_amd/csrc stamps`): where a wave of k_rollout_mlp_h3 spends its shader
cycles, summed over the T steps of one launch (workgroup 0, wave 1)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["MPPI_LIB"] = os.path.join(os.getcwd(), "dnn-mppi-mpc_amd", "lib", "libmppi_hip_stamps.so")
import dnn_mppi_mpc_amd as pkg
from oracle import mppi_oracle as mo
from bench import config2_kwargs
kw = config2_kwargs(K=int(os.environ.get("MLP_K", "16384")))
kw.update(param_exploration=0.05)
c = pkg.MPPIAlgorithms(**kw, learned_dynamics=mo.random_mlp_weights(0), waypoint_mode="frozen")
c._engine.set_state(np.zeros(3))
c._engine.run_closed_loop(3)
lib = C.CDLL(os.environ["MPPI_LIB"])
out = (C.c_ulonglong * 10)()
lib.mppi_debug_mlp_phases(out)
names = ["controls + zbuf + barrier", "input gemm", "input store", "barrier", "hidden gemm x3", "barrier", "hidden store x3",
         "barrier", "out layer", "barrier + advance"]
tot = sum(out)
for n, v in zip(names, out):
    print("%-28s %10d cycles  %5.1f %%" % (n, v, 100.0 * v / tot))
print("total", tot)
