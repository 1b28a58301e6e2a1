"""Diagnostic: latency of the two phases of the reference driver's run at BASELINE config 2 -- holding the goal (search
window of one candidate) and traversing the path (window of 20 candidates, repair launches for the sequential
waypoint index).  MPPI_LIB=<other build> A/Bs two libraries on the same box.  Not part of the product or the tests."""
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch
import dnn_mppi_mpc_amd as pkg
from bench import config2_kwargs
ctrl = pkg.MPPIAlgorithms(**config2_kwargs(), precision="f32", seed=1)
eng = ctrl._engine
eng.set_state(np.zeros(3)); eng.run_closed_loop(300); torch.cuda.synchronize()
res = []
for rep in range(6):
    t0 = time.perf_counter(); eng.run_closed_loop(3000); torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 3000 * 1e6)
eng.set_u_prev(np.zeros((50, 2))); eng.set_waypoint_idx(0); eng.set_state(np.zeros(3)); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.run_closed_loop(23); torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / 23 * 1e6
eng.run_closed_loop(200); eng.enable_timing(True); eng.run_closed_loop(2000); torch.cuda.synchronize(); km = eng.last_kernel_ms(); eng.enable_timing(False)
print("  hold-phase event pairs: rollout %.2f finalize %.2f us" % (1e3 * km["rollout"], 1e3 * km["finalize"]))
print(os.environ.get("MPPI_LIB", "new")[-12:], "hold min %.2f med %.2f" % (min(res), sorted(res)[3]), "traverse %.1f" % tr)
eng.set_u_prev(np.zeros((50, 2))); eng.set_waypoint_idx(0); eng.set_state(np.zeros(3)); torch.cuda.synchronize()
eng.enable_timing(True); eng.run_closed_loop(22); torch.cuda.synchronize(); km = eng.last_kernel_ms(); eng.enable_timing(False)
print("  traversal event pairs: rollout %.2f finalize %.2f us (pair overhead %.2f)" % (1e3 * km["rollout"], 1e3 * km["finalize"], 1e3 * km["event_pair_overhead"]), eng.counters())
