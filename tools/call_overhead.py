"""Diagnostic: fixed cost of one mppi_run_closed_loop call (launch ramp + result copy + synchronisation) against its
per-iteration cost, hold phase of BASELINE config 2."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import dnn_mppi_mpc_amd as pkg
from bench import config2_kwargs
ctrl = pkg.MPPIAlgorithms(**config2_kwargs(), precision="f32", seed=1)
eng = ctrl._engine
eng.set_state(np.zeros(3)); eng.run_closed_loop(300); torch.cuda.synchronize()
for n in (1, 2, 5, 10, 20, 50, 200, 1000, 3000):
    ts = []
    for rep in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.run_closed_loop(n); ts.append(time.perf_counter() - t0)
    print(n, "iterations per call: median %.1f us per call, %.2f us per iteration" % (1e6 * np.median(ts), 1e6 * np.median(ts) / n))
