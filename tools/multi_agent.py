"""Several independent MPPI problems (agents) on ONE GPU: one handle + one HIP stream + one host thread per agent,
closed loops on the device.  Launches of different agents overlap (one agent's serial k_finalize runs beside the
others' rollouts), so the aggregate rate exceeds the single-agent one until the rollouts fill the chip.
Prints one JSON line per agent count."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dnn_mppi_mpc_amd as pkg  # noqa: E402
from bench import HORIZON, K_SAMPLES, config2_kwargs  # noqa: E402

N_ITER = 3000
for n_agents in [int(a) for a in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    ctrls = [pkg.MPPIAlgorithms(**config2_kwargs(), precision="f32", seed=100 + a) for a in range(n_agents)]
    streams = [torch.cuda.Stream() for _ in ctrls]
    for c in ctrls:
        c._engine.set_state(np.zeros(3))
        c._engine.run_closed_loop(200)
    torch.cuda.synchronize()
    go = threading.Barrier(n_agents + 1)

    def work(c, s):
        go.wait()
        c._engine.run_closed_loop(N_ITER, stream=s)

    ths = [threading.Thread(target=work, args=(c, s)) for c, s in zip(ctrls, streams)]
    for t in ths:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"agents": n_agents, "K": K_SAMPLES, "T": HORIZON, "iterations_each": N_ITER,
                      "us_per_iteration_per_agent": 1e6 * dt / N_ITER,
                      "aggregate_traj_steps_per_s": n_agents * K_SAMPLES * HORIZON * N_ITER / dt}))
