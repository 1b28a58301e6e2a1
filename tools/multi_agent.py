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

# the same agents BATCHED in one handle: agents as a grid dimension, one launch per stage for all of them
from dnn_mppi_mpc_amd import _capi as capi  # noqa: E402

kw = config2_kwargs()
for n_agents in [int(a) for a in (sys.argv[1:] or ["1", "2", "4", "8", "16", "32"])]:
    eng = pkg.Engine(model=capi.MODEL_DIFFDRIVE, K=K_SAMPLES, T=HORIZON, delta_t=kw["delta_t"], u_max=[kw["max_speed"], kw["max_omega"]],
                     param_exploration=kw["param_exploration"], param_lambda=kw["param_lambda"], param_alpha=kw["param_alpha"],
                     sigma=np.asarray(kw["sigma"]).reshape(-1), stage_cost_weight=list(kw["stage_cost_weight"]) + [0.0],
                     terminal_cost_weight=list(kw["terminal_cost_weight"]) + [0.0], search_window=20, filter_window=10,
                     clamp_rollout=1, clamp_u_after_update=0, waypoint_mode=capi.WAYPOINT_FROZEN, seed=5, n_agents=n_agents)
    eng.set_ref_path(kw["ref_path"])
    eng.set_state(np.zeros((n_agents, 3)) if n_agents > 1 else np.zeros(3))
    eng.run_closed_loop(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_closed_loop(N_ITER)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"agents_batched_in_one_handle": n_agents, "K": K_SAMPLES, "T": HORIZON, "iterations": N_ITER,
                      "us_per_iteration_all_agents": 1e6 * dt / N_ITER,
                      "aggregate_traj_steps_per_s": n_agents * K_SAMPLES * HORIZON * N_ITER / dt}))
