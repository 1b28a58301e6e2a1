"""Timing of the other BASELINE configs on one GPU (not the headline bench line): config 3 (diff-drive + 8
circles, K=16384), config 4's per-GPU shard and full K (race car + obstacles, T=75), config 5 (MLP dynamics on
MFMA, K=32768).  Closed loop on the device, in-kernel Philox, fp32.  Prints one JSON line per config."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dnn_mppi_mpc_amd as pkg  # noqa: E402
from oracle import mppi_oracle as mo  # noqa: E402  (path generators / random weights only)


def timed(eng, n, warm):
    eng.run_closed_loop(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_closed_loop(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def marginal_rollout(eng, n):
    p1 = timed(eng, n, 5)
    eng.set_rollout_repeats(2)
    p2 = timed(eng, n, 5)
    eng.set_rollout_repeats(1)
    return p1, max(p2 - p1, 1e-9)


def report(name, K, T, eng, n, flop_per_step=None):
    period, t_marg = marginal_rollout(eng, n)
    eng.enable_timing(True)  # in-situ duration of the rollout launch: event pair minus the empty-pair calibration
    eng.run_closed_loop(n)
    torch.cuda.synchronize()
    kms = eng.last_kernel_ms()
    eng.enable_timing(False)
    t_roll = max(kms["rollout"] * 1e-3, 1e-9)
    alg = 16.0 * K * T + 8.0 * K
    out = {"config": name, "K": K, "T": T, "us_per_iter": 1e6 * period, "traj_steps_per_s": K * T / period,
           "rollout_kernel_us": 1e6 * t_roll, "rollout_marginal_us": 1e6 * t_marg,
           "other_launches_us": 1e3 * (kms["reduce"] + kms["finalize"]),
           "algorithmic_GBs": alg / t_roll / 1e9, "hbm_frac": alg / t_roll / 8e12}
    if flop_per_step:
        # algorithmic flop of the network per launch over the launch's duration.  The default kernel issues every product
        # three times on the f16 matrix pipe (operands split into two f16 numbers each): its matrix-pipe load is 3 x that
        # against the 2.5 PFLOP/s dense f16 peak; MPPI_MLP_F32=1 runs the f32-input MFMA kernel (157.3 TFLOP/s peak)
        out["TFLOPs"] = flop_per_step * K * T / t_roll / 1e12
        if os.environ.get("MPPI_MLP_F32"):
            out["kernel"], out["mfma_f32_frac"] = "k_rollout_mlp (f32-input MFMA)", out["TFLOPs"] / 157.3
        else:
            out["kernel"], out["mfma_f16_issue_frac"] = "k_rollout_mlp_h3 (f16 x 3)", 3.0 * out["TFLOPs"] / 2500.0
    print(json.dumps(out))


which = sys.argv[1:] or ["3", "4s", "4", "5"]
dd = dict(delta_t=0.1, max_speed=5.0, max_omega=3.14, sigma=np.array([[0.1, 0.0], [0.0, 0.01]]),
          visualize_optimal_traj=False, visualze_sampled_trajs=False)
if "3" in which:
    rng = np.random.default_rng(1234)
    circles = [[2.0, 2.0, 0.4], [3.0, 3.5, 0.4]]
    while len(circles) < 8:
        x, y = rng.uniform(0.5, 4.5, 2)
        if x * x + y * y > 0.81:
            circles.append([float(x), float(y), 0.4])
    c = pkg.MPPIAlgorithms(**dd, ref_path=mo.generate_point_trajectory((0, 0), (5, 5), 100), num_samples_K=16384,
                           num_horizons_T=50, param_exploration=0.05, param_lambda=10.0, param_alpha=0.98,
                           stage_cost_weight=10 * np.array([5.0, 6.0, 9.0]), terminal_cost_weight=10 * np.array([5.0, 6.0, 9.0]),
                           obstacle_circles=np.array(circles), safety_margin_rate=0.8)
    c._engine.set_state(np.zeros(3))
    report("3: diff-drive + 8 circles", 16384, 50, c._engine, 500)
for tag, K in (("4s", 8192), ("4", 65536)):
    if tag in which:
        lem = mo.generate_lemniscate_racecar(100, 10.0)
        c = pkg.MPPIRacecarController(ref_path=lem, horizon_step_T=75, number_of_samples_K=K,
                                      obstacle_circles=np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]]),
                                      visualize_optimal_traj=False, visualze_sampled_trajs=False)
        c._engine.set_state(lem[0].astype(np.float64))
        report(f"4: race car + 2 circles, K={K}" + (" (one of 8 shards)" if tag == "4s" else " (all of K on one GPU)"),
               K, 75, c._engine, 30 if K > 10000 else 90)  # the driver's path has 100 waypoints: stay below its end
if "5" in which:
    K = int(os.environ.get("MLP_K", "32768"))
    c = pkg.MPPIAlgorithms(**dd, ref_path=mo.generate_point_trajectory((0, 0), (10, -5), 100), num_samples_K=K,
                           num_horizons_T=50, param_exploration=0.05, param_lambda=1.0, param_alpha=0.2,
                           stage_cost_weight=np.array([5.0, 5.0, 10.0]), terminal_cost_weight=np.array([5.0, 5.0, 10.0]),
                           learned_dynamics=mo.random_mlp_weights(0), waypoint_mode="frozen")
    c._engine.set_state(np.zeros(3))
    report("5: diff-drive + residual MLP on MFMA", K, 50, c._engine, 10, flop_per_step=1581056.0)
for tag, K in (("2x8", 32768), ("2x16", 65536), ("2x64", 262144)):
    if tag in which:  # config 2's problem with more samples per launch: where the HBM roofline becomes reachable
        c = pkg.MPPIAlgorithms(**dd, ref_path=mo.generate_point_trajectory((0, 0), (10, -5), 100), num_samples_K=K,
                               num_horizons_T=50, param_exploration=0.0001, param_lambda=1.0, param_alpha=0.2,
                               stage_cost_weight=np.array([5.0, 5.0, 10.0]), terminal_cost_weight=np.array([5.0, 5.0, 10.0]),
                               waypoint_mode="frozen")  # no speculation rounds: every iteration is one rollout launch
        c._engine.set_state(np.zeros(3))
        c._engine.run_closed_loop(100)
        report(f"2 scaled: diff-drive (frozen index), K={K}", K, 50, c._engine, 300)
