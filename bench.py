#!/usr/bin/env python3
"""Headline benchmark: trajectory-steps/s of the MPPI iteration (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5] [--no-cpu-baseline] [--no-batched] [--no-graph-timing]

A "step" is one closed-loop MPPI iteration (sample -> rollout -> cost -> softmin weight -> reduce ->
filter -> shift, then the driver's plant advances the state).  Default workload `c2` = BASELINE config 2:
differential-drive, K=4096 samples x T=50 horizon, fp32, reference `__main__` parameters
(controllers/mppi_differential_drive.py:400-410), synthetic straight-line path.  State, controls and the
Philox-keyed noise live on the GPU; nothing crosses PCIe inside the timed region.

The timed steps are the iterations of the reference driver's own run, episodes back to back: the robot starts at the
head of the 100-waypoint path with a fresh controller and the loop runs tSim = 1000 iterations
(mppi_differential_drive.py:396) -- some 23 of them traverse the path (the waypoint index moves, the search window is
SEARCH_IDX_LEN = 20 candidates long (:204)), the rest hold the goal (window of one candidate).  A run that never leaves
the hold phase would flatter the number, so the bench restarts the episode every 1000 iterations of its run
(initialisation + warm-up + timed steps; three small uploads, timed when they fall into the timed region) and reports
both phase latencies as well.  N > 1: one process per GPU (torch.distributed / RCCL), every rank evaluates K=4096 of
K_global = N*4096 samples and ONE exchange of {rho, eta, eta2, W[T,2]} per iteration merges the softmin (weak scaling).

`--workload c4` = BASELINE config 4, the line north_star's ">= 6x at 8 GPUs" refers to: race car + 2 circular
obstacles, K = 65536 samples x T = 75 in TOTAL, K/N per GPU (strong scaling), same exchange.

`--workload c5` = BASELINE config 5: diff-drive with the learned residual dynamics (Linear 5-512, 3 x [Linear 512-512,
tanh], Linear 512-3; train/train_diff_mlp.py:13-36) on the matrix cores, K = 32768 samples x T = 50 in TOTAL, K/N per GPU
(strong scaling), frozen waypoint index; its `roofline` is the MFMA one.

The default line also carries `batched_agents`: 32 independent config-2 problems in ONE handle (agents as a grid
dimension) -- the launch size at which the chip, not the launch, is the limit.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the launch stream, over whole episodes
of fixed length whatever --steps is (see `kernel_duration`); `cpu_baseline` is the plain-C oracle (test
infrastructure, oracle/mppi_oracle.c) timed on the host, rank 0 at N=1 only -- the only place this file touches `oracle/`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_SAMPLES, HORIZON = 4096, 50
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # same guide: dense f16/bf16 matrix peak
MLP_FLOP_PER_STEP = 1581056.0  # 2 x (5 x 512 + 3 x 512 x 512 + 512 x 3) per trajectory-step
EPISODE = 1000   # tSim of the reference driver (controllers/mppi_differential_drive.py:396)
TRAVERSE = 23    # iterations the robot needs from the head of the path to its goal (measured; reported separately)
X_INIT = np.zeros(3)  # init_x, :394
PROFILE_ROUND = "r03"  # profiles/<round>_pmc.json: counter passes of the bench commands, per kernel, stamped with the build
SHADER_GHZ = 2.07     # shader clock while the analytic rollout kernels run (clock64 against the wall clock, stamps build)


def config2_kwargs(K=K_SAMPLES, T=HORIZON):
    """BASELINE config 2 = the reference's `__main__` problem (mppi_differential_drive.py:394-419) at K=4096, T=50."""
    x = np.linspace(0.0, 10.0, 100)
    y = np.linspace(0.0, -5.0, 100)
    yaw = np.arctan2(-5.0, 10.0) * np.ones(100)
    return dict(delta_t=0.1, ref_path=np.array([x, y, yaw]).T, max_speed=5.0, max_omega=3.14, num_samples_K=K,
                num_horizons_T=T, param_exploration=0.0001, param_lambda=1.0, param_alpha=0.2,
                sigma=np.array([[0.1, 0.0], [0.0, 0.01]]), stage_cost_weight=np.array([5.0, 5.0, 10.0]),
                terminal_cost_weight=np.array([5.0, 5.0, 10.0]), visualize_optimal_traj=False,
                visualze_sampled_trajs=False)


def config4_path():
    """`generate_lemniscate_trajectory(100, 10)` of mppi_race_car_obstacle.py:288-299 (f32 linspace)."""
    t = np.linspace(0, 2 * np.pi, 100, dtype=np.float32)
    x = 10.0 * np.cos(t) / (1 + np.sin(t) ** 2)
    y = 10.0 * np.sin(t) * np.cos(t) / (1 + np.sin(t) ** 2)
    yaw = np.arctan2(np.gradient(y), np.gradient(x))
    return np.stack([x, y, yaw, np.ones_like(t) * 5.0], axis=1)


def config4_kwargs(K=65536, T=75):
    """BASELINE config 4 = the race-car-obstacle controller's defaults (mppi_race_car_obstacle.py:13-29) at K=65536, T=75."""
    return dict(ref_path=config4_path(), horizon_step_T=T, number_of_samples_K=K,
                obstacle_circles=np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]]), collision_safety_margin_rat=1.5,
                visualize_optimal_traj=False, visualze_sampled_trajs=False)


def config3_kwargs(K=16384, T=50):
    """BASELINE config 3 = the `_obs` controller's `__main__` problem (mppi_differential_drive_obs.py:436-452) with 8 circles
    (the two defaults + six drawn with default_rng(1234), none covering the start) at K=16384, T=50."""
    rng = np.random.default_rng(1234)
    circles = [[2.0, 2.0, 0.4], [3.0, 3.5, 0.4]]
    while len(circles) < 8:
        x, y = rng.uniform(0.5, 4.5, 2)
        if x * x + y * y > 0.81:
            circles.append([float(x), float(y), 0.4])
    x = np.linspace(0.0, 5.0, 100)
    return dict(delta_t=0.1, ref_path=np.array([x, x, np.arctan2(5.0, 5.0) * np.ones(100)]).T, max_speed=5.0, max_omega=3.14,
                num_samples_K=K, num_horizons_T=T, param_exploration=0.05, param_lambda=10.0, param_alpha=0.98,
                sigma=np.array([[0.1, 0.0], [0.0, 0.01]]), stage_cost_weight=10 * np.array([5.0, 6.0, 9.0]),
                terminal_cost_weight=10 * np.array([5.0, 6.0, 9.0]), obstacle_circles=np.array(circles),
                safety_margin_rate=0.8, visualize_optimal_traj=False, visualze_sampled_trajs=False)


def config5_kwargs(K=32768, T=50):
    """BASELINE config 5: config 2's problem with x' = x + dt (f + MLP([x, v])) (test/bullet_differential_drive_dnn.py:79-92)."""
    kw = config2_kwargs(K, T)
    kw["param_exploration"] = 0.05
    return kw


def config5_weights():
    """The reference checkpoint saved_models/mlp_diff_300x100_3l.pth as plain arrays (tests/golden, made by
    oracle/gen_golden.py) when the file is there, else random weights of the same architecture."""
    path = os.path.join(ROOT, "tests", "golden", "mlp_diff_300x100_3l_weights.npz")
    if os.path.exists(path):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}, "the reference checkpoint mlp_diff_300x100_3l (as arrays)"
    rng = np.random.default_rng(0)
    w = {"input_layer.weight": rng.normal(0, 0.3, (512, 5)), "input_layer.bias": rng.normal(0, 0.1, 512),
         "out_layer.weight": rng.normal(0, 0.05, (3, 512)), "out_layer.bias": rng.normal(0, 0.01, 3)}
    for i in range(3):
        w[f"hidden_layer.{i}.weight"] = rng.normal(0, 0.04, (512, 512))
        w[f"hidden_layer.{i}.bias"] = rng.normal(0, 0.1, 512)
    return w, "random weights of the reference architecture"


def batched_agents(n_agents=32, iters=400):
    """SURVEY.md section 8 f1, many MPPI problems per launch: `n_agents` independent config-2 problems (own state, nominal
    controls, waypoint index, noise stream) in ONE handle, agents as a grid dimension; frozen waypoint index (the
    sequential one cannot be batched).  Returns the aggregate rate over whole closed-loop iterations."""
    import torch
    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    kw = config2_kwargs()
    eng = pkg.Engine(model=capi.MODEL_DIFFDRIVE, K=K_SAMPLES, T=HORIZON, delta_t=kw["delta_t"],
                     u_max=[kw["max_speed"], kw["max_omega"]], param_exploration=kw["param_exploration"],
                     param_lambda=kw["param_lambda"], param_alpha=kw["param_alpha"], sigma=np.asarray(kw["sigma"]).reshape(-1),
                     stage_cost_weight=list(kw["stage_cost_weight"]) + [0.0],
                     terminal_cost_weight=list(kw["terminal_cost_weight"]) + [0.0], search_window=20, filter_window=10,
                     clamp_rollout=1, clamp_u_after_update=0, waypoint_mode=capi.WAYPOINT_FROZEN, seed=5, n_agents=n_agents)
    eng.set_ref_path(kw["ref_path"])
    eng.set_state(np.zeros((n_agents, 3)))
    eng.run_closed_loop(8)  # initialisation (code objects), as in the main line
    torch.cuda.synchronize()

    def timed(n):
        t0 = time.perf_counter()
        eng.run_closed_loop(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    dt = timed(EPISODE)       # a whole episode of the driver's run from the head of the path (with a frozen index the
    dt_hold = timed(iters)    # agents take some 300 iterations to its end), then the hold phase alone
    per_iter = n_agents * K_SAMPLES * HORIZON
    alg = (16.0 * K_SAMPLES * HORIZON + 8.0 * K_SAMPLES) * n_agents
    return {"agents": n_agents, "K_per_agent": K_SAMPLES, "T": HORIZON, "value": per_iter / dt, "unit": "trajectory-steps/s",
            "us_per_iteration_all_agents": 1e6 * dt, "iterations": EPISODE,
            "hold_phase": {"value": per_iter / dt_hold, "us_per_iteration_all_agents": 1e6 * dt_hold, "iterations": iters},
            "algorithmic_GBs_over_whole_iterations": alg / dt / 1e9, "hbm_frac_over_whole_iterations": alg / dt / 1e9 / HBM_PEAK_GBS,
            "hbm_frac_hold_phase": alg / dt_hold / 1e9 / HBM_PEAK_GBS,
            "rollout_layout": eng.counters()["rollout_layout"],
            "note": "one rollout launch + one finalize launch per iteration for all agents; algorithmic bytes as in `roofline` "
                    "over the WHOLE iteration (serial tail included), so a lower bound on the rollout launch's own fraction"}


def strong_scaling_bound_c4(t_full, ranks=8):
    """What K sharded over `ranks` GPUs can gain at config 4 BEFORE any exchange: one shard's iteration (K / ranks samples, the
    same closed loop, measured here on this GPU) against the whole K's.  The serial tail of an iteration (merge, filter,
    plant, next x0 call, two kernel boundaries) does not shrink with the shard, so the ratio stays below `ranks`."""
    import torch
    import dnn_mppi_mpc_amd as pkg
    K, T = 65536 // ranks, 75
    x_init = config4_path()[0].astype(np.float64)
    ctrl = pkg.MPPIRacecarController(**config4_kwargs(K, T), precision="f32", device=torch.cuda.current_device(), seed=2024)
    eng = ctrl._engine

    def run(n):
        done = 0
        while done < n:  # the driver's loop runs over its 100 waypoints (mppi_race_car_obstacle.py:336)
            if done % 100 == 0:
                ctrl.restart_episode(x_init)
            m = min(n - done, 100 - done % 100)
            eng.run_closed_loop(m)
            done += m
    run(100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(600)
    torch.cuda.synchronize()
    t_shard = (time.perf_counter() - t0) / 600
    eng.enable_timing(True)
    run(200)
    kms = eng.last_kernel_ms()
    eng.enable_timing(False)
    return {"ranks": ranks, "K_per_rank": K, "us_per_iteration_whole_K_one_gpu": 1e6 * t_full, "us_per_iteration_one_shard": 1e6 * t_shard,
            "speedup_bound_before_exchange": t_full / t_shard,
            "shard_kernel_us": {"rollout": 1e3 * kms["rollout"], "merge": 1e3 * kms["reduce"], "finalize": 1e3 * kms["finalize"]},
            "note": "measured on this one GPU (a shard's rollout does not depend on the others'); the exchange of the per-rank "
                    "records (824 B per rank, inside k_finalize or one collective) comes on top"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """Cores this process may really use: the cgroup CPU quota when there is one (a GPU box gives each GPU's job a
    share of the host, e.g. 16 of 256), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get("MPPI_BENCH_CPU_THREADS", "16")))  # (16 = the pool's CPU share per GPU)


def cpu_baseline(budget_s=10.0, budget_all_s=6.0):
    """The C restatement of the reference loop (config 2, closed loop, eps pre-generated): the reference's own sequential
    waypoint index on ONE host core (it cannot be parallelised over samples), and beside it the per-rollout variant (the
    index threads through each sample's own calls and restarts at every sample) on all host cores (OpenMP over K) --
    SURVEY.md section 8d-ii."""
    from oracle import c_oracle, mppi_oracle, philox
    kw = config2_kwargs()
    pool = [philox.sample_epsilon(kw["sigma"], 1, i, K_SAMPLES, HORIZON) for i in range(4)]

    def run(budget, threads):
        o = c_oracle.DiffDriveC(**kw)
        o.iteration(X_INIT.copy(), pool[0], per_rollout_threads=threads)  # warm the caches / start the thread team
        n, spent, state = 0, 0.0, X_INIT.copy()
        while spent < budget:
            if n % EPISODE == 0:  # the same run as the GPU path times: a new episode of the driver every tSim iterations
                o = c_oracle.DiffDriveC(**kw)
                state = X_INIT.copy()
            t0 = time.perf_counter()
            out = o.iteration(state, pool[n % len(pool)], per_rollout_threads=threads)
            spent += time.perf_counter() - t0
            state = mppi_oracle.diffdrive_plant_step(state, out["u0_returned"], kw["delta_t"])
            n += 1
        return n, spent

    n1, s1 = run(budget_s, 0)
    cores = host_cores()
    na, sa = run(budget_all_s, cores)
    return {"value": K_SAMPLES * HORIZON * n1 / s1, "unit": "trajectory-steps/s", "cores": 1, "kind": "port",
            "cpu_model": cpu_model(),
            "sample": f"the first {n1} iterations of the same run (K=4096, T=50, episodes of {EPISODE}) in {s1:.1f} s "
                      "(oracle/mppi_oracle.c, scalar f64, the reference's sequential waypoint index, noise pre-generated "
                      "and excluded)",
            "ms_per_step": 1e3 * s1 / n1,
            "all_cores": {"value": K_SAMPLES * HORIZON * na / sa, "unit": "trajectory-steps/s", "cores": cores,
                          "ms_per_step": 1e3 * sa / na,
                          "sample": f"{na} iterations of the same run in {sa:.1f} s with the waypoint index threaded PER ROLLOUT "
                                    "(through each sample's own calls, mppi_differential_drive.py:228,:244, restarting at every "
                                    "sample: samples independent), OpenMP over K on every core this process may use"}}


def pmc_kernel(pmc, kernel, workgroups):
    """Counters of ONE kernel instantiation from profiles/<round>_pmc.json (tools/collect_profiles.py): the entry whose
    rocprofv3 name contains `kernel` and -- the same instantiation serves launches of several sizes -- whose launch had
    `workgroups` workgroups.  None when there is none: a figure of another kernel or launch size is never quoted."""
    if pmc is None:
        return None
    hits = [e for e in pmc.get("kernels", []) if kernel in e["kernel"] and (workgroups is None or e["grid"] // e["wg"] == workgroups)]
    return max(hits, key=lambda e: e.get("launches_averaged", 0)) if hits else None


def stamped_profile(name, build_id):
    """profiles/<round>_<name>.json if it was taken with THIS build of the kernels, else (None, why)."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{name}.json")
    if not os.path.exists(path):
        return None, f"no profiles/{PROFILE_ROUND}_{name}.json"
    d = json.load(open(path))
    if d.get("build_id") != build_id:
        return None, f"profiles/{PROFILE_ROUND}_{name}.json is of build {d.get('build_id')}, this is {build_id}"
    return d, f"profiles/{PROFILE_ROUND}_{name}.json (build {build_id})"


def eps_hbm_line(args):
    """`--eps hbm`: the form of the analytic path north_star calls memory-bound -- the noise tensor of `_calc_epsilon`
    (mppi_differential_drive.py:273-283) MATERIALISED in HBM and read by the rollout as coalesced [K, T, 2] rows, instead of
    drawn in registers.  A ring of 8 tensors (mppi_set_noise_ring; > 256 MiB in total, so that no slot survives in the
    Infinity Cache until it is read again) is filled once by the engine's sampler; the timed closed loop then reads one
    slot per iteration.  Workloads: `c4` = config 4's size on one GPU (K = 65536 x T = 75, 39 MB of noise per launch);
    `c2` = 32 batched config-2 agents (52 MB per launch: a single config-2 problem's 1.6 MB never leaves the caches).
    The rollout launch's duration is taken with per-launch event pairs (calibrated) -- a repeated launch would re-read its
    slot from the cache, so the marginal method of the default line does not apply -- and `traffic` from the FETCH_SIZE /
    WRITE_SIZE passes of THIS command under profiles/.  One GPU."""
    import torch
    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    torch.cuda.set_device(0)
    n_slots = 8
    if args.workload == "c4":
        K, T, n_agents, episode = 65536, 75, 1, 100
        ctrl = pkg.MPPIRacecarController(**config4_kwargs(K, T), precision="f32", device=0, seed=2024)
        eng, x_init = ctrl._engine, config4_path()[0].astype(np.float64)
        restart = lambda: ctrl.restart_episode(x_init)
        what = ("BASELINE config 4's size on one GPU: race-car + 2 circular obstacles, K=65536 x T=75, noise read from a "
                "materialised [K,T,2] tensor in HBM")
    else:
        K, T, n_agents, episode = K_SAMPLES, HORIZON, 32, EPISODE
        kw = config2_kwargs()
        eng = pkg.Engine(model=capi.MODEL_DIFFDRIVE, K=K, T=T, delta_t=kw["delta_t"], u_max=[kw["max_speed"], kw["max_omega"]],
                         param_exploration=kw["param_exploration"], param_lambda=kw["param_lambda"], param_alpha=kw["param_alpha"],
                         sigma=np.asarray(kw["sigma"]).reshape(-1), stage_cost_weight=list(kw["stage_cost_weight"]) + [0.0],
                         terminal_cost_weight=list(kw["terminal_cost_weight"]) + [0.0], search_window=20, filter_window=10,
                         clamp_rollout=1, clamp_u_after_update=0, waypoint_mode=capi.WAYPOINT_FROZEN, seed=5, n_agents=n_agents)
        eng.set_ref_path(kw["ref_path"])

        def restart():
            eng.set_u_prev(np.zeros((n_agents, T, 2)))
            eng.set_waypoint_idx(0)
            eng.set_state(np.zeros((n_agents, 3)))
        what = ("32 independent BASELINE config-2 problems (diff-drive, K=4096 x T=50 each, frozen waypoint index) batched in "
                "one handle, noise read from a materialised [32,K,T,2] tensor in HBM")
    ring = torch.empty((n_slots,) + eng._lead + (K, T, 2), dtype=torch.float32, device="cuda:0")
    for i in range(n_slots):
        eng.sample_epsilon(i, out=ring[i])
    torch.cuda.synchronize()
    steps = args.steps if args.steps is not None else 400
    warm = args.warmup if args.warmup is not None else 40

    def run(n, pos=[0]):
        while n > 0:
            if pos[0] % episode == 0:
                restart()
            m = min(n, episode - pos[0] % episode)
            eng.run_closed_loop(m)
            pos[0] += m
            n -= m

    def timed(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    out = {}
    for mode in ("philox", "hbm"):
        eng.set_noise_ring(ring if mode == "hbm" else None)
        run.__defaults__[0][0] = 0
        run(8)
        run(max(1, warm))
        dt = timed(steps)
        eng.enable_timing(True)
        run(min(steps, 400))
        kms = eng.last_kernel_ms()
        eng.enable_timing(False)
        out[mode] = {"s_per_step": dt / steps, "rollout_us": 1e3 * kms["rollout"], "finalize_us": 1e3 * kms["finalize"],
                     "merge_us": 1e3 * kms["reduce"], "kernel": eng.rollout_kernel()}
    units = n_agents * K * T
    alg = (16.0 * K * T + 8.0 * K) * n_agents
    read = 8.0 * K * T * n_agents  # what the rollout really has to fetch: one pass over the noise (the second stays in registers)
    build_id = pkg.source_id()
    h = out["hbm"]
    pmc, pmc_src = stamped_profile("pmc", build_id)
    # (the kernel that serves a noise tensor runs in this command only: its one entry is this launch size)
    pk = pmc_kernel(pmc, h["kernel"], None) if h["kernel"] else None
    traffic = None
    if pk is not None and "FETCH_SIZE" in pk["counters"] and "WRITE_SIZE" in pk["counters"]:
        traffic = (2.0 * pk["counters"]["FETCH_SIZE"] + pk["counters"]["WRITE_SIZE"]) * 1024.0
    t_roll = 1e-6 * h["rollout_us"]
    valu = None
    if pk is not None and "SQ_INSTS_VALU" in pk["counters"]:
        per_wave = pk["counters"]["SQ_INSTS_VALU"] / pk["counters"]["SQ_WAVES"]
        issue_us = per_wave * pk["counters"]["SQ_WAVES"] / 1024.0 * 4.0 / SHADER_GHZ * 1e-3
        valu = {"valu_instructions_per_wave": per_wave, "waves_per_simd": pk["counters"]["SQ_WAVES"] / 1024.0,
                "issue_us": issue_us, "valu_issue_frac": min(1.0, issue_us / h["rollout_us"])}
    line = {"metric": "trajectory-steps/sec (KxT/iter_time), noise tensor read from HBM, " + ("race-car K=65536 T=75" if args.workload == "c4" else "32 x diff-drive K=4096 T=50"),
            "value": units / h["s_per_step"], "unit": "trajectory-steps/s", "n_gpus": 1, "steps": steps, "warmup": warm,
            "ms_per_step": 1e3 * h["s_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": what, "K_per_problem": K, "T": T, "problems_per_launch": n_agents,
                       "noise": "ring of %d materialised tensors (%.0f MB each, %.0f MB in all), slot = iteration mod %d, "
                                "filled by the engine's Philox sampler" % (n_slots, read / 1e6, n_slots * read / 1e6, n_slots),
                       "build_id": build_id},
            "roofline": {"bound": "hbm", "kernel": h["kernel"], "achieved": alg / t_roll / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / t_roll / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": pmc_src if pk is not None else (pmc_src if pmc is None else pmc_src + " holds no counters of this launch"),
                         "algorithmic_bytes_per_launch": alg, "noise_bytes_read_per_launch": read,
                         "noise_read_GBs": read / t_roll / 1e9, "noise_read_frac_of_peak": read / t_roll / 1e9 / HBM_PEAK_GBS,
                         "kernel_us": h["rollout_us"], "kernel_us_method": "per-launch event pairs minus the empty-pair calibration "
                                                                           "(mppi_enable_timing), averaged over the timed launches",
                         "valu_issue": valu,
                         "note": "algorithmic bytes = 16 B per trajectory-step (SURVEY.md section 8d counts the reference's two "
                                 "passes over the noise) + 8 B per trajectory; the fused kernel reads the tensor ONCE (8 B per "
                                 "step, `noise_bytes_read_per_launch`) and keeps it in registers for the weighted sum, so the "
                                 "bytes it really moves are half the algorithmic figure"},
            "same_run_with_the_in_kernel_sampler": {"value": units / out["philox"]["s_per_step"], "ms_per_step": 1e3 * out["philox"]["s_per_step"],
                                                    "rollout_kernel_us": out["philox"]["rollout_us"], "kernel": out["philox"]["kernel"]},
            "finalize_us": h["finalize_us"], "merge_us": h["merge_us"]}
    print(json.dumps(line))


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves -- one child process per GPU
    running this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly what `torch.distributed.run` would
    export -- relay rank 0's JSON line and fail if any rank does.  This parent never imports torch and never touches a
    GPU (nothing that has initialised the GPU is ever re-exec'd; the children are fresh interpreters)."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL and the peer-to-peer exchange need it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    import threading
    chunks, rc = [], 0
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("MPPI_BENCH_RANK_TIMEOUT_S", "1500"))
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                break  # a rank failed: the others would wait for it at the next barrier or exchange
            if time.time() > deadline:
                print("bench.py: the ranks did not finish within MPPI_BENCH_RANK_TIMEOUT_S; stopping them", file=sys.stderr)
                rc = 1
                break
            time.sleep(0.05)
    finally:
        for p in procs:  # a rank that failed (or an interrupt here) must not leave the others waiting at a barrier
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
    reader.join(timeout=10)
    out0 = "".join(c or "" for c in chunks)
    rcs = [p.returncode for p in procs]
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    for ln in out0.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if any(rcs) or not lines:
        print(f"bench.py: ranks exited with {rcs}" + ("" if lines else "; rank 0 printed no JSON line"), file=sys.stderr)
        rc = next((c for c in rcs if c), 1)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5"], default="c2")
    ap.add_argument("--eps", choices=["philox", "hbm"], default="philox",
                    help="hbm: the noise tensor materialised in HBM and read by the rollout (one GPU, --workload c2 = 32 batched "
                         "config-2 agents, or c4); see eps_hbm_line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true")
    ap.add_argument("--no-graph-timing", action="store_true",
                    help="skip the graph-replay measurement of the rollout launch (the profiler passes: counter collection "
                         "and graph replays do not go together)")
    args = ap.parse_args()
    c5 = args.workload == "c5"
    c3 = args.workload == "c3"
    if c3 and args.gpus > 1:
        raise SystemExit("--workload c3 is BASELINE's one-GPU config (sequential waypoint index: not sharded)")
    if args.eps == "hbm":
        if args.gpus != 1 or args.workload not in ("c2", "c4"):
            raise SystemExit("--eps hbm: one GPU, --workload c2 (32 batched agents) or c4")
        return eps_hbm_line(args)
    if args.steps is None:
        args.steps = 40 if c5 else 2000  # (an iteration of config 5 takes milliseconds)
    if args.warmup is None:
        args.warmup = 5 if c5 else 200
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us (the driver's `python3 bench.py --gpus N ...`): start the N ranks, relay rank 0's line
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MPPI_BENCH_LAUNCH_CHECK"):
        # tests/test_bench_launcher.py (no GPU): the ranks the launcher started find each other and rank 0's line travels
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        dist.barrier()
        if os.environ["MPPI_BENCH_LAUNCH_CHECK"] == "fail" and rank == world - 1:
            raise SystemExit(3)
        if rank == 0:
            print("a line that is not the result")
            print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": float(t.item()), "steps": args.steps}))
        dist.destroy_process_group()
        return
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # rehearsal of the N>1 path on a one-GPU box: every rank on MPPI_BENCH_DEVICE, gloo instead of RCCL (which
    # needs one GPU per rank); the driver's runs set neither
    if "MPPI_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MPPI_BENCH_DEVICE"])
    backend = os.environ.get("MPPI_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    pg = None
    sharded = world > 1 or bool(os.environ.get("MPPI_BENCH_FORCE_SHARDED"))  # rehearsal of the N>1 path on 1 GPU
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        pg = dist.group.WORLD

    import dnn_mppi_mpc_amd as pkg
    c4 = args.workload == "c4"
    if c5:
        K_global, T, episode, traverse = 32768, 50, 1000, 0
        K_local = pkg.distributed.shard_range(K_global, rank, world)[1]
        x_init = X_INIT
        weights, weights_src = config5_weights()
        make = lambda: pkg.MPPIAlgorithms(**config5_kwargs(K_global, T), precision="f32", device=local_rank, seed=2024,
                                          process_group=pg, waypoint_mode="frozen", learned_dynamics=weights)
    elif c3:
        K_global, T, episode, traverse = 16384, 50, EPISODE, 0
        K_local, x_init = K_global, X_INIT
        make = lambda: pkg.MPPIAlgorithms(**config3_kwargs(K_global, T), precision="f32", device=local_rank, seed=2024)
    elif c4:
        K_global, T, episode, traverse = 65536, 75, 100, 0  # the driver's loop runs over its 100 waypoints (:336)
        K_local = pkg.distributed.shard_range(K_global, rank, world)[1]
        x_init = config4_path()[0].astype(np.float64)
        make = lambda: pkg.MPPIRacecarController(**config4_kwargs(K_global, T), precision="f32", device=local_rank,
                                                 seed=2024, process_group=pg)
    else:
        K_global, T, episode, traverse = K_SAMPLES * world, HORIZON, EPISODE, TRAVERSE
        K_local, x_init = K_SAMPLES, X_INIT
        # K sharded: the reference's one index cannot travel between ranks inside an iteration.  MPPI_BENCH_SHARD_MODE =
        # frozen (default: every call searches from the x0 call's index, the race-car files' rule, 12 us per iteration) or
        # per_rollout (the index threads through each sample's own calls, closer to :228,:244, 27 us per iteration while the
        # robot travels: a dependent chain of T + 1 searches per sample)
        shard_mode = os.environ.get("MPPI_BENCH_SHARD_MODE", "frozen") if sharded else None
        make = lambda: pkg.MPPIAlgorithms(**config2_kwargs(K=K_global), precision="f32", device=local_rank, seed=2024,
                                          process_group=pg, waypoint_mode=shard_mode)
    stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def measure():
        ctrl = make()
        eng = ctrl._engine

        def loop(n):
            if not sharded:
                eng.run_closed_loop(n, stream=stream)  # n complete iterations, one sync at the end
            else:
                ctrl.run_closed_loop_sharded(n)

        pos = [0]  # iterations done in the current episode

        def restart():
            ctrl.restart_episode(x_init)
            pos[0] = 0

        def run(n):  # n iterations of the driver's run, a new episode every `episode` iterations
            while n > 0:
                if pos[0] == episode:
                    restart()
                m = min(n, episode - pos[0])
                loop(m)
                pos[0] += m
                n -= m

        # one continuous run of the driver's loop: initialisation, warm-up, then the timed steps (BASELINE.md section 3:
        # closed-loop iterations after the warm-ups); a new episode begins every `episode` iterations of that run
        restart()
        run(8)  # initialisation, not warm-up: the first launches load the code objects (milliseconds)
        barrier()
        run(max(1, args.warmup))
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        if sharded:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        idx_timed = int(eng.stats.idx_after)

        def phase(n_skip, n, reps=1):  # wall time per iteration of iterations [n_skip, n_skip + n) of an episode
            ts = []                      # (one call of n iterations, its fixed cost included; median of `reps` episodes)
            for _ in range(reps):
                restart()
                if n_skip:
                    run(n_skip)
                barrier()
                t1 = time.perf_counter()
                run(n)
                barrier()
                ts.append((time.perf_counter() - t1) / n)
            return float(np.median(ts))
        phases = None
        if traverse:
            phases = {"traverse": phase(0, traverse, 9), "hold": phase(episode // 2, episode // 2)}

        # Kernel duration, live, with HIP events on the launch stream -- over WHOLE EPISODES of fixed length, whatever
        # --steps is, and over the same episodes (same noise counter) both times:
        # (a) the dominant kernel's launch-to-launch duration = growth of the region's duration when that (idempotent)
        #     kernel is launched twice wherever it is launched once, divided by the number of launches -- two events
        #     around the whole region, so no per-launch event overhead enters;
        # (b) per-launch event pairs with an empty-pair calibration (exclude dispatch): the cross-check and fallback.
        n_kernel_iters = 20 if c5 else max(2 * episode, 2000)
        it0 = int(eng.counters()["iterations"])

        def kernel_duration(repeats):
            eng.set_rollout_repeats(repeats)
            eng.set_iteration(it0)  # the same noise, hence the same run (launch for launch), every time
            restart()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0 = eng.counters()
            barrier()
            e0.record(stream)
            run(n_kernel_iters)
            e1.record(stream)
            barrier()
            c1 = eng.counters()
            eng.set_rollout_repeats(1)
            return (e0.elapsed_time(e1) * 1e-3, c1["rollout_launches"] - c0["rollout_launches"],
                    c1["finalize_launches"] - c0["finalize_launches"])

        k1 = kernel_duration(1)
        k2 = kernel_duration(2)
        eng.set_iteration(it0)
        restart()
        eng.enable_timing(True)
        run(min(n_kernel_iters, 2000))
        barrier()
        kms = eng.last_kernel_ms()
        eng.enable_timing(False)
        # (c) the same growth measured on graph replays (mppi_time_rollout_launch): the host enqueues four graph launches
        #     instead of thousands of kernel launches, so a slow or shared host core cannot enter the figure.  At the end
        #     of the run above the waypoint index rests (hold phase / frozen index), which is what a replay needs.
        graph = None
        under_profiler = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES",
                                                                                   "ROCPROFILER_REGISTER_FORCE_LOAD"))
        if not sharded and not args.no_graph_timing and not under_profiler:
            try:
                graph = eng.time_rollout_launch(20 if c5 else 500, 2, stream=stream)
            except pkg.MppiError:
                graph = None
        kms = dict(kms, graph=graph, ran=eng.rollout_kernel())
        return ctrl, eng, dt, idx_timed, k1, k2, n_kernel_iters, kms, phases

    try:
        ctrl, eng, dt, idx_timed, k1, k2, n_kernel_iters, kms, phases = measure()
    except pkg.MppiError as ex:
        # the peer-to-peer exchange lost a rank (every rank then fails within its timeout): measure again with the
        # one collective per iteration instead
        if not sharded or ex.code != pkg._capi.ERR_COMM or os.environ.get("MPPI_EXCHANGE") in ("rccl", "collective"):
            raise
        os.environ["MPPI_EXCHANGE"] = "rccl"  # (falls through to the torch.distributed collective if RCCL cannot be set up)
        barrier()
        ctrl, eng, dt, idx_timed, k1, k2, n_kernel_iters, kms, phases = measure()

    # host-in-the-loop latency: x0 from the host, u0 back to the host every iteration
    lat = None
    if not sharded and not c3 and not c4 and not c5:
        import contextlib
        import io
        state = eng.get_state()
        ts = []
        with contextlib.redirect_stdout(io.StringIO()):  # the controller prints at the path end, like the reference
            for _ in range(200):
                t1 = time.perf_counter()
                u0 = ctrl._calc_input_control(state)[0]
                ts.append(time.perf_counter() - t1)
                # the driver's plant on the host, DifferentialDrive.update_state (mppi_differential_drive.py:33-40)
                state = state + 0.1 * np.array([u0[0] * np.cos(state[2]), u0[0] * np.sin(state[2]), u0[1]])
        lat = float(np.median(ts))

    if rank == 0:
        units = K_global * T
        # ALGORITHMIC HBM bytes: 16 B per trajectory-step (two passes over the f32
        # noise: rollout, weighted reduce) + 8 B per trajectory (S out, S in).  The fused rollout kernel does BOTH
        # passes in one launch (the noise stays in registers), so one launch owns the whole figure.
        alg_bytes = 16.0 * K_local * T + 8.0 * K_local
        (t1x, l1, f1), (t2x, l2, f2) = k1, k2
        ev_pair = kms["rollout"] * 1e-3
        t_marg = (t2x - t1x) / max(l1, 1)
        period_per_launch = t1x / max(l1, 1)  # everything an iteration does, per rollout launch: an upper bound
        method = "marginal (region with the kernel launched twice minus the plain region, per launch)"
        t_roll = t_marg
        ok = (l2 == 2 * l1 and t2x > t1x and t_marg >= 0.5 * ev_pair and t_marg <= period_per_launch)
        if sharded and ctrl.exchange not in ("p2p", "rccl"):
            ok = False  # paced by the collective and the host: a repeated launch hides in the slack
        graph = kms.get("graph")
        t_graph = None if graph is None else 1e-6 * graph["rollout_us"]
        if t_graph is not None and t_graph >= 0.5 * ev_pair and (not ok or t_marg > 1.25 * t_graph):
            # The eager figure is the host's pace, not the kernel's (a slow or shared host core: it exceeds what one more
            # launch costs a stream the host is not part of by more than a quarter), or it was not usable at all: the
            # graph-replay figure stands in.  On a host that keeps the queue full the two agree within ~10 %, and the eager
            # one is kept because it is measured the way rocprofv3 sees the launches of the timed run.
            t_roll = t_graph
            ok = True
            method = ("marginal on graph replays (iterations captured with the kernel launched 3x minus as they are, per "
                      "extra launch; mppi_time_rollout_launch) -- the eager marginal figure (kernel_us_marginal) was paced by "
                      "the host on this box")
        if not ok:
            t_roll = ev_pair
            method = ("per-launch event pairs minus the empty-pair calibration (excludes dispatch) -- fallback: the "
                      "marginal measurement was not usable here (launch counts %d/%d, regions %.3f/%.3f ms)"
                      % (l1, l2, 1e3 * t1x, 1e3 * t2x))
        build_id = pkg.source_id()
        frac = alg_bytes / t_roll / 1e9 / HBM_PEAK_GBS if t_roll > 0 else None
        if frac is not None and not (0.0 < frac <= 1.0):  # never report more than the roof: say what was seen instead
            method += "; REJECTED (implied %.3g of the HBM peak)" % frac
            frac, t_roll = None, None
        pmc, pmc_src = stamped_profile("pmc", build_id)
        ran = kms["ran"]  # the instantiation the timed launches took, as rocprofv3 spells it
        layout = eng.counters()["rollout_layout"]
        # workgroups of one rollout launch: 16 waves with a sample each, or two (the dual layout: low bits == 1, three steps
        # per lane: == 3), or two in sequence on top (+4); the learned-dynamics kernel: 64 samples per workgroup
        per_wg = 64 if c5 else 16 * (2 if (layout & 3) in (1, 3) else 1) * (2 if layout & 4 else 1)
        wgs_launch = (K_local + per_wg - 1) // per_wg
        pk = pmc_kernel(pmc, ran, wgs_launch) if ran else None
        pk_why = pmc_src if pk is not None else (pmc_src if pmc is None else
                                                  f"{pmc_src} holds no counters of {ran} at {wgs_launch} workgroups per launch")
        traffic = None
        if pk is not None and "FETCH_SIZE" in pk["counters"] and "WRITE_SIZE" in pk["counters"]:
            # MI355X_MICROARCH.md (HBM): FETCH_SIZE counts half of a wide coalesced read on gfx950 (x2), WRITE_SIZE exact; KB
            traffic = (2.0 * pk["counters"]["FETCH_SIZE"] + pk["counters"]["WRITE_SIZE"]) * 1024.0
        kernel_name = (("k_rollout_mlp_h3 (operands split into two f16 numbers, TWO v_mfma_f32_32x32x16_f16 per product: MPPI_MLP_TERMS=2)"
                        if os.environ.get("MPPI_MLP_TERMS") == "2" else
                        "k_rollout_mlp_h3 (operands split into two f16 numbers, three v_mfma_f32_32x32x16_f16 per product)") if c5 else
                       "k_rollout_dual<float, diffdrive + circles, 2 samples per wave / 2 steps per lane>" if c3 else
                       ("k_rollout_tri<float, racecar, 2 samples per wave / 3 steps per lane>" if (layout & 3) == 3 else
                        "k_rollout_dual<float, racecar, 1 sample per wave / 2 steps per lane>") if c4 else
                       "k_rollout_fused<float, diffdrive, 1 chunk, single agent, PLAIN>")
        sequential = not (sharded or c4 or c5)
        phase = ("hold phase only (graph replays need a waypoint index at rest; the eager marginal figure beside it is "
                 "over whole episodes)" if (t_roll is not None and t_roll == t_graph and sequential) else
                 "whole episodes of the driver's run" + (", traversal included" if sequential else ""))
        roof = {"bound": "valu_issue", "kernel": kernel_name, "kernel_instantiation": ran,
                "achieved": None if t_roll is None else alg_bytes / t_roll / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": frac, "hbm_frac": frac,
                "traffic": traffic, "traffic_source": pk_why, "workgroups_per_launch": wgs_launch,
                "kernel_us_phase": phase,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_us": None if t_roll is None else 1e6 * t_roll, "kernel_us_method": method,
                "kernel_us_marginal": 1e6 * t_marg, "kernel_us_event_pair": 1e6 * ev_pair,
                "kernel_us_graph": None if graph is None else graph["rollout_us"],
                "graph_iteration_us": None if graph is None else graph["graph_iteration_us"],
                "measured_over": {"iterations": n_kernel_iters, "rollout_launches": l1, "finalize_launches": f1,
                                  "rollout_launches_per_iteration": l1 / n_kernel_iters,
                                  "region_ms": {"1x_rollout": 1e3 * t1x, "2x_rollout": 1e3 * t2x}},
                "event_pair_us": {k: 1e3 * v for k, v in kms.items() if k not in ("graph", "ran")},
                "note": "achieved/peak/frac are the ALGORITHMIC bytes of one launch over its live duration against the "
                        "HBM roof SURVEY.md section 8d nominates (hbm_frac = frac).  The roof that BINDS the launch is "
                        "VALU issue (bound): the noise is drawn in-kernel (Philox) and never touches HBM, so the PMC "
                        "traffic is a tenth of the algorithmic bytes; valu_issue_frac = VALU instructions per wave x "
                        "waves per SIMD x 4 clocks / kernel_us, instruction counts from the PMC pass of this build "
                        "(null when no such pass is committed) -- an instruction-count model: measured issue costs on "
                        "gfx950 are 2.5 / 4.3 / 8.3 cycles by instruction class (tools/issue_cost.hip, DESIGN.md section 6), "
                        "and dependent chains leave issue slots empty at four waves per SIMD.  Counter figures are "
                        "quoted only from a pass over THIS kernel instantiation at THIS launch size."}
        if c5:
            # the learned-dynamics rollout is bound by the matrix pipe: algorithmic flop of the network per launch (SURVEY.md
            # section 8d: 1 581 056 per trajectory-step) over the launch's duration, against the dense f16 peak.  The kernel
            # issues every product three times (f32-like accuracy from f16 operands), which `mfma_issue_frac` counts.
            f32_kernel = bool(os.environ.get("MPPI_MLP_F32"))
            flop = MLP_FLOP_PER_STEP * K_local * T
            peak = 157.3 if f32_kernel else MFMA_F16_PEAK_TFLOPS
            ach = None if t_roll is None else flop / t_roll / 1e12
            roof = {"bound": "mfma", "kernel": "k_rollout_mlp (f32-input MFMA)" if f32_kernel else kernel_name,
                    "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": None if ach is None else ach / peak,
                    "mfma_issue_frac": None if ach is None else (1.0 if f32_kernel else 2.0 if os.environ.get("MPPI_MLP_TERMS") == "2" else 3.0) * ach / peak,
                    "traffic": traffic, "traffic_source": pk_why, "algorithmic_flop_per_launch": flop,
                    "kernel_instantiation": ran,
                    # rocprofv3 --pmc pass of this command (profiles/): the matrix pipe's busy cycles over the launch's
                    # cycles on all 1024 SIMDs, and the MFMA operations the counters saw against the algorithmic count
                    # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 matrix pipes (32 per 32x32x16 MFMA);
                    # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (it equals 8 x the launch's duration x the shader clock)
                    "MfmaUtil": (None if pk is None or "SQ_VALU_MFMA_BUSY_CYCLES" not in pk["counters"] or
                                 not pk["counters"].get("GRBM_GUI_ACTIVE") else
                                 pk["counters"]["SQ_VALU_MFMA_BUSY_CYCLES"] / (pk["counters"]["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)),
                    "shader_clock_GHz_in_kernel": (None if pk is None or not pk["counters"].get("GRBM_GUI_ACTIVE") or not t_roll else
                                                   pk["counters"]["GRBM_GUI_ACTIVE"] / 8.0 / t_roll * 1e-9),
                    "mfma_flop_counted": (None if pk is None or "SQ_INSTS_VALU_MFMA_MOPS_F16" not in pk["counters"] else
                                          512.0 * pk["counters"]["SQ_INSTS_VALU_MFMA_MOPS_F16"]),
                    "pmc": None if pk is None else pk["counters"],
                    "kernel_us": None if t_roll is None else 1e6 * t_roll, "kernel_us_method": method,
                    "kernel_us_marginal": 1e6 * t_marg, "kernel_us_event_pair": 1e6 * ev_pair,
                    "kernel_us_graph": None if graph is None else graph["rollout_us"],
                    "measured_over": {"iterations": n_kernel_iters, "rollout_launches": l1},
                    "note": "achieved = algorithmic flop of the network per launch / the launch's duration; peak = dense "
                            "f16 MFMA (the f32-input MFMA's 157.3 TFLOP/s with MPPI_MLP_F32=1); mfma_issue_frac counts "
                            "the three MFMAs the kernel issues per product.  weights: " + weights_src}
        elif t_roll:
            roof["valu_issue_frac"] = None
            roof["valu_issue"] = {"source": pk_why}
            if pk is not None and "SQ_INSTS_VALU" in pk["counters"]:
                per_wave = pk["counters"]["SQ_INSTS_VALU"] / pk["counters"]["SQ_WAVES"]
                wps = pk["counters"]["SQ_WAVES"] / 1024.0  # 256 compute units x 4 SIMDs
                issue_us = per_wave * wps * 4.0 / SHADER_GHZ * 1e-3
                roof["valu_issue_frac"] = min(1.0, issue_us / (1e6 * t_roll))
                roof["valu_issue"] = {"valu_instructions_per_wave": per_wave, "waves_per_simd": wps, "issue_us": issue_us,
                                      "shader_clock_GHz": SHADER_GHZ, "source": pk_why, "vgprs": pk.get("vgprs")}
        out = {"metric": "trajectory-steps/sec (KxT/iter_time), " + ("diff-drive + learned MLP dynamics K=32768 T=50" if c5 else
                                                                     "diff-drive + 8 circular obstacles K=16384 T=50" if c3 else
                                                                     "race-car K=65536 T=75" if c4 else "diff-drive K=4096 T=50"),
               "value": units * args.steps / dt,
               "unit": "trajectory-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if (c4 or c5) else "weak",
               "vs_baseline": None,
               "dtype": ("f32 (network products as TWO f16 MFMAs on split operands: MPPI_MLP_TERMS=2, an opt-in that keeps u within "
                         "1e-4 RMSE but S only within 3e-3)" if os.environ.get("MPPI_MLP_TERMS") == "2" else
                         "f32 (network products as three f16 MFMAs on split operands)") if c5 else "f32",
               "data": "synthetic",
               "config": {"workload": ("BASELINE config 3: differential-drive + 8 static circular obstacles (mppi_differential_drive_obs), "
                                       "K=16384 x T=50 on one GPU, closed loop with the driver's plant on the device") if c3 else
                                      ("BASELINE config 5: differential-drive with learned residual dynamics (MLP 5-512-512-512-512-3 on "
                                       "the matrix cores), K=32768 x T=50 in total, K/N per GPU, FROZEN waypoint index (the "
                                       "sequential one is pinned at K <= 1024 by the tests), closed loop with the driver's plant "
                                       "on the device") if c5 else
                                      ("BASELINE config 4: race-car bicycle dynamics + 2 circular obstacles "
                                       "(mppi_race_car_obstacle defaults), K=65536 x T=75 in total, K/N per GPU, closed loop "
                                       "with the driver's plant on the device") if c4 else
                                      ("BASELINE config 2: differential-drive analytic dynamics, K=4096 x T=50 per GPU, "
                                       "reference __main__ parameters, closed loop with the driver's plant on the device"),
                          "K_per_gpu": K_local, "K_global": K_global, "T": T,
                          "waypoint_mode": ("frozen" if (c4 or c5) else os.environ.get("MPPI_BENCH_SHARD_MODE", "frozen") + " (K-sharded)")
                                           if (sharded or c4 or c5) else "sequential (reference-exact)",
                          "noise": "Philox4x32-10 in-kernel",
                          "timed_iterations": "closed-loop iterations %d..%d of the reference driver's run, which restarts "
                                              "from its initial state every %d iterations"
                                              % (8 + max(1, args.warmup), 8 + max(1, args.warmup) + args.steps, episode),
                          "waypoint_idx_at_end_of_timing": idx_timed,
                          "build_id": build_id,
                          "exchange": {"none": "none (one GPU)", "p2p": "peer-to-peer stores + flags inside k_finalize",
                                       "rccl": "one ncclAllGather per iteration enqueued by the library (mppi_comm_init)",
                                       "collective": "one all-gather per iteration (torch.distributed, RCCL)"}[ctrl.exchange]},
               "iter_latency_us": 1e6 * dt / args.steps,
               "phase_latency_us": None if phases is None else
                                   {"traverse (first %d iterations of an episode, one call, median of 9 episodes)" % traverse: 1e6 * phases["traverse"],
                                    "hold (second half of an episode)": 1e6 * phases["hold"]},
               "host_in_loop_latency_us": None if lat is None else 1e6 * lat,
               "roofline": roof}
        if world == 1 and c4:
            out["strong_scaling_bound"] = strong_scaling_bound_c4(1e-3 * out["ms_per_step"])
        if world == 1 and not c3 and not c4 and not c5 and not args.no_batched:
            out["batched_agents"] = batched_agents()
        if world == 1 and not c3 and not c4 and not c5 and not args.no_cpu_baseline:  # (the C restatement timed is config 2's)
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
