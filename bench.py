#!/usr/bin/env python3
"""Headline benchmark: trajectory-steps/s of the MPPI iteration (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

A "step" is one closed-loop MPPI iteration (sample -> rollout -> cost -> softmin weight -> reduce ->
filter -> shift, then the driver's plant advances the state) of BASELINE config 2: differential-drive,
K=4096 samples x T=50 horizon, fp32, reference `__main__` parameters
(controllers/mppi_differential_drive.py:400-410), synthetic straight-line path.  State, controls and the
Philox-keyed noise live on the GPU; nothing crosses PCIe inside the timed region.

The timed steps are the iterations of the reference driver's own run, episodes back to back: the robot starts at the
head of the 100-waypoint path with a fresh controller and the loop runs tSim = 1000 iterations
(mppi_differential_drive.py:396) -- some 25 of them traverse the path (the waypoint index moves, the search window is
SEARCH_IDX_LEN = 20 candidates long (:204) and the sequential index needs repair launches: ~60 us per iteration), the rest hold the
goal (window of one candidate: ~9 us).  A run that never leaves the hold phase would flatter the number, so the bench
restarts the episode every 1000 iterations of its run (initialisation + warm-up + timed steps; three small uploads,
timed when they fall into the timed region) -- with the default 2000 steps two traversals are inside the timed region --
and reports both phase latencies as well.  N > 1: one process
per GPU (torch.distributed / RCCL), every rank evaluates K=4096 of K_global = N*4096 samples and one
all-gather of {rho, eta, eta2, W[T,2]} per iteration merges the softmin (weak scaling).

Prints ONE JSON line (rank 0).  `roofline` is measured in a second pass of the same K steps with HIP
events bracketing every kernel launch on the launch stream; `cpu_baseline` is the plain-C oracle
(test infrastructure, oracle/mppi_oracle.c) timed on one host core, rank 0 at N=1 only.
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
EPISODE = 1000   # tSim of the reference driver (controllers/mppi_differential_drive.py:396)
TRAVERSE = 25    # iterations the robot needs from the head of the path to its goal (measured; reported separately)
X_INIT = np.zeros(3)  # init_x, :394


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


def cpu_baseline(budget_s=12.0):
    """The C restatement of the reference loop on ONE host core, closed loop, eps pre-generated."""
    from oracle import c_oracle, mppi_oracle, philox
    kw = config2_kwargs()
    o = c_oracle.DiffDriveC(**kw)
    pool = [philox.sample_epsilon(kw["sigma"], 1, i, K_SAMPLES, HORIZON) for i in range(4)]
    state = X_INIT.copy()
    o.iteration(state, pool[0])  # warm the caches
    o = c_oracle.DiffDriveC(**kw)
    n, spent = 0, 0.0
    while spent < budget_s:
        if n % EPISODE == 0:  # the same run as the GPU path times: a new episode of the driver every tSim iterations
            o = c_oracle.DiffDriveC(**kw)
            state = X_INIT.copy()
        t0 = time.perf_counter()
        out = o.iteration(state, pool[n % len(pool)])
        spent += time.perf_counter() - t0
        state = mppi_oracle.diffdrive_plant_step(state, out["u0_returned"], kw["delta_t"])
        n += 1
    return {"value": K_SAMPLES * HORIZON * n / spent, "unit": "trajectory-steps/s", "cores": 1, "kind": "port",
            "sample": f"the first {n} iterations of the same run (K=4096, T=50, episodes of {EPISODE}) in {spent:.1f} s "
                      "(oracle/mppi_oracle.c, scalar f64, noise pre-generated and excluded)",
            "ms_per_step": 1e3 * spent / n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    kw = config2_kwargs(K=K_SAMPLES * world)  # K_global; each rank evaluates K_SAMPLES of them
    stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def measure():
        ctrl = pkg.MPPIAlgorithms(**kw, precision="f32", device=local_rank, seed=2024, process_group=pg)
        eng = ctrl._engine

        def loop(n):
            if not sharded:
                eng.run_closed_loop(n, stream=stream)  # n complete iterations, one sync at the end
            else:
                ctrl.run_closed_loop_sharded(n)

        pos = [0]  # iterations done in the current episode

        def restart():
            ctrl.restart_episode(X_INIT)
            pos[0] = 0

        def run(n):  # n iterations of the driver's run, a new episode every EPISODE iterations
            while n > 0:
                if pos[0] == EPISODE:
                    restart()
                m = min(n, EPISODE - pos[0])
                loop(m)
                pos[0] += m
                n -= m

        # one continuous run of the driver's loop: initialisation, warm-up, then the timed steps (BASELINE.md section 3:
        # closed-loop iterations after the warm-ups); a new episode begins every EPISODE iterations of that run
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

        def phase(n_skip, n):  # wall time per iteration of iterations [n_skip, n_skip + n) of an episode
            restart()
            if n_skip:
                run(n_skip)
            barrier()
            t1 = time.perf_counter()
            run(n)
            barrier()
            return (time.perf_counter() - t1) / n
        phases = {"traverse": phase(0, TRAVERSE), "hold": phase(EPISODE // 2, EPISODE // 2)}

        # Kernel duration, measured live with HIP events on the launch stream over the same number of steps:
        # (a) the dominant kernel's launch-to-launch duration = growth of the iteration period when that
        #     (idempotent) kernel is launched twice per iteration -- two events around the whole region, so no
        #     per-launch event overhead enters; (b) per-launch event pairs with an empty-pair calibration.
        def timed_region(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            e0.record(stream)
            run(n)
            e1.record(stream)
            barrier()
            return e0.elapsed_time(e1) * 1e-3 / n

        period_1x = timed_region(args.steps)
        eng.set_rollout_repeats(2)
        period_2x = timed_region(args.steps)
        eng.set_rollout_repeats(1)
        eng.enable_timing(True)
        run(min(args.steps, 2000))
        barrier()
        kms = eng.last_kernel_ms()
        eng.enable_timing(False)
        return ctrl, eng, dt, idx_timed, period_1x, period_2x, kms, phases

    try:
        ctrl, eng, dt, idx_timed, period_1x, period_2x, kms, phases = measure()
    except pkg.MppiError as ex:
        # the peer-to-peer exchange lost a rank (every rank then fails within its timeout): measure again with the
        # one collective per iteration instead
        if not sharded or ex.code != pkg._capi.ERR_COMM or os.environ.get("MPPI_EXCHANGE") == "collective":
            raise
        os.environ["MPPI_EXCHANGE"] = "collective"
        barrier()
        ctrl, eng, dt, idx_timed, period_1x, period_2x, kms, phases = measure()
    t_rollout = max(period_2x - period_1x, 1e-9)

    # host-in-the-loop latency: x0 from the host, u0 back to the host every iteration
    lat = None
    if not sharded:
        from oracle import mppi_oracle
        import contextlib
        import io
        state = eng.get_state()
        ts = []
        with contextlib.redirect_stdout(io.StringIO()):  # the controller prints at the path end, like the reference
            for _ in range(200):
                t1 = time.perf_counter()
                u0 = ctrl._calc_input_control(state)[0]
                ts.append(time.perf_counter() - t1)
                state = mppi_oracle.diffdrive_plant_step(state, u0, kw["delta_t"])
        lat = float(np.median(ts))

    if rank == 0:
        units = K_SAMPLES * world * HORIZON
        # ALGORITHMIC HBM bytes: 16 B per trajectory-step (two passes over the f32
        # noise: rollout, weighted reduce) + 8 B per trajectory (S out, S in).  k_rollout_fused does BOTH
        # passes in one launch (the noise stays in registers), so one launch owns the whole figure.
        alg_bytes = 16.0 * K_SAMPLES * HORIZON + 8.0 * K_SAMPLES
        t_roll = t_rollout
        if sharded and ctrl.exchange != "p2p":  # paced by the collective and the host: a repeated launch hides in
            t_roll = max(kms["rollout"] * 1e-3, 1e-9)  # the slack, use the calibrated per-launch event pairs instead
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc_file):  # rocprofv3 --pmc passes of this same command (see profiles/README.md)
            traffic = json.load(open(pmc_file)).get("k_rollout_fused_hbm_bytes_per_launch")
        roof = {"bound": "hbm", "kernel": "k_rollout_fused<float, diffdrive, 1 chunk, single agent, PLAIN>",
                "achieved": alg_bytes / t_roll / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / t_roll / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_us": 1e6 * t_rollout,
                "period_us": {"1x_rollout": 1e6 * period_1x, "2x_rollout": 1e6 * period_2x},
                "event_pair_us": {k: 1e3 * v for k, v in kms.items()},
                "iteration_achieved_GBs": alg_bytes / period_1x / 1e9,
                "note": "kernel_us = launch-to-launch duration of k_rollout_fused on its stream (period with the kernel "
                        "launched twice per iteration minus the normal period, HIP events around the whole region); "
                        "event_pair_us = per-launch event pairs minus the empty-pair calibration (excludes dispatch). "
                        "Averaged over whole episodes, i.e. including the repair launches of the traversal phase. "
                        "Noise is drawn in-kernel (Philox) and never touches HBM, so PMC traffic is far below the "
                        "algorithmic bytes: the launch is bound by VALU issue, not by HBM (valu_issue)"}
        valu_file = os.path.join(ROOT, "profiles", "r01_pmc_valu.json")
        if os.path.exists(valu_file):  # PMC instruction counters of this same command (profiles/README.md)
            v = next(x for k, x in json.load(open(valu_file)).items() if k.startswith("config 2"))
            roof["valu_issue"] = {"valu_instructions_per_wave": v["per_wave"]["VALU"], "waves_per_simd": v["waves_per_simd"],
                                  "issue_us": v["valu_issue_us"], "of_event_pair_rollout_us": 1e3 * kms["rollout"],
                                  "source": "profiles/r01_pmc_valu.json (hold-phase launches)"}
        out = {"metric": "trajectory-steps/sec (KxT/iter_time), diff-drive K=4096 T=50", "value": units * args.steps / dt,
               "unit": "trajectory-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE config 2: differential-drive analytic dynamics, K=4096 x T=50 per GPU, "
                                      "reference __main__ parameters, closed loop with the driver's plant on the device",
                          "K_per_gpu": K_SAMPLES, "K_global": K_SAMPLES * world, "T": HORIZON,
                          "waypoint_mode": "frozen (K-sharded)" if sharded else "sequential (reference-exact)",
                          "noise": "Philox4x32-10 in-kernel",
                          "timed_iterations": "closed-loop iterations %d..%d of the reference driver's run, which restarts "
                                              "from its initial state every %d iterations (tSim): path traversal + "
                                              "holding the goal" % (8 + max(1, args.warmup), 8 + max(1, args.warmup) + args.steps,
                                                                    EPISODE),
                          "exchange": {"none": "none (one GPU)", "p2p": "peer-to-peer stores + flags inside k_finalize",
                                       "collective": "one all-gather per iteration (RCCL)"}[ctrl.exchange]},
               "iter_latency_us": 1e6 * dt / args.steps,
               "phase_latency_us": {"traverse (first %d iterations of an episode)" % TRAVERSE: 1e6 * phases["traverse"],
                                    "hold (second half of an episode)": 1e6 * phases["hold"]},
               "host_in_loop_latency_us": None if lat is None else 1e6 * lat,
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
