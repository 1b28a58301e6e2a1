"""GPU: engine-level checks at BASELINE sizes -- sampler vs its NumPy restatement, the full-size
config against the C oracle, and size-independent properties (shard invariance, softmin shift
invariance, permutation invariance of the weighted reduce)."""
import os

import numpy as np
import pytest

import golden_util as gu
from oracle import c_oracle, mppi_oracle, philox

pytestmark = pytest.mark.gpu


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, float) - np.asarray(b, float)) ** 2)))


def dd_kwargs(K, T, **over):
    kw = dict(delta_t=0.1, ref_path=mppi_oracle.generate_point_trajectory((0.0, 0.0), (10.0, -5.0), 100),
              max_speed=5.0, max_omega=3.14, num_samples_K=K, num_horizons_T=T, param_exploration=0.05,
              param_lambda=1.0, param_alpha=0.2, sigma=np.array([[0.1, 0.0], [0.0, 0.01]]),
              stage_cost_weight=np.array([5.0, 5.0, 10.0]), terminal_cost_weight=np.array([5.0, 5.0, 10.0]),
              visualize_optimal_traj=False, visualze_sampled_trajs=False)
    kw.update(over)
    return kw


def test_sampler_matches_numpy_restatement():
    import dnn_mppi_mpc_amd as pkg
    sigma = np.array([[0.5, 0.1], [0.1, 0.2]])
    c = pkg.MPPIAlgorithms(**dd_kwargs(4096, 50, sigma=sigma), seed=0x1234567890ABCDEF)
    for it in (0, 7):
        got = c._engine.sample_epsilon(it).cpu().numpy()
        want = philox.sample_epsilon(sigma, 0x1234567890ABCDEF, it, 4096, 50)
        # device: fp32 log/sqrt/sincos; restatement: f64 rounded to f32 -> a few f32 ulps of |z| <= 6
        np.testing.assert_allclose(got, want, rtol=0, atol=4e-6)
    e = got.reshape(-1, 2).astype(np.float64)
    assert abs(e.mean(0)).max() < 5e-3
    np.testing.assert_allclose(np.cov(e.T), sigma, atol=6e-3)


def test_inkernel_sampler_equals_materialised_sampler():
    """eps = NULL (Philox inside the rollout/reduce kernels) and the same numbers passed as a tensor
    must give the same iteration."""
    import dnn_mppi_mpc_amd as pkg
    kw = dd_kwargs(1024, 50)
    a = pkg.MPPIAlgorithms(**kw, seed=99)
    b = pkg.MPPIAlgorithms(**kw, seed=99)
    x0 = np.array([0.3, -0.1, -0.4])
    for it in range(3):
        eps = b._engine.sample_epsilon(it)
        b._calc_epsilon = lambda *aa, _e=eps, **k: _e
        ua = a._calc_input_control(x0)[1].copy()
        ub = b._calc_input_control(x0)[1].copy()
        np.testing.assert_allclose(ua, ub, rtol=0, atol=1e-7)
        np.testing.assert_allclose(a.sample_costs(), b.sample_costs(), rtol=1e-6, atol=1e-6)  # (costs near zero: absolute)


@pytest.mark.parametrize("name", ["dd_c2_k4096_default", "dd_c2_k4096_moderate"])
def test_config2_full_size_against_c_oracle_and_reference(name):
    fx = gu.load(name)
    eps = gu.eps_of(fx)
    o = c_oracle.DiffDriveC(**fx["meta"])
    o.u_prev[:] = fx["u_prev_in"]
    o.prev_way_point_idx = int(fx["idx_before"])
    ref = o.iteration(fx["x0"], eps)
    import dnn_mppi_mpc_amd as pkg
    for precision, tol, stol in (("f64", 1e-8, 1e-9), ("f32", 1e-4, 3e-4)):
        c = pkg.MPPIAlgorithms(**fx["meta"], precision=precision)
        c.u_prev[:] = fx["u_prev_in"]
        c.prev_way_point_idx = int(fx["idx_before"])
        c._calc_epsilon = lambda *a, **k: eps
        u0, u, _, _ = c._calc_input_control(fx["x0"])
        S = c.sample_costs()
        np.testing.assert_allclose(S, ref["S"], rtol=stol, atol=stol)
        np.testing.assert_allclose(S, fx["S"], rtol=stol, atol=stol)
        assert int(np.argmin(S)) == int(np.argmin(fx["S"]))
        assert rmse(u, fx["u_returned"]) <= tol
        assert rmse(u, ref["u_returned"]) <= tol
        assert c.prev_way_point_idx == int(fx["idx_after"]) == ref["idx_after"]
        assert c.last_stats.rounds >= 1


def test_config3_obstacles_k16384_against_c_oracle():
    """BASELINE config 3: 8 circles, K=16384, T=50 (C oracle ~50 ms)."""
    rng = np.random.default_rng(1234)
    circles = [[2.0, 2.0, 0.4], [3.0, 3.5, 0.4]]
    while len(circles) < 8:
        x, y = rng.uniform(0.5, 4.5, 2)
        if x * x + y * y > 0.81:
            circles.append([float(x), float(y), 0.4])
    kw = dd_kwargs(16384, 50, ref_path=mppi_oracle.generate_point_trajectory((0.0, 0.0), (5.0, 5.0), 100),
                   param_exploration=0.05, param_lambda=10.0, param_alpha=0.98,
                   stage_cost_weight=10 * np.array([5.0, 6.0, 9.0]),
                   terminal_cost_weight=10 * np.array([5.0, 6.0, 9.0]),
                   obstacle_circles=np.array(circles), safety_margin_rate=0.8)
    eps = philox.sample_epsilon(kw["sigma"], 77, 0, 16384, 50)
    tt = np.arange(50)
    u_in = np.stack([1.0 + 0.3 * np.sin(0.2 * tt), 0.05 * np.cos(0.1 * tt)], axis=1)
    x0 = np.array([0.6, 0.5, 0.75])
    o = c_oracle.DiffDriveC(**kw)
    o.u_prev[:] = u_in
    ref = o.iteration(x0, eps)
    import dnn_mppi_mpc_amd as pkg
    for precision, tol in (("f64", 1e-8), ("f32", 1e-4)):
        c = pkg.MPPIAlgorithms(**kw, precision=precision)
        c.u_prev[:] = u_in
        c._calc_epsilon = lambda *a, **k: eps
        u = c._calc_input_control(x0)[1]
        S = c.sample_costs()
        if precision == "f64":
            np.testing.assert_array_equal(S > 1e9, ref["S"] > 1e9)
            np.testing.assert_allclose(S, ref["S"], rtol=1e-9, atol=1e-9)
        else:
            assert np.mean((S > 1e9) != (ref["S"] > 1e9)) < 1e-3  # circle-boundary flips in f32
        assert rmse(u, ref["u_returned"]) <= tol
        assert c.prev_way_point_idx == ref["idx_after"]
        # mppi_stats.n_collided: the samples whose cost carries a collision penalty, counted on the device
        assert c.last_stats.n_collided == int(np.sum(S > 1e9)) > 0
        assert c.last_stats.iter_us > 0.0


def test_stats_collision_count_across_layouts():
    """`mppi_stats.n_collided` through every record path: one sample per wave (K = 700), the index resolved in one launch
    and by speculation rounds, two samples per wave (K = 9000), the streaming kernel (frozen), the race car's outline test,
    the merge of > 512 records (K = 40000) and several agents; 0 without obstacles; the device closed loop reports the
    last iteration's."""
    import dnn_mppi_mpc_amd as pkg
    circles = np.array([[1.0, -0.4, 0.3], [2.5, -1.4, 0.4]])
    x0 = np.array([0.4, -0.1, -0.35])
    for K, mode in ((700, None), (9000, None), (9000, "frozen"), (40000, "frozen")):
        kw = dd_kwargs(K, 30, obstacle_circles=circles, safety_margin_rate=0.8, param_exploration=0.05)
        c = pkg.MPPIAlgorithms(**kw, precision="f32", seed=3, **({} if mode is None else {"waypoint_mode": mode}))
        tt = np.arange(30)
        c.u_prev[:] = np.stack([2.0 + 0.3 * np.sin(0.2 * tt), 0.05 * np.cos(0.1 * tt)], axis=1)
        c._calc_input_control(x0)
        n = int(np.sum(c.sample_costs() > 1e9))
        assert c.last_stats.n_collided == n and n > 0, (K, mode, c.last_stats.n_collided, n, c._engine.rollout_kernel())
        c._engine.set_state(x0)
        _, st = c._engine.run_closed_loop(3)
        assert st.n_collided == int(np.sum(c._engine.costs() > 1e9))
    c = pkg.MPPIAlgorithms(**dd_kwargs(700, 30), precision="f32", seed=3)
    c._calc_input_control(x0)
    assert c.last_stats.n_collided == 0
    assert c.last_stats.kernel_us == 0.0 and c.last_stats.iter_us > 0.0  # (kernel time only while timing is enabled)
    c._engine.enable_timing(True)
    c._engine.set_state(x0)
    _, st = c._engine.run_closed_loop(50)
    assert 0.5 < st.kernel_us < st.iter_us * 1.5, (st.kernel_us, st.iter_us)  # rollout + finalize by events: microseconds
    c._engine.enable_timing(False)
    lem = mppi_oracle.generate_lemniscate_racecar(100, 10.0)
    rc = pkg.MPPIRacecarController(ref_path=lem, horizon_step_T=75, number_of_samples_K=3000, obstacle_circles=np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]]),
                                   visualize_optimal_traj=False, visualze_sampled_trajs=False, precision="f32", seed=5)
    rc._calc_control_input(lem[0].astype(np.float64))  # (the driver's start: most samples cross a circle)
    n = int(np.sum(rc.sample_costs() >= 1e10))
    assert rc.last_stats.n_collided == n and n > 0


def test_config4_racecar_k65536_shard_against_c_oracle():
    """BASELINE config 4's per-GPU shard (K=65536/8, T=75) against the f32 C oracle."""
    lem = mppi_oracle.generate_lemniscate_racecar(100, 10.0)
    kw = dict(ref_path=lem, horizon_step_T=75, number_of_samples_K=8192,
              obstacle_circles=np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]]), collision_safety_margin_rat=1.5,
              visualize_optimal_traj=False, visualze_sampled_trajs=False)
    sigma = np.array([[0.5, 0.0], [0.0, 0.1]])
    eps = philox.sample_epsilon(sigma, 5, 0, 8192, 75)
    o = c_oracle.RaceCarC(**kw)
    ref = o.iteration(lem[0], eps)
    import dnn_mppi_mpc_amd as pkg
    c = pkg.MPPIRacecarController(**kw)
    c._calc_epsilon = lambda *a, **k: eps
    u = c._calc_control_input(lem[0])[1]
    S = c.sample_costs()
    # number of colliding steps per sample: an outline point within an f32 ulp of a circle may flip
    n_hit, n_hit_ref = np.rint(S / 1e10), np.rint(ref["S"] / 1e10)
    assert np.mean(n_hit != n_hit_ref) < 2e-3
    same = n_hit == n_hit_ref
    np.testing.assert_allclose(S[same], ref["S"][same], rtol=5e-5, atol=1e-2)
    assert rmse(u, ref["u_returned"]) <= 1e-4


def test_config4_racecar_full_k65536_on_one_gpu_against_c_oracle():
    """BASELINE config 4 at its full size on ONE handle (K=65536, T=75: what `bench.py --workload c4` runs at N=1, and the
    union of the 8 shards) against the f32 C oracle; plus the size-independent property that the 8 contiguous shards'
    softmin records merge to this handle's update (the exchange the ranks make, evaluated on the host in f64)."""
    lem = mppi_oracle.generate_lemniscate_racecar(100, 10.0)
    K, T = 65536, 75
    kw = dict(ref_path=lem, horizon_step_T=T, number_of_samples_K=K,
              obstacle_circles=np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]]), collision_safety_margin_rat=1.5,
              visualize_optimal_traj=False, visualze_sampled_trajs=False)
    sigma = np.array([[0.5, 0.0], [0.0, 0.1]])
    eps = philox.sample_epsilon(sigma, 5, 0, K, T)
    ref = c_oracle.RaceCarC(**kw).iteration(lem[0], eps)
    import dnn_mppi_mpc_amd as pkg
    c = pkg.MPPIRacecarController(**kw, seed=5)  # in-kernel Philox, iteration 0 = the tensor above
    u = c._calc_control_input(lem[0])[1].copy()
    S = c.sample_costs()
    n_hit, n_hit_ref = np.rint(S / 1e10), np.rint(ref["S"] / 1e10)
    assert np.mean(n_hit != n_hit_ref) < 2e-3  # an outline point within an f32 ulp of a circle may flip
    same = n_hit == n_hit_ref
    np.testing.assert_allclose(S[same], ref["S"][same], rtol=5e-5, atol=1e-2)
    assert rmse(u, ref["u_returned"]) <= 1e-4
    # the 8 shards' records {rho, eta, eta2, W} merged (f64) = the unsharded weighted noise
    beta = 1.0 / 50.0
    recs = [mppi_oracle.softmin_partial(S[r * 8192:(r + 1) * 8192], eps[r * 8192:(r + 1) * 8192], beta) for r in range(8)]
    rho, eta, ess, w_eps = mppi_oracle.merge_partials(recs, beta)
    wk = np.exp(-beta * (S - S.min()))
    np.testing.assert_allclose(w_eps, np.einsum("k,ktd->td", wk / wk.sum(), eps.astype(np.float64)), rtol=1e-9, atol=1e-12)
    assert abs(ess - c.last_stats.ess) <= 2e-3 * ess


def test_shard_invariance_two_shards_one_gpu():
    """K split over two handles (k_offset 0 / K/2) and merged through the split-step ABI gives the
    single-handle result: exploit/explore split and Philox are keyed by the global sample index."""
    import torch

    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    lem = mppi_oracle.generate_lemniscate_racecar(100, 10.0)
    base = dict(model=capi.MODEL_RACECAR, T=40, delta_t=0.05, u_max=[0.523, 2.0], wheel_base=2.5,
                param_exploration=0.1, param_lambda=50.0, param_alpha=0.9, sigma=[0.5, 0.0, 0.0, 0.1],
                stage_cost_weight=[50.0, 50.0, 1.0, 20.0], terminal_cost_weight=[50.0, 50.0, 1.0, 20.0],
                beta_mode=capi.BETA_INV_LAMBDA, accumulate_stage_cost=1, waypoint_mode=capi.WAYPOINT_FROZEN,
                search_window=200, wrap_yaw_stage=1, wrap_yaw_terminal=1, clamp_rollout=1, clamp_u_after_update=1,
                filter_mode=capi.FILTER_RACECAR, filter_window=10, obstacle_model=capi.OBSTACLE_NONE,
                collision_penalty=1e10, seed=4242, precision=capi.PREC_F64)
    K = 3000
    whole = pkg.Engine(K=K, **base)
    parts = [pkg.Engine(K=1400, K_global=K, k_offset=0, **base), pkg.Engine(K=1600, K_global=K, k_offset=1400, **base)]
    for e in [whole] + parts:
        e.set_ref_path(lem)
    x0 = lem[2].astype(np.float64)
    n = whole.partial_len()
    for it in range(3):
        u_ref, u0_ref, _ = whole.step(x0)
        gathered = torch.empty(2 * n, dtype=torch.float64, device="cuda")
        for r, e in enumerate(parts):
            e.step_begin(x0, None, gathered[r * n:(r + 1) * n])
        outs = [e.step_end(gathered, 2) for e in parts]
        for u, u0, _ in outs:
            np.testing.assert_allclose(u, u_ref, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(u0, u0_ref, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(np.concatenate([e.costs() for e in parts]), whole.costs(), rtol=1e-12)


def test_peer_exchange_three_shards_in_one_process():
    """The peer-to-peer exchange with the peers living in this process (`local_ptrs` of mppi_comm_connect): three
    handles own 1000 / 1000 / 1000 of K = 3000 samples, each runs its closed loop from its own host thread and stream,
    and every finalize kernel waits for the other two.  All three must end where the unsharded handle ends."""
    import threading

    import torch

    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    lem = mppi_oracle.generate_lemniscate_racecar(100, 10.0)
    base = dict(model=capi.MODEL_RACECAR, T=40, delta_t=0.05, u_max=[0.523, 2.0], wheel_base=2.5,
                param_exploration=0.1, param_lambda=50.0, param_alpha=0.9, sigma=[0.5, 0.0, 0.0, 0.1],
                stage_cost_weight=[50.0, 50.0, 1.0, 20.0], terminal_cost_weight=[50.0, 50.0, 1.0, 20.0],
                beta_mode=capi.BETA_INV_LAMBDA, accumulate_stage_cost=1, waypoint_mode=capi.WAYPOINT_FROZEN,
                search_window=200, wrap_yaw_stage=1, wrap_yaw_terminal=1, clamp_rollout=1, clamp_u_after_update=1,
                filter_mode=capi.FILTER_RACECAR, filter_window=10, obstacle_model=capi.OBSTACLE_NONE,
                collision_penalty=1e10, seed=4242, precision=capi.PREC_F64)
    K, n_it = 3000, 6
    whole = pkg.Engine(K=K, **base)
    parts = [pkg.Engine(K=1000, K_global=K, k_offset=1000 * r, **base) for r in range(3)]
    for e in [whole] + parts:
        e.set_ref_path(lem)
        e.set_state(lem[2].astype(np.float64))
    handles = [e.comm_export(3) for e in parts]
    ptrs = [e.comm_buffer() for e in parts]
    for r, e in enumerate(parts):
        e.comm_connect(r, handles, local_ptrs=[None if q == r else ptrs[q] for q in range(3)])
    streams = [torch.cuda.Stream() for _ in parts]
    errs = []

    def run(e, s):
        try:
            e.comm_probe(s)
            e.run_closed_loop(n_it, stream=s)
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)

    ths = [threading.Thread(target=run, args=(e, s)) for e, s in zip(parts, streams)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not errs, errs
    whole.run_closed_loop(n_it)
    for e in parts:
        np.testing.assert_allclose(e.get_u_prev(), whole.get_u_prev(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(e.get_state(), whole.get_state(), rtol=1e-9, atol=1e-12)
        e.comm_close()


def test_peer_connect_refuses_a_local_peer_it_cannot_reach():
    """`local_ptrs` entries are used as they are, so mppi_comm_connect checks them: a pointer that is not device memory
    (here: pinned host memory and a plain host address) is refused with MPPI_ERR_UNSUPPORTED instead of being stored
    into by the finalize kernel; a peer on another GPU gets peer access enabled (or the same refusal) -- one GPU here."""
    import ctypes

    import torch

    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    lem = mppi_oracle.generate_lemniscate_racecar(100, 10.0)
    e = pkg.Engine(model=capi.MODEL_RACECAR, K=256, K_global=512, k_offset=0, T=20, delta_t=0.05, u_max=[0.523, 2.0],
                   wheel_base=2.5, param_exploration=0.1, param_lambda=50.0, param_alpha=0.9, sigma=[0.5, 0.0, 0.0, 0.1],
                   stage_cost_weight=[50.0, 50.0, 1.0, 20.0], terminal_cost_weight=[50.0, 50.0, 1.0, 20.0],
                   beta_mode=capi.BETA_INV_LAMBDA, accumulate_stage_cost=1, waypoint_mode=capi.WAYPOINT_FROZEN,
                   search_window=200, filter_mode=capi.FILTER_RACECAR, filter_window=10, seed=1)
    e.set_ref_path(lem)
    handles = [e.comm_export(2), b"\0" * e.lib.mppi_comm_handle_bytes()]
    pinned = torch.empty(4096, dtype=torch.uint8).pin_memory()
    plain = ctypes.create_string_buffer(4096)
    for bad in (pinned.data_ptr(), ctypes.addressof(plain)):
        with pytest.raises(pkg.MppiError) as ei:
            e.comm_connect(0, handles, local_ptrs=[None, bad])
        assert ei.value.code == capi.ERR_UNSUPPORTED, ei.value
    dev = torch.empty(1 << 16, dtype=torch.uint8, device="cuda")  # device memory of the handle's own GPU: accepted
    e.comm_connect(0, handles, local_ptrs=[None, dev.data_ptr()])
    e.comm_close()


def test_peer_exchange_missing_rank_times_out(monkeypatch):
    """A rank that never arrives: the others give up after MPPI_EXCHANGE_TIMEOUT_MS with MPPI_ERR_COMM (no hang), the
    failure is sticky for the queued iterations, and mppi_comm_close gives the handle back."""
    import threading

    import torch

    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    monkeypatch.setenv("MPPI_EXCHANGE_TIMEOUT_MS", "200")
    base = dict(model=capi.MODEL_DIFFDRIVE, T=20, delta_t=0.1, u_max=[1.0, 1.0], param_exploration=0.1,
                param_lambda=1.0, param_alpha=0.5, sigma=[0.1, 0.0, 0.0, 0.1], stage_cost_weight=[1, 1, 1, 0],
                terminal_cost_weight=[1, 1, 1, 0], search_window=20, filter_window=10, clamp_rollout=1,
                waypoint_mode=capi.WAYPOINT_FROZEN)
    parts = [pkg.Engine(K=64, K_global=192, k_offset=64 * r, **base) for r in range(3)]
    ref = mppi_oracle.generate_point_trajectory((0, 0), (3, 1), 50)
    for e in parts:
        e.set_ref_path(ref)
        e.set_state(np.zeros(3))
    handles = [e.comm_export(3) for e in parts]
    ptrs = [e.comm_buffer() for e in parts]
    for r, e in enumerate(parts):
        e.comm_connect(r, handles, local_ptrs=[None if q == r else ptrs[q] for q in range(3)])
    codes = []

    def run(e, s):
        try:
            e.run_closed_loop(5, stream=s)
            codes.append(0)
        except pkg.MppiError as ex:
            codes.append(ex.code)

    ths = [threading.Thread(target=run, args=(e, torch.cuda.Stream())) for e in parts[:2]]  # rank 2 stays away
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=60)
    assert codes == [capi.ERR_COMM, capi.ERR_COMM]
    with pytest.raises(pkg.MppiError) as ex:  # sticky until the exchange is closed
        parts[0].run_closed_loop(1)
    assert ex.value.code == capi.ERR_COMM
    for e in parts:
        e.comm_close()
    import torch as _t
    n = parts[0].partial_len()  # and the handle works again through the split step
    gathered = _t.empty(3 * n, dtype=_t.float64, device="cuda")
    for r, e in enumerate(parts):
        e.step_begin(np.zeros(3), None, gathered[r * n:(r + 1) * n])
    us = [e.step_end(gathered, 3)[0] for e in parts]
    np.testing.assert_allclose(us[0], us[1], rtol=0, atol=0)
    np.testing.assert_allclose(us[0], us[2], rtol=0, atol=0)


def test_softmin_shift_and_permutation_invariance():
    """Adding a constant to every cost leaves the update unchanged; permuting the samples (with their
    noise rows) leaves the weighted reduce unchanged."""
    import torch

    import dnn_mppi_mpc_amd as pkg
    kw = dd_kwargs(2048, 50, param_exploration=0.5)
    eps = philox.sample_epsilon(kw["sigma"], 9, 0, 2048, 50)
    x0 = np.array([0.2, 0.0, -0.3])
    a = pkg.MPPIAlgorithms(**kw, precision="f64", waypoint_mode="frozen")
    a._calc_epsilon = lambda *aa, **k: eps
    ua = a._calc_input_control(x0)[1].copy()
    # all samples exploit (k < (1-expl)K) only for the first half: keep the permutation inside each half
    perm = np.concatenate([np.random.default_rng(0).permutation(1024), 1024 + np.random.default_rng(1).permutation(1024)])
    b = pkg.MPPIAlgorithms(**kw, precision="f64", waypoint_mode="frozen")
    b._calc_epsilon = lambda *aa, **k: eps[perm]
    ub = b._calc_input_control(x0)[1].copy()
    np.testing.assert_allclose(ub, ua, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.sort(b.sample_costs()), np.sort(a.sample_costs()), rtol=1e-12)
    w = a._compute_weight()
    assert abs(w.sum() - 1.0) < 1e-12 and (w >= 0).all()
    assert abs(a.last_stats.ess - 1.0 / np.sum(w ** 2)) < 1e-6 * a.last_stats.ess


def test_device_closed_loop_matches_host_loop():
    """mppi_run_closed_loop (plant + x0 call on the device, no host round trip) equals stepping from
    the host with the oracle's plant in between."""
    import dnn_mppi_mpc_amd as pkg
    kw = dd_kwargs(1024, 30)
    a = pkg.MPPIAlgorithms(**kw, precision="f64", seed=5)
    b = pkg.MPPIAlgorithms(**kw, precision="f64", seed=5)
    n = 25
    a._engine.set_state(np.zeros(3))
    trace, st = a._engine.run_closed_loop(n, trace=True)
    state = np.zeros(3)
    for it in range(n):
        u0 = b._calc_input_control(state)[0].copy()
        np.testing.assert_allclose(trace[it], u0, rtol=1e-9, atol=1e-12)
        state = mppi_oracle.diffdrive_plant_step(state, u0, kw["delta_t"])
    np.testing.assert_allclose(a._engine.get_state(), state, rtol=1e-9, atol=1e-12)
    assert st.idx_after == b.prev_way_point_idx
    assert st.iteration == n


def test_restart_episode_reproduces_a_fresh_controller():
    """`restart_episode` (what bench.py calls every tSim iterations): nominal controls, waypoint index and plant
    state go back to the driver's initial condition -- with the noise counter rewound too, the run that follows is
    the fresh controller's run bit for bit, traversal of the path (index moves, repair launches) included."""
    import dnn_mppi_mpc_amd as pkg
    kw = dd_kwargs(1024, 30)
    n = 40
    fresh = pkg.MPPIAlgorithms(**kw, precision="f32", seed=9)
    fresh._engine.set_state(np.zeros(3))
    tr_a, st_a = fresh._engine.run_closed_loop(n, trace=True)
    used = pkg.MPPIAlgorithms(**kw, precision="f32", seed=9)
    used._engine.set_state(np.array([0.3, -0.2, 0.1]))
    used._engine.run_closed_loop(60)
    assert used._engine.get_waypoint_idx() > 0 and np.abs(used._engine.get_u_prev()).max() > 0
    used.restart_episode(np.zeros(3))
    assert used.prev_way_point_idx == 0 and not used.u_prev.any()
    used._engine.set_iteration(0)
    tr_b, st_b = used._engine.run_closed_loop(n, trace=True)
    np.testing.assert_array_equal(tr_a, tr_b)
    np.testing.assert_array_equal(fresh._engine.get_state(), used._engine.get_state())
    assert st_a.idx_after == st_b.idx_after and st_a.idx_after > 0


def test_specialised_closed_loop_kernels_equal_the_general_ones():
    """The closed loop of the diff-drive NumPy controller runs the PLAIN instantiations of k_rollout_fused and
    k_finalize (run-time switches as constants); asking for the u0 trace selects the general k_finalize, injected noise
    the general rollout.  Same run, bit for bit."""
    import dnn_mppi_mpc_amd as pkg
    kw = dd_kwargs(2048, 40)
    a = pkg.MPPIAlgorithms(**kw, precision="f32", seed=21)
    b = pkg.MPPIAlgorithms(**kw, precision="f32", seed=21)
    for c in (a, b):
        c._engine.set_state(np.array([0.1, -0.05, 0.2]))
    n = 60  # traversal of the path and the first iterations at its end
    a._engine.run_closed_loop(n)                       # PLAIN rollout + PLAIN finalize
    tr, st_b = b._engine.run_closed_loop(n, trace=True)  # PLAIN rollout + general finalize
    np.testing.assert_array_equal(a._engine.get_state(), b._engine.get_state())
    np.testing.assert_array_equal(a._engine.get_u_prev(), b._engine.get_u_prev())
    assert a._engine.get_waypoint_idx() == b._engine.get_waypoint_idx() == st_b.idx_after
    # general rollout (noise handed over as a tensor) against the PLAIN one (drawn in the kernel), one iteration
    x0 = np.array([0.3, -0.1, -0.4])
    c1 = pkg.MPPIAlgorithms(**kw, precision="f32", seed=5)
    c2 = pkg.MPPIAlgorithms(**kw, precision="f32", seed=5)
    eps = c2._engine.sample_epsilon(0)
    c2._calc_epsilon = lambda *aa, _e=eps, **k: _e
    u1 = c1._calc_input_control(x0)[1].copy()
    u2 = c2._calc_input_control(x0)[1].copy()
    # (the instantiations contract their multiply-adds differently here and there: last-bit differences)
    np.testing.assert_allclose(c1.sample_costs(), c2.sample_costs(), rtol=1e-6, atol=0)
    np.testing.assert_allclose(u1, u2, rtol=0, atol=1e-6)


def test_fused_and_unfused_paths_agree(monkeypatch):
    """T <= 128 runs rollout+softmin partial in one launch; MPPI_FORCE_UNFUSED=1 selects the separate
    k_rollout / k_reduce launches (the only path for longer horizons).  Same iteration either way."""
    import dnn_mppi_mpc_amd as pkg
    kw = dd_kwargs(1000, 50, param_exploration=0.3)
    eps = philox.sample_epsilon(kw["sigma"], 17, 0, 1000, 50)
    x0 = np.array([1.0, -0.6, -0.2])
    outs = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("MPPI_FORCE_UNFUSED", "1")
        c = pkg.MPPIAlgorithms(**kw, precision="f64")
        c._calc_epsilon = lambda *a, **k: eps
        c.prev_way_point_idx = 3
        u = c._calc_input_control(x0)[1].copy()
        outs.append((u, c.sample_costs(), c.prev_way_point_idx, c.last_stats.rounds))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-13)
    assert outs[0][2] == outs[1][2]
    # the fused kernels resolve the moving waypoint index in one launch, the separate ones by speculation rounds
    assert outs[0][3] == 1 and outs[1][3] >= 1


@pytest.mark.parametrize("T", [64, 65, 128, 150])
def test_long_horizons_against_oracle(T):
    """Horizon chunking: 64 steps per wave pass (T=65, 128: two chunks fused; T=150: three, unfused)."""
    import dnn_mppi_mpc_amd as pkg
    kw = dd_kwargs(300, T, param_exploration=0.2, delta_t=0.05)
    eps = philox.sample_epsilon(kw["sigma"], 23, 0, 300, T)
    x0 = np.array([0.5, -0.2, -0.5])
    ref = mppi_oracle.DiffDriveOracle(**kw).iteration(x0, eps.astype(np.float64))
    c = pkg.MPPIAlgorithms(**kw, precision="f64")
    c._calc_epsilon = lambda *a, **k: eps
    u = c._calc_input_control(x0)[1]
    np.testing.assert_allclose(c.sample_costs(), ref["S"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(u, ref["u_returned"], rtol=1e-8, atol=1e-10)
    assert c.prev_way_point_idx == ref["idx_after"]


def _mlp_case(K, T, seed, **over):
    kw = dd_kwargs(K, T, param_exploration=0.05, **over)
    w = mppi_oracle.random_mlp_weights(seed)
    eps = philox.sample_epsilon(kw["sigma"], 31 + seed, 0, K, T)
    return kw, w, eps


@pytest.mark.parametrize("waypoint_mode", ["sequential", "frozen"])
def test_learned_dynamics_mfma_rollout_against_oracle(waypoint_mode):
    """Config 5 (scaled down so the f64 NumPy oracle takes seconds): rollout through the residual MLP on the
    f32 matrix cores vs the f64 restatement.  Tolerance: the north-star 1e-4 RMSE on u; S to 1e-3 relative
    (f32 MFMA chains through 3 x 512-wide layers and 30 recurrent steps)."""
    import dnn_mppi_mpc_amd as pkg
    K, T = 1500, 30
    kw, w, eps = _mlp_case(K, T, 1)
    x0 = np.array([0.4, -0.1, -0.35])
    tt = np.arange(T)
    u_in = np.stack([1.2 + 0.3 * np.sin(0.2 * tt), 0.05 * np.cos(0.1 * tt)], axis=1)
    o = mppi_oracle.DiffDriveMlpOracle(**kw, mlp_weights=w)
    o.u_prev[:] = u_in
    if waypoint_mode == "frozen":  # frozen index: every call searches from the x0 index (race-car semantics)
        p0 = o.nearest_waypoint(x0[0], x0[1], 0)
        v = o.clamp(np.where((np.arange(K) < mppi_oracle.exploit_threshold(kw["param_exploration"], K))[:, None, None],
                             u_in[None] + eps, eps.astype(np.float64)))
        X = o.rollout(x0, v)
        R = o.ref_path
        win = R[p0:p0 + 20]
        xT, yT, yawT = X[:, -1, 0], X[:, -1, 1], X[:, -1, 2]
        i = p0 + np.argmin((xT[:, None] - win[:, 0]) ** 2 + (yT[:, None] - win[:, 1]) ** 2, axis=1)
        ws, wt = o.stage_cost_weight, o.terminal_cost_weight
        q = u_in[T - 1] @ np.linalg.inv(o.Sigma)
        S_ref = ((ws[0] + wt[0]) * (xT - R[i, 0]) ** 2 + (ws[1] + wt[1]) * (yT - R[i, 1]) ** 2
                 + (ws[2] + wt[2]) * (yawT - R[i, 2]) ** 2 + o.param_gamma * (q[0] * v[:, -1, 0] + q[1] * v[:, -1, 1]))
        ref = None
    else:
        ref = o.iteration(x0, eps.astype(np.float64))
        S_ref = ref["S"]
    c = pkg.MPPIAlgorithms(**kw, learned_dynamics=w, waypoint_mode=waypoint_mode)
    c.u_prev[:] = u_in
    c._calc_epsilon = lambda *a, **k: eps
    u = c._calc_input_control(x0)[1]
    S = c.sample_costs()
    np.testing.assert_allclose(S, S_ref, rtol=1e-3, atol=1e-3)
    if ref is not None:
        assert rmse(u, ref["u_returned"]) <= 1e-4
        assert c.prev_way_point_idx == ref["idx_after"]


def test_learned_dynamics_visualisation_rollouts():
    """`visualze_sampled_trajs=True` with the learned model: the reference's visualisation loop (:144-159 -- step t driven
    by control t-1, always clamped, u clamped in place) with the transition swapped, against the oracle's
    `viz_trajectories` (f64).  States after T recurrent steps through the network: 1e-3."""
    import dnn_mppi_mpc_amd as pkg
    K, T = 200, 20
    kw, w, eps = _mlp_case(K, T, 2, visualize_optimal_traj=True, visualze_sampled_trajs=True, max_speed=1.5)
    x0 = np.array([0.2, 0.1, -0.2])
    tt = np.arange(T)
    u_in = np.stack([1.3 + 0.4 * np.sin(0.3 * tt), 0.1 * np.cos(0.2 * tt)], axis=1)  # (clamping active in places)
    o = mppi_oracle.DiffDriveMlpOracle(**kw, mlp_weights=w)
    o.u_prev[:] = u_in
    ref = o.iteration(x0, eps.astype(np.float64))
    opt_ref, smp_ref = o.viz_trajectories(x0, ref["u_pre_shift"], ref["v"])
    c = pkg.MPPIAlgorithms(**kw, learned_dynamics=w)
    c.u_prev[:] = u_in
    c._calc_epsilon = lambda *a, **k: eps
    u0, u, opt, smp = c._calc_input_control(x0)
    assert rmse(u, ref["u_returned"]) <= 1e-4
    assert opt.shape == (T, 3) and smp.shape == (K, T, 3)
    np.testing.assert_allclose(opt, opt_ref, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(smp, smp_ref, rtol=1e-3, atol=1e-3)
    # only the nominal sequence's trajectory: the samples' workgroups are left out of the launch
    opt_only, none = c._engine.rollout_viz(True, False)
    assert none is None
    np.testing.assert_allclose(opt_only.cpu().numpy(), opt_ref, rtol=1e-3, atol=1e-3)


def test_learned_dynamics_state_transition_stage_method():
    """`_state_transition(x_t, v_t)` with a learned model loaded: x + dt (f(x, v) + MLP([x, v]))
    (test/bullet_differential_drive_dnn.py:79-92) for a single call and for a ragged batch, against the f64 forward."""
    import dnn_mppi_mpc_amd as pkg
    kw, w, _ = _mlp_case(64, 20, 3)
    c = pkg.MPPIAlgorithms(**kw, learned_dynamics=w)
    rng = np.random.default_rng(8)
    for n in (1, 5, 64, 131):
        x = rng.normal(0, 1.0, (n, 3))
        v = np.column_stack([rng.uniform(-2, 2, n), rng.uniform(-1, 1, n)])
        r = mppi_oracle.mlp_forward(w, np.concatenate([x, v], axis=1))
        f = np.stack([v[:, 0] * np.cos(x[:, 2]), v[:, 0] * np.sin(x[:, 2]), v[:, 1]], axis=1)
        want = x + kw["delta_t"] * (f + r)
        np.testing.assert_allclose(c._state_transition(x, v), want, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c._state_transition(x[0], v[0]), want[0], rtol=1e-5, atol=1e-5)  # the reference's call shape


@pytest.mark.parametrize("obstacles", [0, 5])
def test_learned_dynamics_zero_residual_equals_analytic_kernel(obstacles):
    """out_layer = 0: the MFMA rollout must reproduce the analytic scan kernel (same costs, same update), with
    and without circular obstacles (the collision test of k_rollout_mlp reads the same in-register table)."""
    import dnn_mppi_mpc_amd as pkg
    kw, w, eps = _mlp_case(1024, 40, 2)
    x0 = np.array([0.2, 0.1, -0.5])
    tt = np.arange(40)
    u_in = np.stack([1.0 + 0.2 * np.sin(0.2 * tt), 0.03 * np.cos(0.1 * tt)], axis=1)
    if obstacles:  # the reference tests the state after the LAST step only (:130): circles around where the nominal rollout ends
        xe = x0.copy()
        for t in range(40):
            xe = mppi_oracle.diffdrive_plant_step(xe, u_in[t], kw["delta_t"])
        rng = np.random.default_rng(3)
        kw.update(obstacle_circles=np.column_stack([xe[0] + rng.uniform(-0.6, 0.6, obstacles), xe[1] + rng.uniform(-0.6, 0.6, obstacles),
                                                    rng.uniform(0.05, 0.2, obstacles)]), safety_margin_rate=0.8)
    w["out_layer.weight"][:] = 0
    w["out_layer.bias"][:] = 0
    a = pkg.MPPIAlgorithms(**kw)
    b = pkg.MPPIAlgorithms(**kw, learned_dynamics=w)
    a.u_prev[:] = u_in
    b.u_prev[:] = u_in
    for c in (a, b):
        c._calc_epsilon = lambda *aa, **k: eps
    ua = a._calc_input_control(x0)[1].copy()
    ub = b._calc_input_control(x0)[1].copy()
    hit = a.sample_costs() > 1e9
    if obstacles:
        assert 0 < hit.sum() < 1024
    # (a terminal pose within f32 rounding of a circle may flip between the two kernels' dynamics arithmetic)
    assert np.count_nonzero((b.sample_costs() > 1e9) != hit) <= 2
    same = (b.sample_costs() > 1e9) == hit
    np.testing.assert_allclose(b.sample_costs()[same & ~hit], a.sample_costs()[same & ~hit], rtol=2e-4, atol=2e-4)
    if np.array_equal(b.sample_costs() > 1e9, hit):
        assert rmse(ua, ub) <= 1e-4
    assert a.prev_way_point_idx == b.prev_way_point_idx


def test_learned_dynamics_with_standard_scalers():
    """The StandardScaler statistics of the reference's training pipeline (SURVEY.md App. C values) folded into
    the first/last Linear: kernel output == oracle with explicitly scaled inputs/outputs."""
    import dnn_mppi_mpc_amd as pkg
    kw, w, eps = _mlp_case(512, 20, 4)
    sc = dict(in_mean=np.array([4.390025834218122, -0.1261226463393075, -0.08013703690080425, 0.3585598113405002, -0.03134529264047761]),
              in_scale=np.array([5.586594593004116, 3.6412154342302943, 1.0597377625187672, 1.0238108433208775, 1.8363768322750769]),
              out_mean=np.array([-0.5611190600527631, 0.02943515716319776, -0.0151000663932475]) * 0.05,
              out_scale=np.array([5.701096415279686, 3.590083579774014, 0.9961674472612329]) * 0.05)

    class ScaledOracle(mppi_oracle.DiffDriveMlpOracle):
        def rollout(self, x0, v):
            K, T = v.shape[:2]
            X = np.empty((K, T, 3))
            s = np.tile(np.asarray(x0, np.float64), (K, 1))
            for t in range(T):
                z = (np.concatenate([s, v[:, t]], axis=1) - sc["in_mean"]) / sc["in_scale"]
                r = mppi_oracle.mlp_forward(self.mlp_weights, z) * sc["out_scale"] + sc["out_mean"]
                f = np.stack([v[:, t, 0] * np.cos(s[:, 2]), v[:, t, 0] * np.sin(s[:, 2]), v[:, t, 1]], axis=1)
                s = s + self.delta_t * (f + r)
                X[:, t] = s
            return X

    x0 = np.array([0.3, 0.2, 0.1])
    ref = ScaledOracle(**kw, mlp_weights=w).iteration(x0, eps.astype(np.float64))
    c = pkg.MPPIAlgorithms(**kw, learned_dynamics=w, learned_scalers=sc)
    c._calc_epsilon = lambda *a, **k: eps
    u = c._calc_input_control(x0)[1]
    np.testing.assert_allclose(c.sample_costs(), ref["S"], rtol=1e-3, atol=1e-3)
    assert rmse(u, ref["u_returned"]) <= 1e-4


@pytest.mark.parametrize("dual", ["0", "1"])
@pytest.mark.parametrize("model", ["diff", "race"])
def test_batched_agents_equal_separate_handles(monkeypatch, dual, model):
    """`n_agents` problems in one handle (one launch per stage, agents as a grid dimension) against the same problems
    in separate single-agent handles with `noise_stream` = the agent: identical closed loops."""
    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    monkeypatch.setenv("MPPI_DUAL", dual)
    B, K, n_it = 5, 300, 6
    if model == "diff":
        ref = mppi_oracle.generate_point_trajectory((0.0, 0.0), (10.0, -5.0), 100)
        base = dict(model=capi.MODEL_DIFFDRIVE, T=50, delta_t=0.1, u_max=[5.0, 3.14], param_exploration=0.05,
                    param_lambda=1.0, param_alpha=0.2, sigma=[0.1, 0.0, 0.0, 0.01], stage_cost_weight=[5, 5, 10, 0],
                    terminal_cost_weight=[5, 5, 10, 0], search_window=20, filter_window=10, clamp_rollout=1,
                    waypoint_mode=capi.WAYPOINT_FROZEN, obstacle_model=capi.OBSTACLE_CIRCLE, safety_margin=0.8,
                    collision_penalty=1e10, seed=99, precision=capi.PREC_F64)
        obstacles = np.array([[2.0, -1.0, 0.4], [5.0, -2.5, 0.5]])
        x0 = np.stack([[0.1 * a, -0.05 * a, 0.2 * a - 0.3] for a in range(B)])
    else:
        ref = mppi_oracle.generate_lemniscate_racecar(100, 10.0)
        base = dict(model=capi.MODEL_RACECAR, T=40, delta_t=0.05, u_max=[0.523, 2.0], wheel_base=2.5, param_exploration=0.1,
                    param_lambda=50.0, param_alpha=0.9, sigma=[0.5, 0.0, 0.0, 0.1], stage_cost_weight=[50.0, 50.0, 1.0, 20.0],
                    terminal_cost_weight=[50.0, 50.0, 1.0, 20.0], beta_mode=capi.BETA_INV_LAMBDA, accumulate_stage_cost=1,
                    waypoint_mode=capi.WAYPOINT_FROZEN, search_window=200, wrap_yaw_stage=1, wrap_yaw_terminal=1,
                    clamp_rollout=1, clamp_u_after_update=1, filter_mode=capi.FILTER_RACECAR, filter_window=10,
                    obstacle_model=capi.OBSTACLE_OUTLINE, safety_margin=1.5, vehicle_w=3.0, vehicle_l=4.0,
                    collision_penalty=1e10, seed=7, precision=capi.PREC_F32)
        obstacles = np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]])
        x0 = np.stack([ref[2 + 3 * a].astype(np.float64) for a in range(B)])
    batch = pkg.Engine(K=K, n_agents=B, **base)
    batch.set_ref_path(ref)
    batch.set_obstacles(obstacles)
    batch.set_state(x0)
    u_in = np.random.default_rng(3).normal(0, 0.1, (B, base["T"], 2))
    batch.set_u_prev(u_in)
    batch.run_closed_loop(n_it)
    ub, xb, Sb = batch.get_u_prev(), batch.get_state(), batch.costs()
    assert ub.shape == (B, base["T"], 2) and xb.shape == (B, batch.nx) and Sb.shape == (B, K)
    tol = dict(rtol=1e-12, atol=1e-14) if model == "diff" else dict(rtol=1e-6, atol=1e-7)
    for a in range(B):
        one = pkg.Engine(K=K, noise_stream=a, **base)
        one.set_ref_path(ref)
        one.set_obstacles(obstacles)
        one.set_state(x0[a])
        one.set_u_prev(u_in[a])
        one.run_closed_loop(n_it)
        np.testing.assert_allclose(ub[a], one.get_u_prev(), **tol)
        np.testing.assert_allclose(xb[a], one.get_state(), **tol)
        np.testing.assert_allclose(Sb[a], one.costs(), **tol)
    assert not np.allclose(ub[0], ub[1])  # the agents are different problems
    with pytest.raises(pkg.MppiError) as ex:  # the host-in-the-loop step is single-agent
        batch.step(x0[0])
    assert ex.value.code == capi.ERR_UNSUPPORTED
    with pytest.raises(pkg.MppiError) as ex:  # the sequential waypoint index cannot be batched
        pkg.Engine(K=K, n_agents=2, **dict(base, waypoint_mode=capi.WAYPOINT_SEQUENTIAL))
    assert ex.value.code == capi.ERR_UNSUPPORTED


def test_batched_agents_pick_the_layout_by_the_size_of_the_launch():
    """Four agents of K = 2048 bring 8192 samples per launch: the batched handle takes the two-samples-per-wave layout on
    its own, a single agent of K = 2048 the one-sample layout.  Same draws, same problem: f64 results agree to the
    re-association of the prefix sums."""
    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    B, K, T = 4, 2048, 50
    ref = mppi_oracle.generate_point_trajectory((0.0, 0.0), (10.0, -5.0), 100)
    base = dict(model=capi.MODEL_DIFFDRIVE, T=T, delta_t=0.1, u_max=[5.0, 3.14], param_exploration=0.05, param_lambda=1.0,
                param_alpha=0.2, sigma=[0.1, 0.0, 0.0, 0.01], stage_cost_weight=[5, 5, 10, 0],
                terminal_cost_weight=[5, 5, 10, 0], search_window=20, filter_window=10, clamp_rollout=1,
                waypoint_mode=capi.WAYPOINT_FROZEN, seed=41, precision=capi.PREC_F64)
    x0 = np.stack([[0.2 * a, -0.1 * a, 0.1 * a] for a in range(B)])
    batch = pkg.Engine(K=K, n_agents=B, **base)
    batch.set_ref_path(ref)
    batch.set_state(x0)
    batch.run_closed_loop(3)
    assert batch.counters()["rollout_layout"] & capi.LAYOUT_KIND == capi.LAYOUT_DUAL
    for a in range(B):
        one = pkg.Engine(K=K, noise_stream=a, **base)
        one.set_ref_path(ref)
        one.set_state(x0[a])
        one.run_closed_loop(3)
        assert one.counters()["rollout_layout"] & capi.LAYOUT_KIND == capi.LAYOUT_FUSED
        np.testing.assert_allclose(batch.get_u_prev()[a], one.get_u_prev(), rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(batch.get_state()[a], one.get_state(), rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(batch.costs()[a], one.costs(), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("kernel", ["f16x3", "f32"])
@pytest.mark.parametrize("name", gu.names("c5_"))
def test_config5_checkpoint_against_patched_reference(monkeypatch, name, kernel):
    """BASELINE config 5 pinned through the reference itself (tests/golden/c5_*.npz: the reference's MPPIAlgorithms with
    `_state_transition` = x + dt (f + MLP), MLP = its MultiLayerPerceptron with saved_models/mlp_diff_300x100_3l.pth;
    oracle/gen_golden.py gen_config5).  The weights travel as plain arrays.  Tolerance: north star, u within 1e-4 RMSE;
    waypoint index equal; S to 1e-3 (f32 matrix cores through 3 x 512-wide layers and T recurrent steps)."""
    import dnn_mppi_mpc_amd as pkg
    # both rollout kernels: operands split into two f16 numbers on the f16 matrix pipe (default), and f32-input MFMA
    if kernel == "f32":
        monkeypatch.setenv("MPPI_MLP_F32", "1")
    fx = gu.load(name)
    c = pkg.MPPIAlgorithms(**fx["meta"], learned_dynamics=gu.mlp_weights(name))
    c.u_prev[:] = fx["u_prev_in"]
    c.prev_way_point_idx = int(fx["idx_before"])
    eps = gu.eps_of(fx)
    c._calc_epsilon = lambda *a, **k: eps
    u0, u, _, _ = c._calc_input_control(fx["x0"])
    S = c.sample_costs()
    # the measured margins (pytest -s shows them; tools/parity_margins.py collects them into profiles/)
    print(f"\nPARITY_MARGIN config5 {name} {kernel}: u_rmse={rmse(u, fx['u_returned']):.3e} u0_rmse={rmse(u0, fx['u0_returned']):.3e} "
          f"S_max_rel={float(np.max(np.abs(S - fx['S']) / np.maximum(np.abs(fx['S']), 1e-3))):.3e} (bars: 1e-4, 1e-4, 1e-3)")
    np.testing.assert_allclose(S, fx["S"], rtol=1e-3, atol=1e-3)
    assert rmse(u, fx["u_returned"]) <= 1e-4
    assert rmse(u0, fx["u0_returned"]) <= 1e-4
    assert c.prev_way_point_idx == int(fx["idx_after"])


@pytest.mark.parametrize("name", gu.names("c5_"))
def test_config5_two_term_split_is_an_opt_in(monkeypatch, name):
    """MPPI_MLP_TERMS=2: the split product without the a_lo b_hi term (the activations rounded to f16): a third less
    matrix work, 6.5 -> 5.0 ms per iteration at BASELINE config 5.  It meets the north star's stated tolerance -- the
    optimal-control sequence within 1e-4 RMSE, here below 1e-6 -- and the waypoint index, but not the 1e-3 on S the default
    kernel is held to (2.8e-3 at K = 1024): an opt-in, never the default."""
    import dnn_mppi_mpc_amd as pkg
    monkeypatch.setenv("MPPI_MLP_TERMS", "2")
    fx = gu.load(name)
    c = pkg.MPPIAlgorithms(**fx["meta"], learned_dynamics=gu.mlp_weights(name))
    c.u_prev[:] = fx["u_prev_in"]
    c.prev_way_point_idx = int(fx["idx_before"])
    eps = gu.eps_of(fx)
    c._calc_epsilon = lambda *a, **k: eps
    u0, u, _, _ = c._calc_input_control(fx["x0"])
    S = c.sample_costs()
    assert c._engine.rollout_kernel() == "k_rollout_mlp_h3<false, 8, 2, 2>"
    print(f"\nPARITY_MARGIN config5 {name} f16x2 (opt-in): u_rmse={rmse(u, fx['u_returned']):.3e} u0_rmse={rmse(u0, fx['u0_returned']):.3e} "
          f"S_max_rel={float(np.max(np.abs(S - fx['S']) / np.maximum(np.abs(fx['S']), 1e-3))):.3e} (bars: 1e-4, 1e-4, 5e-3)")
    np.testing.assert_allclose(S, fx["S"], rtol=5e-3, atol=5e-3)
    assert rmse(u, fx["u_returned"]) <= 1e-4
    assert rmse(u0, fx["u0_returned"]) <= 1e-4
    assert c.prev_way_point_idx == int(fx["idx_after"])


def test_config5_full_size_k32768_subset_against_oracle():
    """BASELINE config 5 at its full size (K = 32768, T = 50, the real checkpoint) with the frozen waypoint index, in
    which samples are independent: the costs of a strided 1024-sample subset against the f64 NumPy restatement, then
    size-independent properties on all K -- the softmin weights sum to one, the update equals the weighted noise of
    the costs the kernel reported, and a constant shift of every cost leaves it unchanged (by construction of rho)."""
    import dnn_mppi_mpc_amd as pkg
    K, T, seed = 32768, 50, 555
    w = gu.mlp_weights()
    kw = dd_kwargs(K, T, param_exploration=0.05)
    tt = np.arange(T)
    u_in = np.stack([1.2 + 0.3 * np.sin(0.2 * tt), 0.05 * np.cos(0.1 * tt)], axis=1)
    x0 = np.array([0.4, -0.1, -0.35])
    c = pkg.MPPIAlgorithms(**kw, learned_dynamics=w, waypoint_mode="frozen", seed=seed)
    c.u_prev[:] = u_in
    u = c._calc_input_control(x0)[1].copy()  # in-kernel Philox, iteration 0
    S = c.sample_costs()
    assert S.shape == (K,) and np.isfinite(S).all()
    # --- subset against the restatement
    eps = philox.sample_epsilon(kw["sigma"], seed, 0, K, T)
    sub = np.arange(0, K, 32)
    o = mppi_oracle.DiffDriveMlpOracle(**dict(kw, num_samples_K=sub.size), mlp_weights=w)
    p0 = o.nearest_waypoint(x0[0], x0[1], 0)
    exploit = (sub < mppi_oracle.exploit_threshold(kw["param_exploration"], K))[:, None, None]
    v = o.clamp(np.where(exploit, u_in[None] + eps[sub], eps[sub].astype(np.float64)))
    X = o.rollout(x0, v)
    R, win = o.ref_path, o.ref_path[p0:p0 + 20]
    xT, yT, yawT = X[:, -1, 0], X[:, -1, 1], X[:, -1, 2]
    i = p0 + np.argmin((xT[:, None] - win[:, 0]) ** 2 + (yT[:, None] - win[:, 1]) ** 2, axis=1)
    ws, wt = o.stage_cost_weight, o.terminal_cost_weight
    q = u_in[T - 1] @ np.linalg.inv(o.Sigma)
    S_ref = ((ws[0] + wt[0]) * (xT - R[i, 0]) ** 2 + (ws[1] + wt[1]) * (yT - R[i, 1]) ** 2
             + (ws[2] + wt[2]) * (yawT - R[i, 2]) ** 2 + o.param_gamma * (q[0] * v[:, -1, 0] + q[1] * v[:, -1, 1]))
    np.testing.assert_allclose(S[sub], S_ref, rtol=1e-3, atol=1e-3)
    # --- all K: weights and the update from the kernel's own costs (f64 on the host)
    beta = 1.0 / kw["param_exploration"]
    e = np.exp(-beta * (S - S.min()))
    wk = e / e.sum()
    assert abs(c._compute_weight().sum() - 1.0) < 1e-9
    np.testing.assert_allclose(c._compute_weight(), wk, rtol=1e-5, atol=1e-12)
    w_eps = mppi_oracle.moving_average_diffdrive(np.einsum("k,ktd->td", wk, eps.astype(np.float64)), 10)
    un = u_in + w_eps
    expect = np.vstack([un[1:], un[-1:]])
    assert rmse(u, expect) <= 1e-4
    assert 1.0 <= c.last_stats.ess <= K and abs(c.last_stats.ess - 1.0 / np.sum(wk ** 2)) <= 1e-3 * c.last_stats.ess


def test_f32_device_closed_loop_matches_host_loop_over_a_traversal():
    """The bench's timed path is the f32 closed loop on the device.  Step it one iteration at a time and give a second
    controller, stepped from the host (`_calc_input_control`), the device's state before each iteration: same kernels,
    same noise -- the returned control, the waypoint index after every iteration (the whole traversal of the path and
    the first iterations at its end) and the nominal sequence must be identical.  The x0 nearest-waypoint call is the
    one piece with two implementations (finalize kernel / host side of mppi_step): both run in f64."""
    import dnn_mppi_mpc_amd as pkg
    kw = dd_kwargs(2048, 40)
    a = pkg.MPPIAlgorithms(**kw, precision="f32", seed=77)
    b = pkg.MPPIAlgorithms(**kw, precision="f32", seed=77)
    a._engine.set_state(np.zeros(3))
    moved = 0
    for it in range(45):
        x = a._engine.get_state()
        tr, st = a._engine.run_closed_loop(1, trace=True)
        u0 = b._calc_input_control(x)[0].copy()
        np.testing.assert_array_equal(tr[0], u0)
        assert st.idx_after == b.prev_way_point_idx, it
        assert st.idx_start == b.last_stats.idx_start, it
        moved += st.idx_after != st.idx_start
        np.testing.assert_array_equal(a._engine.get_u_prev(), b.u_prev)
    assert moved >= 10 and b.prev_way_point_idx == 99  # the run did traverse the path


@pytest.mark.parametrize("case", ["dd_sequential", "dd_frozen_batched", "racecar"])
def test_closed_loop_from_the_cached_graph_equals_eager_launches(monkeypatch, case):
    """With MPPI_GRAPH=1 long closed-loop calls replay their iterations from a cached HIP graph once the waypoint index
    rests (frozen index, or the sequential one at the end of the path): the same kernels with the same arguments, so the run
    must equal the eagerly launched one (the default) bit for bit -- across a change of the repeat count and calls too
    short for a graph."""
    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi

    def make():
        if case == "racecar":
            lem = mppi_oracle.generate_lemniscate_racecar(400, 10.0)
            c = pkg.MPPIRacecarController(ref_path=lem, horizon_step_T=30, number_of_samples_K=512,
                                          obstacle_circles=np.array([[5.0, 5.0, 1.0]]), visualize_optimal_traj=False,
                                          visualze_sampled_trajs=False, seed=3)
            c._engine.set_state(lem[0].astype(np.float64))
            return c._engine
        if case == "dd_frozen_batched":
            ref = mppi_oracle.generate_point_trajectory((0.0, 0.0), (10.0, -5.0), 100)
            e = pkg.Engine(model=capi.MODEL_DIFFDRIVE, K=256, n_agents=3, T=30, delta_t=0.1, u_max=[5.0, 3.14],
                           param_exploration=0.05, param_lambda=1.0, param_alpha=0.2, sigma=[0.1, 0.0, 0.0, 0.01],
                           stage_cost_weight=[5, 5, 10, 0], terminal_cost_weight=[5, 5, 10, 0], search_window=20,
                           filter_window=10, clamp_rollout=1, waypoint_mode=capi.WAYPOINT_FROZEN, seed=9)
            e.set_ref_path(ref)
            e.set_state(np.array([[0.0, 0.0, 0.0], [0.5, -0.2, 0.1], [1.0, -0.6, -0.3]]))
            return e
        c = pkg.MPPIAlgorithms(**dd_kwargs(512, 30), seed=5)
        c._engine.set_state(np.zeros(3))
        return c._engine

    def run(e):
        out = []
        e.run_closed_loop(60)    # (sequential index: still travelling -- eager either way)
        e.run_closed_loop(500)   # long: graph replays + eager remainder once the index rests
        out.append((e.get_u_prev().copy(), e.get_state().copy()))
        e.set_rollout_repeats(2)  # other arguments: another graph
        e.run_closed_loop(300)
        e.set_rollout_repeats(1)
        e.run_closed_loop(40)    # too short for a graph
        e.run_closed_loop(200)   # the first graph again
        out.append((e.get_u_prev().copy(), e.get_state().copy()))
        return out, e.counters()

    monkeypatch.delenv("MPPI_GRAPH", raising=False)
    eager, c_eager = run(make())
    monkeypatch.setenv("MPPI_GRAPH", "1")
    graph, c_graph = run(make())
    for (ue, xe), (ug, xg) in zip(eager, graph):
        np.testing.assert_array_equal(ug, ue)
        np.testing.assert_array_equal(xg, xe)
    assert c_graph["iterations"] == c_eager["iterations"] == 1100
    assert c_graph["rollout_launches"] == c_eager["rollout_launches"]
    assert c_graph["finalize_launches"] == c_eager["finalize_launches"]


def test_racecar_closed_loop_does_not_depend_on_how_it_is_chunked():
    """run(a) + run(b) == run(a + b) with the frozen waypoint index on a path that passes close to itself (the
    lemniscate's crossing): the finalize kernel has already made the next iteration's x0 call, a second one at the start
    of the next mppi_run_closed_loop would search from its result instead of prev_waypoints_idx."""
    import dnn_mppi_mpc_amd as pkg
    lem = mppi_oracle.generate_lemniscate_racecar(400, 10.0)[:300]
    kw = dict(ref_path=lem, horizon_step_T=20, number_of_samples_K=512, visualize_optimal_traj=False,
              visualze_sampled_trajs=False)
    whole = pkg.MPPIRacecarController(**kw, seed=3)
    parts = pkg.MPPIRacecarController(**kw, seed=3)
    for c in (whole, parts):
        c._engine.set_state(lem[0].astype(np.float64))
    n = 60
    tr_w, st_w = whole._engine.run_closed_loop(n, trace=True)
    tr_p = []
    for m in (1, 7, 20, 1, 31):
        tr, st_p = parts._engine.run_closed_loop(m, trace=True)
        tr_p.append(tr)
    np.testing.assert_array_equal(np.concatenate(tr_p), tr_w)
    np.testing.assert_array_equal(parts._engine.get_state(), whole._engine.get_state())
    assert st_p.idx_after == st_w.idx_after and st_w.idx_after > 0


@pytest.mark.parametrize("case", ["fused", "dual", "race", "batched"])
def test_closed_loop_from_a_noise_ring_equals_the_sampler(case):
    """`mppi_set_noise_ring`: `_calc_epsilon` materialised for the device closed loop (mppi_differential_drive.py:273-283).
    A ring filled with the sampler's own tensors of iterations 0..n-1 must give the run the in-kernel sampler gives --
    the rollout reads its noise from HBM instead of drawing it (general instantiations instead of the PLAIN ones: last-bit
    differences in f32), the slot is picked inside the kernels from the device's iteration counter, and the ring wraps."""
    import torch

    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    n_slots, n_it = 4, 7  # (wraps: iterations 4..6 reuse slots 0..2, so the reference run below rewinds its counter)
    if case == "race":
        ref = mppi_oracle.generate_lemniscate_racecar(100, 10.0)
        base = dict(model=capi.MODEL_RACECAR, K=1000, T=75, delta_t=0.05, u_max=[0.523, 2.0], wheel_base=2.5,
                    param_exploration=0.1, param_lambda=50.0, param_alpha=0.9, sigma=[0.5, 0.0, 0.0, 0.1],
                    stage_cost_weight=[50.0, 50.0, 1.0, 20.0], terminal_cost_weight=[50.0, 50.0, 1.0, 20.0],
                    beta_mode=capi.BETA_INV_LAMBDA, accumulate_stage_cost=1, waypoint_mode=capi.WAYPOINT_FROZEN,
                    search_window=200, wrap_yaw_stage=1, wrap_yaw_terminal=1, clamp_rollout=1, clamp_u_after_update=1,
                    filter_mode=capi.FILTER_RACECAR, filter_window=10, obstacle_model=capi.OBSTACLE_OUTLINE, safety_margin=1.5,
                    vehicle_w=3.0, vehicle_l=4.0, collision_penalty=1e10, seed=7, precision=capi.PREC_F64)
        x0 = ref[2].astype(np.float64)
    else:
        ref = mppi_oracle.generate_point_trajectory((0.0, 0.0), (10.0, -5.0), 100)
        base = dict(model=capi.MODEL_DIFFDRIVE, K=9000 if case == "dual" else 700, T=50, delta_t=0.1, u_max=[5.0, 3.14],
                    param_exploration=0.05, param_lambda=1.0, param_alpha=0.2, sigma=[0.1, 0.0, 0.0, 0.01],
                    stage_cost_weight=[5, 5, 10, 0], terminal_cost_weight=[5, 5, 10, 0], search_window=20, filter_window=10,
                    clamp_rollout=1, waypoint_mode=capi.WAYPOINT_FROZEN if case == "batched" else capi.WAYPOINT_SEQUENTIAL,
                    seed=99, precision=capi.PREC_F64)
        x0 = np.array([0.1, -0.05, 0.2])
        if case == "batched":
            base["n_agents"] = 3
            x0 = np.stack([x0 + 0.1 * a for a in range(3)])

    def make():
        e = pkg.Engine(**base)
        e.set_ref_path(ref)
        if case == "race":
            e.set_obstacles(np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]]))
        e.set_state(x0)
        return e
    a, b = make(), make()
    ring = torch.stack([b.sample_epsilon(i) for i in range(n_slots)])
    assert tuple(ring.shape[1:]) == b._lead + (b.K, b.T, 2)
    b.set_noise_ring(ring)
    b.run_closed_loop(n_it)
    a.run_closed_loop(n_slots)       # the sampler's run over iterations 0..3, then 0..2 again
    a.set_iteration(0)
    a.run_closed_loop(n_it - n_slots)
    b_iter = b.counters()["iterations"]
    assert b_iter == n_it
    np.testing.assert_allclose(b.get_u_prev(), a.get_u_prev(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b.get_state(), a.get_state(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b.costs(), a.costs(), rtol=1e-9, atol=1e-9)
    b.set_noise_ring(None)  # back to the sampler: the handle continues with the draw of its iteration counter
    a.set_iteration(n_it)
    a.run_closed_loop(2)
    b.run_closed_loop(2)
    np.testing.assert_allclose(b.get_u_prev(), a.get_u_prev(), rtol=1e-9, atol=1e-12)
    with pytest.raises(pkg.MppiError):  # three slots: not a power of two
        b.set_noise_ring(ring[:3].contiguous())


def test_rollout_kernel_name_follows_the_layout():
    """`mppi_get_rollout_kernel`: the instantiation the last rollout launch took, as rocprofv3 spells it (bench.py looks the
    launch's counters up by it)."""
    import dnn_mppi_mpc_amd as pkg
    c = pkg.MPPIAlgorithms(**dd_kwargs(4096, 50), precision="f32", seed=1)
    c._engine.set_state(np.zeros(3))
    c._engine.run_closed_loop(2)
    assert c._engine.rollout_kernel() == "k_rollout_fused<float, 0, 1, false, 2, true>"  # the index can still move
    c._engine.run_closed_loop(60)
    c._engine.run_closed_loop(2)
    assert c._engine.rollout_kernel() == "k_rollout_fused<float, 0, 1, false, 2, false>"  # at the end of the path: lean
    d = pkg.MPPIAlgorithms(**dd_kwargs(16384, 50), precision="f32", seed=1, waypoint_mode="frozen")
    d._engine.set_state(np.zeros(3))
    d._engine.run_closed_loop(1)
    assert d._engine.rollout_kernel() == "k_rollout_stream<float, false, false, true>"  # frozen index, `S[k] =`: streamed
    d = pkg.MPPIAlgorithms(**dd_kwargs(16384, 50), precision="f32", seed=1)  # the sequential index: two samples per wave
    d._engine.set_state(np.zeros(3))
    c0 = d._engine.counters()
    d._engine.run_closed_loop(6)
    # (the index moving: the instantiation with the look-back, and ONE rollout launch per iteration)
    assert d._engine.rollout_kernel() == "k_rollout_dual<float, 0, 2, false, 1, true, true>"
    c1 = d._engine.counters()
    assert c1["rollout_launches"] - c0["rollout_launches"] == 6 and c1["iterations"] - c0["iterations"] == 6
    d._engine.run_closed_loop(60)
    d._engine.run_closed_loop(2)
    assert d._engine.rollout_kernel() == "k_rollout_dual<float, 0, 2, false, 1, true, false>"  # at the end of the path


def test_learned_dynamics_outside_the_f16_range():
    """The default learned-dynamics kernel carries operands as pairs of f16 numbers.  (i) A path in UTM-sized coordinates
    (1e5 m, no scalers): raw inputs and first-layer pre-activations beyond 65504 are handled by per-sample power-of-two
    scales inside the kernel -- costs finite and equal to the f32-input MFMA kernel's to the split's accuracy; (ii) a WEIGHT
    beyond the range (a StandardScaler scale of 1e-6 folded into the first Linear) cannot be split: mppi_set_mlp selects
    the f32-input kernel for that model and says so."""
    import dnn_mppi_mpc_amd as pkg
    rng = np.random.default_rng(11)
    w = {"input_layer.weight": rng.normal(0, 0.3, (512, 5)), "input_layer.bias": rng.normal(0, 0.1, 512),
         "out_layer.weight": rng.normal(0, 0.05, (3, 512)), "out_layer.bias": rng.normal(0, 0.01, 3)}
    for i in range(3):
        w[f"hidden_layer.{i}.weight"] = rng.normal(0, 0.04, (512, 512))
        w[f"hidden_layer.{i}.bias"] = rng.normal(0, 0.1, 512)
    off = np.array([4.0e5, 5.5e6, 0.0])  # (UTM easting / northing)
    path = mppi_oracle.generate_point_trajectory((0.0, 0.0), (10.0, -5.0), 100) + off
    kw = dd_kwargs(512, 20, ref_path=path)
    x0 = np.array([0.3, -0.1, -0.4]) + off
    runs = {}
    for kernel in ("f16x3", "f32"):
        if kernel == "f32":
            os.environ["MPPI_MLP_F32"] = "1"
        try:
            c = pkg.MPPIAlgorithms(**kw, learned_dynamics=w, waypoint_mode="frozen", seed=3)
        finally:
            os.environ.pop("MPPI_MLP_F32", None)
        assert c._engine.rollout_kernel() == ("k_rollout_mlp(" if kernel == "f32" else "k_rollout_mlp_h3<false, 8, 2, 3>")
        u = c._calc_input_control(x0)[1].copy()
        runs[kernel] = (u, c.sample_costs().copy())
    (u_h, S_h), (u_f, S_f) = runs["f16x3"], runs["f32"]
    assert np.isfinite(S_h).all() and np.isfinite(u_h).all()
    # at 5e6 m an f32 position resolves 0.5 m: the two kernels agree to that arithmetic, not to centimetres
    np.testing.assert_allclose(S_h, S_f, rtol=2e-3)
    assert rmse(u_h, u_f) <= 1e-3
    # (ii) a weight beyond the f16 range
    scalers = dict(in_mean=np.zeros(5), in_scale=np.array([1e-6, 1.0, 1.0, 1.0, 1.0]), out_mean=np.zeros(3), out_scale=np.ones(3))
    c = pkg.MPPIAlgorithms(**dd_kwargs(256, 20), learned_dynamics=w, learned_scalers=scalers, waypoint_mode="frozen", seed=3)
    assert c._engine.rollout_kernel() == "k_rollout_mlp("
    c._calc_input_control(np.array([1e-7, 0.0, 0.0]))
    assert np.isfinite(c.sample_costs()).all()


@pytest.mark.parametrize("obstacles", [False, True])
@pytest.mark.parametrize("precision,T", [("f32", 50), ("f64", 37)])
def test_streaming_rollout_of_a_noise_tensor_against_the_c_oracle(precision, T, obstacles):
    """`k_rollout_stream` (a noise TENSOR read from HBM, several batches per workgroup, private softmin records merged
    online): diff-drive, frozen waypoint index, K = 20000 (625 batches of 32, the last workgroup short of its share),
    against the plain-C restatement of the reference loop with the same injected noise -- S, returned controls and the
    waypoint index.  T = 37: a horizon of odd length (rows of the tensor 8-byte aligned only); moderate temperature so that
    thousands of samples carry weight.  Tolerances: the f32 / f64 bars of the other full-size tests."""
    import torch

    import dnn_mppi_mpc_amd as pkg
    K = 20000
    kw = dd_kwargs(K, T, param_exploration=0.05)
    if obstacles:
        kw.update(obstacle_circles=np.array([[1.0, -0.4, 0.3], [2.5, -1.4, 0.4]]), safety_margin_rate=0.8)
    tt = np.arange(T)
    u_in = np.stack([1.0 + 0.3 * np.sin(0.2 * tt), 0.05 * np.cos(0.1 * tt)], axis=1)
    x0 = np.array([0.4, -0.1, -0.35])
    eps = philox.sample_epsilon(kw["sigma"], 77, 0, K, T)
    c = pkg.MPPIAlgorithms(**kw, precision=precision, waypoint_mode="frozen", seed=1)
    c.u_prev[:] = u_in
    c._calc_epsilon = lambda *a, **k: torch.from_numpy(eps).cuda()
    u0, u, _, _ = c._calc_input_control(x0)
    assert c._engine.rollout_kernel().startswith("k_rollout_stream<")
    o = c_oracle.DiffDriveC(**kw)
    o.u_prev[:] = u_in
    ref = o.iteration(x0, eps, frozen_threads=8)
    S = c.sample_costs()
    if precision == "f64":
        np.testing.assert_allclose(S, ref["S"], rtol=1e-9, atol=1e-9)
        assert rmse(u, ref["u_returned"]) <= 1e-8
    else:
        finite = ref["S"] < 1e9  # (a collision penalty of 1e10 swallows the tracking cost in f32)
        np.testing.assert_allclose(S[finite], ref["S"][finite], rtol=2e-4, atol=1e-4)
        np.testing.assert_allclose(S[~finite], ref["S"][~finite], rtol=1e-6)
        assert rmse(u, ref["u_returned"]) <= 1e-4
    assert c.prev_way_point_idx == ref["idx_after"]


@pytest.mark.parametrize("case", ["f64", "f32", "f64_obstacles", "f64_T100", "f64_two_shards"])
def test_per_rollout_waypoint_index_against_the_c_oracle(case):
    """MPPI_WAYPOINT_PER_ROLLOUT: the index threads through each sample's own T stage calls and its terminal call
    (mppi_differential_drive.py:228,:244) and restarts from the x0 call's index at every sample, so samples stay
    independent (K sharded, all host cores).  Against the plain-C restatement of that rule with injected noise: costs,
    returned controls, the index after the iteration (= the x0 call's); the index really moves inside the rollouts here.
    `two_shards`: the same K split over two handles (global sample index for the exploit split), merged by the split step."""
    import torch

    import dnn_mppi_mpc_amd as pkg
    f32 = case == "f32"
    K, T = 3000, 100 if case == "f64_T100" else 40
    kw = dd_kwargs(K, T, param_exploration=0.05)
    if case == "f64_obstacles":
        kw.update(obstacle_circles=np.array([[1.0, -0.4, 0.3], [2.5, -1.4, 0.4]]), safety_margin_rate=0.8)
    tt = np.arange(T)
    u_in = np.stack([2.0 + 0.3 * np.sin(0.2 * tt), -0.1 + 0.05 * np.cos(0.1 * tt)], axis=1)  # fast: the rollouts run along the path
    x0 = np.array([0.4, -0.1, -0.35])
    eps = philox.sample_epsilon(kw["sigma"], 91, 0, K, T)
    o = c_oracle.DiffDriveC(**kw)
    o.u_prev[:] = u_in
    ref = o.iteration(x0, eps, per_rollout_threads=8)
    fz = c_oracle.DiffDriveC(**kw)
    fz.u_prev[:] = u_in
    assert not np.allclose(fz.iteration(x0, eps, frozen_threads=8)["S"], ref["S"], rtol=1e-3)  # the mode matters here
    if case == "f64_two_shards":
        from dnn_mppi_mpc_amd import _capi as capi
        base = dict(model=capi.MODEL_DIFFDRIVE, T=T, delta_t=0.1, u_max=[5.0, 3.14], param_exploration=0.05, param_lambda=1.0,
                    param_alpha=0.2, sigma=[0.1, 0.0, 0.0, 0.01], stage_cost_weight=[5, 5, 10, 0], terminal_cost_weight=[5, 5, 10, 0],
                    search_window=20, filter_window=10, clamp_rollout=1, clamp_u_after_update=1,
                    waypoint_mode=capi.WAYPOINT_PER_ROLLOUT, precision=capi.PREC_F64)
        parts = [pkg.Engine(K=1700, K_global=K, k_offset=0, **base), pkg.Engine(K=1300, K_global=K, k_offset=1700, **base)]
        eps_t = torch.from_numpy(eps).cuda()
        recs = torch.empty((2, parts[0].partial_len()), dtype=torch.float64, device="cuda")
        for e, (a, b) in zip(parts, ((0, 1700), (1700, 3000))):
            e.set_ref_path(kw["ref_path"])
            e.set_u_prev(u_in)
        for r, (e, (a, b)) in enumerate(zip(parts, ((0, 1700), (1700, 3000)))):
            e.step_begin(x0, eps_t[a:b].contiguous(), recs[r])
        S = np.concatenate([e.costs() for e in parts])
        u = parts[0].step_end(recs.reshape(-1), 2)[0]
        np.testing.assert_allclose(S, ref["S"], rtol=1e-9, atol=1e-9)
        assert rmse(u, ref["u_returned"]) <= 1e-8
        return
    c = pkg.MPPIAlgorithms(**kw, precision="f32" if f32 else "f64", waypoint_mode="per_rollout", seed=1)
    c.u_prev[:] = u_in
    c._calc_epsilon = lambda *a, **k: torch.from_numpy(eps).cuda()
    u = c._calc_input_control(x0)[1]
    S = c.sample_costs()
    if f32:
        np.testing.assert_allclose(S, ref["S"], rtol=5e-4, atol=1e-3)
        assert rmse(u, ref["u_returned"]) <= 1e-4
    else:
        np.testing.assert_allclose(S, ref["S"], rtol=1e-9, atol=1e-9)
        assert rmse(u, ref["u_returned"]) <= 1e-8
    assert c.prev_way_point_idx == ref["idx_after"] == ref["idx_start"]


def test_per_rollout_waypoint_index_with_learned_dynamics():
    """The same mode through the matrix-core rollout (the index is per lane = per sample there by construction): against the
    NumPy restatement with the per-rollout scan."""
    import torch

    import dnn_mppi_mpc_amd as pkg
    K, T = 512, 30
    w = mppi_oracle.random_mlp_weights(3)
    kw = dd_kwargs(K, T, param_exploration=0.05)
    tt = np.arange(T)
    u_in = np.stack([2.0 + 0.3 * np.sin(0.2 * tt), -0.1 + 0.05 * np.cos(0.1 * tt)], axis=1)
    x0 = np.array([0.4, -0.1, -0.35])
    eps = philox.sample_epsilon(kw["sigma"], 92, 0, K, T)
    o = mppi_oracle.DiffDriveMlpOracle(**kw, mlp_weights=w)
    v = o.clamp(np.where((np.arange(K) < mppi_oracle.exploit_threshold(kw["param_exploration"], K))[:, None, None],
                         u_in[None] + eps, eps.astype(np.float64)))
    X = o.rollout(x0, v)
    p0 = o.nearest_waypoint(x0[0], x0[1], 0)
    idx = mppi_oracle.per_rollout_waypoint_scan(X, o.ref_path[:, :2], p0, 20)
    R, ws, wt = o.ref_path, o.stage_cost_weight, o.terminal_cost_weight
    i_s, i_t = idx[:, T - 1], idx[:, T]
    xT, yT, yawT = X[:, -1, 0], X[:, -1, 1], X[:, -1, 2]
    q = u_in[T - 1] @ np.linalg.inv(o.Sigma)
    S_ref = (ws[0] * (xT - R[i_s, 0]) ** 2 + ws[1] * (yT - R[i_s, 1]) ** 2 + ws[2] * (yawT - R[i_s, 2]) ** 2
             + o.param_gamma * (q[0] * v[:, -1, 0] + q[1] * v[:, -1, 1])
             + wt[0] * (xT - R[i_t, 0]) ** 2 + wt[1] * (yT - R[i_t, 1]) ** 2 + wt[2] * (yawT - R[i_t, 2]) ** 2)
    assert (idx[:, -1] > p0).any()
    c = pkg.MPPIAlgorithms(**kw, learned_dynamics=w, waypoint_mode="per_rollout", seed=1)
    c.u_prev[:] = u_in
    c._calc_epsilon = lambda *a, **k: torch.from_numpy(eps).cuda()
    c._calc_input_control(x0)
    np.testing.assert_allclose(c.sample_costs(), S_ref, rtol=1e-3, atol=1e-3)
    assert c.prev_way_point_idx == p0


def test_per_rollout_batched_agents_and_unsupported_combinations():
    """MPPI_WAYPOINT_PER_ROLLOUT with several agents per handle (agents as a grid dimension of the same kernels) against
    the agents run one by one; the race-car model has no such bookkeeping and is refused."""
    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    B, K, T, n_it = 3, 400, 40, 5
    ref = mppi_oracle.generate_point_trajectory((0.0, 0.0), (10.0, -5.0), 100)
    base = dict(model=capi.MODEL_DIFFDRIVE, T=T, delta_t=0.1, u_max=[5.0, 3.14], param_exploration=0.05, param_lambda=1.0,
                param_alpha=0.2, sigma=[0.1, 0.0, 0.0, 0.01], stage_cost_weight=[5, 5, 10, 0], terminal_cost_weight=[5, 5, 10, 0],
                search_window=20, filter_window=10, clamp_rollout=1, waypoint_mode=capi.WAYPOINT_PER_ROLLOUT, seed=13,
                precision=capi.PREC_F64)
    x0 = np.stack([[0.2 * a, -0.1 * a, 0.1 * a - 0.4] for a in range(B)])
    tt = np.arange(T)
    u_in = np.stack([np.stack([2.0 + 0.2 * a + 0.3 * np.sin(0.2 * tt), -0.1 + 0.05 * np.cos(0.1 * tt)], axis=1) for a in range(B)])
    batch = pkg.Engine(K=K, n_agents=B, **base)
    batch.set_ref_path(ref)
    batch.set_state(x0)
    batch.set_u_prev(u_in)
    batch.run_closed_loop(n_it)
    for a in range(B):
        one = pkg.Engine(K=K, noise_stream=a, **base)
        one.set_ref_path(ref)
        one.set_state(x0[a])
        one.set_u_prev(u_in[a])
        one.run_closed_loop(n_it)
        np.testing.assert_allclose(batch.get_u_prev()[a], one.get_u_prev(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(batch.costs()[a], one.costs(), rtol=1e-12, atol=1e-14)
    with pytest.raises(pkg.MppiError) as ex:
        pkg.Engine(K=K, **dict(base, model=capi.MODEL_RACECAR, wheel_base=2.5))
    assert ex.value.code == capi.ERR_UNSUPPORTED


def test_noise_ring_with_learned_dynamics():
    """The noise ring through the matrix-core rollout: a closed loop reading the sampler's own tensors from the ring equals
    the closed loop that draws them in the kernel."""
    import torch

    import dnn_mppi_mpc_amd as pkg
    w = mppi_oracle.random_mlp_weights(5)
    kw = dd_kwargs(256, 20, param_exploration=0.05)
    a = pkg.MPPIAlgorithms(**kw, learned_dynamics=w, waypoint_mode="frozen", seed=8)
    b = pkg.MPPIAlgorithms(**kw, learned_dynamics=w, waypoint_mode="frozen", seed=8)
    for c in (a, b):
        c._engine.set_state(np.array([0.1, -0.05, 0.2]))
    ring = torch.stack([b._engine.sample_epsilon(i) for i in range(4)])
    b._engine.set_noise_ring(ring)
    a._engine.run_closed_loop(4)
    b._engine.run_closed_loop(4)
    np.testing.assert_allclose(b._engine.get_u_prev(), a._engine.get_u_prev(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(b._engine.get_state(), a._engine.get_state(), rtol=0, atol=1e-6)
