"""GPU: ragged and degenerate shapes, both wave layouts, error behaviour at the boundary."""
import numpy as np
import pytest

from oracle import mppi_oracle, philox

pytestmark = pytest.mark.gpu


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, float) - np.asarray(b, float)) ** 2)))


def dd_case(rng, K, T, n_ref, obstacles):
    ref = mppi_oracle.generate_point_trajectory((0.0, 0.0), (rng.uniform(3, 10), rng.uniform(-5, 5)), n_ref)
    kw = dict(delta_t=float(rng.choice([0.05, 0.1])), ref_path=ref, max_speed=float(rng.uniform(0.5, 5.0)),
              max_omega=float(rng.uniform(0.3, 3.14)), num_samples_K=K, num_horizons_T=T,
              param_exploration=float(rng.choice([0.0, 0.05, 0.3, 1.0])) or 0.02, param_lambda=float(rng.uniform(0.5, 20)),
              param_alpha=float(rng.uniform(0.1, 0.99)),
              sigma=np.array([[0.2, 0.03], [0.03, 0.05]]) if rng.random() < 0.5 else np.array([[0.1, 0.0], [0.0, 0.01]]),
              stage_cost_weight=rng.uniform(1, 10, 3), terminal_cost_weight=rng.uniform(1, 10, 3),
              visualize_optimal_traj=False, visualze_sampled_trajs=bool(rng.random() < 0.5))
    if obstacles:
        kw.update(obstacle_circles=np.column_stack([rng.uniform(0.5, 5, obstacles), rng.uniform(-3, 3, obstacles),
                                                    rng.uniform(0.2, 0.6, obstacles)]), safety_margin_rate=0.8)
    return kw


@pytest.mark.parametrize("dual", ["0", "1"])
@pytest.mark.parametrize("K,T,n_ref,obstacles", [(1, 10, 100, 0), (2, 11, 100, 0), (17, 10, 3, 0), (33, 25, 1, 2),
                                                  (31, 64, 100, 1), (100, 33, 25, 0), (257, 50, 100, 3), (1000, 63, 7, 0)])
def test_ragged_diffdrive_shapes_match_oracle(monkeypatch, dual, K, T, n_ref, obstacles):
    """K not a multiple of the block size, odd horizons, paths shorter than the search window, both layouts."""
    import dnn_mppi_mpc_amd as pkg
    monkeypatch.setenv("MPPI_DUAL", dual)
    rng = np.random.default_rng(K * 1000 + T)
    kw = dd_case(rng, K, T, n_ref, obstacles)
    eps = philox.sample_epsilon(kw["sigma"], K + T, 0, K, T)
    x0 = np.array([rng.uniform(0, 1), rng.uniform(-0.5, 0.5), rng.uniform(-1, 1)])
    u_in = rng.normal(0, 0.3, (T, 2))
    o = mppi_oracle.DiffDriveOracle(**kw)
    o.u_prev[:] = u_in
    c = pkg.MPPIAlgorithms(**kw, precision="f64")
    c.u_prev[:] = u_in
    c._calc_epsilon = lambda *a, **k: eps
    for it in range(2):
        ref = o.iteration(x0, eps.astype(np.float64))
        u = c._calc_input_control(x0)[1]
        S = c.sample_costs()
        hit = ref["S"] > 1e9
        np.testing.assert_array_equal(S > 1e9, hit)
        np.testing.assert_allclose(S[~hit], ref["S"][~hit], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(u, ref["u_returned"], rtol=1e-7, atol=1e-9)
        assert c.prev_way_point_idx == ref["idx_after"]
        x0 = mppi_oracle.diffdrive_plant_step(x0, ref["u0_returned"], kw["delta_t"])


@pytest.mark.parametrize("dual", ["0", "1"])
@pytest.mark.parametrize("K,T", [(1, 5), (19, 5), (64, 64), (65, 63), (130, 30), (33, 65), (40, 75), (3, 94), (65, 96), (20, 97)])
def test_ragged_racecar_shapes_match_oracle(monkeypatch, dual, K, T):
    """(64 < T <= 96: `dual` switches between three steps per lane, k_rollout_tri, and two, k_rollout_dual<.., 1, ..>.)"""
    import dnn_mppi_mpc_amd as pkg
    monkeypatch.setenv("MPPI_DUAL", dual)
    monkeypatch.setenv("MPPI_TRI", dual)
    lem = mppi_oracle.generate_lemniscate_racecar(60, 10.0)
    kw = dict(ref_path=lem, horizon_step_T=T, number_of_samples_K=K, param_exploration=0.2, param_alpha=0.8,
              param_lambda=30.0, visualize_optimal_traj=True, visualze_sampled_trajs=False)
    sigma = np.array([[0.5, 0.0], [0.0, 0.1]])
    eps = philox.sample_epsilon(sigma, K * T, 0, K, T)
    o = mppi_oracle.RaceCarOracle(**kw)
    c = pkg.MPPIRacecarController(**kw, precision="f32")
    c._calc_epsilon = lambda *a, **k: eps
    ref = o.iteration(lem[4], eps)
    u = c._calc_control_input(lem[4])[1]
    np.testing.assert_allclose(c.sample_costs(), ref["S"], rtol=3e-5, atol=1e-3)
    assert rmse(u, ref["u_returned"]) <= 1e-4
    assert c.prev_waypoints_idx == ref["idx_after"]
    if 64 < T <= 96:
        assert c._engine.rollout_kernel().startswith("k_rollout_tri<" if dual == "1" else "k_rollout_dual<float, 1, 1, ")


def test_boundary_errors():
    import torch

    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    base = dict(model=capi.MODEL_DIFFDRIVE, K=64, T=20, delta_t=0.1, u_max=[1.0, 1.0], param_exploration=0.1,
                param_lambda=1.0, param_alpha=0.5, sigma=[0.1, 0.0, 0.0, 0.1], stage_cost_weight=[1, 1, 1, 0],
                terminal_cost_weight=[1, 1, 1, 0], search_window=20, filter_window=10, clamp_rollout=1)
    e = pkg.Engine(**base)
    with pytest.raises(pkg.MppiError) as ex:  # ref_path never set
        e.step(np.zeros(3))
    assert ex.value.code == capi.ERR_STATE
    e.set_ref_path(mppi_oracle.generate_point_trajectory((0, 0), (1, 1), 10))
    with pytest.raises(ValueError):  # wrong noise shape
        e.step(np.zeros(3), torch.zeros(64, 19, 2, device="cuda"))
    with pytest.raises(ValueError):  # wrong dtype
        e.step(np.zeros(3), torch.zeros(64, 20, 2, device="cuda", dtype=torch.float64))
    with pytest.raises(ValueError):  # host tensor where a device tensor is required
        e.step(np.zeros(3), torch.zeros(64, 20, 2))
    with pytest.raises(pkg.MppiError) as ex:  # sequential index cannot be sharded
        pkg.Engine(**dict(base, K=32, K_global=64, k_offset=0))
    assert ex.value.code == capi.ERR_UNSUPPORTED
    with pytest.raises(pkg.MppiError) as ex:  # non-SPD sigma
        pkg.Engine(**dict(base, sigma=[0.1, 0.5, 0.5, 0.1]))
    assert ex.value.code == capi.ERR_BAD_ARG
    with pytest.raises(pkg.MppiError) as ex:  # horizon below the filter window (the reference's filter raises)
        pkg.Engine(**dict(base, T=9))
    assert ex.value.code == capi.ERR_SHAPE
    u, u0, st = e.step(np.zeros(3))  # and the handle still works after the failed calls
    assert np.isfinite(u).all() and st.iteration == 1


def test_checkpoint_resume_reproduces_the_run():
    """u_prev + waypoint index + sampler iteration are the whole controller state."""
    import dnn_mppi_mpc_amd as pkg
    from test_gpu_engine import dd_kwargs
    kw = dd_kwargs(512, 30)
    a = pkg.MPPIAlgorithms(**kw, precision="f64", seed=11)
    x = np.array([0.1, 0.0, -0.2])
    for _ in range(4):
        u0 = a._calc_input_control(x)[0].copy()
        x = mppi_oracle.diffdrive_plant_step(x, u0, kw["delta_t"])
    b = pkg.MPPIAlgorithms(**kw, precision="f64", seed=11)
    b.u_prev[:] = a.u_prev
    b.prev_way_point_idx = a.prev_way_point_idx
    b._engine.set_iteration(4)
    ua = a._calc_input_control(x)[1].copy()
    ub = b._calc_input_control(x)[1].copy()
    np.testing.assert_array_equal(ua, ub)


@pytest.mark.parametrize("pair", ["0", "1"])
@pytest.mark.parametrize("K,T,obstacles", [(33, 65, 0), (100, 100, 2), (17, 128, 0), (300, 77, 1)])
def test_long_horizon_layouts_match_oracle(monkeypatch, pair, K, T, obstacles):
    """64 < T <= 128: two 64-step chunks per wave (MPPI_PAIR=0) and one pass with two steps per lane (=1)."""
    import dnn_mppi_mpc_amd as pkg
    monkeypatch.setenv("MPPI_PAIR", pair)
    rng = np.random.default_rng(K * 977 + T)
    kw = dd_case(rng, K, T, 100, obstacles)
    eps = philox.sample_epsilon(kw["sigma"], K + T, 0, K, T)
    x0 = np.array([rng.uniform(0, 1), rng.uniform(-0.5, 0.5), rng.uniform(-1, 1)])
    u_in = rng.normal(0, 0.3, (T, 2))
    o = mppi_oracle.DiffDriveOracle(**kw)
    o.u_prev[:] = u_in
    c = pkg.MPPIAlgorithms(**kw, precision="f64")
    c.u_prev[:] = u_in
    c._calc_epsilon = lambda *a, **k: eps
    ref = o.iteration(x0, eps.astype(np.float64))
    u = c._calc_input_control(x0)[1]
    S = c.sample_costs()
    hit = ref["S"] > 1e9
    np.testing.assert_array_equal(S > 1e9, hit)
    np.testing.assert_allclose(S[~hit], ref["S"][~hit], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(u, ref["u_returned"], rtol=1e-7, atol=1e-9)
    assert c.prev_way_point_idx == ref["idx_after"]


@pytest.mark.parametrize("pair", ["0", "1"])
@pytest.mark.parametrize("T,K", [(75, 150), (128, 37), (65, 16), (127, 300)])
def test_long_horizon_racecar(monkeypatch, pair, T, K):
    """64 < T <= 128 with the f32 `S[k] +=` order: both layouts; T = 128 fills the last column of the cost rows the
    workgroup-wide ordered sum keeps in LDS."""
    import dnn_mppi_mpc_amd as pkg
    monkeypatch.setenv("MPPI_PAIR", pair)
    lem = mppi_oracle.generate_lemniscate_racecar(80, 10.0)
    kw = dict(ref_path=lem, horizon_step_T=T, number_of_samples_K=K, param_exploration=0.1, param_alpha=0.9,
              obstacle_circles=np.array([[5.0, 5.0, 1.0], [7.0, 7.0, 1.0]]), visualize_optimal_traj=True,
              visualze_sampled_trajs=False)
    eps = philox.sample_epsilon(np.array([[0.5, 0.0], [0.0, 0.1]]), 11, 0, K, T)
    o = mppi_oracle.RaceCarOracle(**kw)
    c = pkg.MPPIRacecarController(**kw, precision="f32")
    c._calc_epsilon = lambda *a, **k: eps
    ref = o.iteration(lem[2], eps)
    u = c._calc_control_input(lem[2])[1]
    hit = ref["S"] > 1e9
    np.testing.assert_allclose(c.sample_costs()[~hit], ref["S"][~hit], rtol=3e-5, atol=1e-3)
    assert rmse(u, ref["u_returned"]) <= 1e-4


@pytest.mark.parametrize("case", ["diff-dual", "diff-pair", "race-dual", "race-pair"])
def test_two_samples_per_wave_in_sequence(monkeypatch, case):
    """MPPI_SEQ=2: k_rollout_dual's second pass (picked by itself only past one workgroup per CU, K >= 8192): ragged
    sample counts where the second pass is partly or wholly empty, obstacles, the sequential waypoint index."""
    monkeypatch.setenv("MPPI_SEQ", "2")
    if case == "diff-dual":
        for K, T, n_ref, obstacles in [(1, 10, 100, 0), (33, 25, 1, 2), (65, 64, 100, 1), (257, 50, 100, 3), (1000, 63, 7, 0)]:
            test_ragged_diffdrive_shapes_match_oracle(monkeypatch, "1", K, T, n_ref, obstacles)
    elif case == "diff-pair":
        for K, T, obstacles in [(17, 128, 0), (33, 65, 0), (300, 77, 1)]:
            test_long_horizon_layouts_match_oracle(monkeypatch, "1", K, T, obstacles)
    elif case == "race-dual":
        for K, T in [(19, 5), (65, 63), (130, 30)]:
            test_ragged_racecar_shapes_match_oracle(monkeypatch, "1", K, T)
    else:
        for T, K in [(75, 150), (128, 37)]:
            test_long_horizon_racecar(monkeypatch, "1", T, K)


def _curved_path(rng, kind, n):
    """Paths whose nearest waypoint is NOT monotone along a rollout: a loop and a hairpin (the first minimum of a call
    can lie behind the running index, or far ahead of it), besides the straight line of the reference's driver."""
    s = np.linspace(0.0, 1.0, n)
    if kind == "line":
        x, y = 8.0 * s, -3.0 * s
    elif kind == "loop":
        a = 2.0 * np.pi * s
        x, y = 3.0 * np.sin(a), 3.0 * np.sin(a) * np.cos(a)
    else:  # hairpin: out and back, 0.4 apart
        x = np.where(s < 0.5, 10.0 * s, 10.0 * (1.0 - s))
        y = np.where(s < 0.5, 0.0, 0.4) + 0.05 * np.sin(20.0 * s)
    yaw = np.arctan2(np.gradient(y), np.gradient(x))
    return np.stack([x, y, yaw], axis=1)


@pytest.mark.parametrize("seed", range(24))
def test_sequential_index_exact_on_curved_paths(monkeypatch, seed):
    """Randomised sweep of the reference-exact (sequential) waypoint index: curved paths, a start in the middle of the
    path, every rollout layout.  Exercises what the straight path of the other tests does not: first minima behind
    the running index (the search proper inside the threading), windows that stop short of the path's end, index
    jumps of many waypoints -- the index, the costs and the controls must equal the oracle's over three iterations."""
    import dnn_mppi_mpc_amd as pkg
    rng = np.random.default_rng(7000 + seed)
    kind = ("line", "loop", "hairpin")[seed % 3]
    n_ref = int(rng.choice([30, 100, 260]))
    K = int(rng.integers(1, 150))
    T = int(rng.choice([10, 17, 40, 64, 65, 90, 128]))  # (the reference filter needs T >= 10)
    monkeypatch.setenv("MPPI_DUAL", str(int(rng.integers(0, 2))))
    monkeypatch.setenv("MPPI_PAIR", str(int(rng.integers(0, 2))))
    monkeypatch.setenv("MPPI_SEQ", str(int(rng.integers(1, 3))))
    kw = dd_case(rng, K, T, n_ref, int(rng.integers(0, 3)))
    kw["ref_path"] = _curved_path(rng, kind, n_ref)
    kw["max_speed"] = float(rng.uniform(1.0, 5.0))
    i0 = int(rng.integers(0, max(1, n_ref // 2)))
    x0 = kw["ref_path"][i0] + rng.normal(0, [0.15, 0.15, 0.3])
    u_in = np.column_stack([rng.uniform(0.3, 1.0, T) * kw["max_speed"], rng.normal(0, 0.3, T)])
    eps = philox.sample_epsilon(kw["sigma"], 31 + seed, 0, K, T)
    o = mppi_oracle.DiffDriveOracle(**kw)
    variant = "cuda" if seed % 4 == 3 else "numpy"  # (a quarter of the sweep with the 10-candidate window of `_cuda`)
    if variant == "cuda":
        o.SEARCH_IDX_LEN, o.WRAP_YAW_TERMINAL = 10, True
    c = pkg.MPPIAlgorithms(**kw, precision="f64", variant=variant)
    o.u_prev[:] = u_in
    c.u_prev[:] = u_in
    o.prev_way_point_idx = c.prev_way_point_idx = max(0, i0 - int(rng.integers(0, 4)))
    c._calc_epsilon = lambda *a, **k: eps
    for it in range(3):
        ref = o.iteration(x0, eps.astype(np.float64))
        u = c._calc_input_control(x0)[1]
        assert c.prev_way_point_idx == ref["idx_after"], (kind, it)
        S = c.sample_costs()
        hit = ref["S"] > 1e9
        np.testing.assert_array_equal(S > 1e9, hit)
        np.testing.assert_allclose(S[~hit], ref["S"][~hit], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(u, ref["u_returned"], rtol=1e-7, atol=1e-9)
        x0 = mppi_oracle.diffdrive_plant_step(x0, ref["u0_returned"], kw["delta_t"])


@pytest.mark.parametrize("model", ["diff", "race"])
def test_more_than_64_obstacles(model):
    """The obstacle table sits in the lanes of a wave (64 circles); the rest is read from memory.  70 small circles,
    some of them on the path: the collision flags must equal the oracle's sample by sample."""
    import dnn_mppi_mpc_amd as pkg
    rng = np.random.default_rng(6464)
    if model == "diff":
        K, T = 200, 40
        kw = dd_case(rng, K, T, 100, 0)
        kw.update(obstacle_circles=np.column_stack([rng.uniform(0.3, 8, 70), rng.uniform(-5, 5, 70), rng.uniform(0.05, 0.25, 70)]),
                  safety_margin_rate=0.8)
        eps = philox.sample_epsilon(kw["sigma"], 5, 0, K, T)
        o = mppi_oracle.DiffDriveOracle(**kw)
        c = pkg.MPPIAlgorithms(**kw, precision="f64")
        c._calc_epsilon = lambda *a, **k: eps
        x0 = np.array([0.1, 0.0, -0.3])
        ref = o.iteration(x0, eps.astype(np.float64))
        u = c._calc_input_control(x0)[1]
        hit = ref["S"] > 1e9
        assert 0 < hit.sum() < K
        np.testing.assert_array_equal(c.sample_costs() > 1e9, hit)
        np.testing.assert_allclose(c.sample_costs()[~hit], ref["S"][~hit], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(u, ref["u_returned"], rtol=1e-7, atol=1e-9)
    else:
        K, T = 150, 75
        lem = mppi_oracle.generate_lemniscate_racecar(80, 10.0)
        circles = np.column_stack([rng.uniform(-12, 12, 90), rng.uniform(-8, 8, 90), rng.uniform(0.1, 0.5, 90)])
        circles = circles[np.hypot(circles[:, 0] - lem[2, 0], circles[:, 1] - lem[2, 1]) > 4.0]  # none on the start pose
        kw = dict(ref_path=lem, horizon_step_T=T, number_of_samples_K=K, param_exploration=0.1, param_alpha=0.9,
                  obstacle_circles=circles, visualize_optimal_traj=True, visualze_sampled_trajs=False)
        assert len(circles) > 64
        eps = philox.sample_epsilon(np.array([[0.5, 0.0], [0.0, 0.1]]), 12, 0, K, T)
        o = mppi_oracle.RaceCarOracle(**kw)
        c = pkg.MPPIRacecarController(**kw, precision="f32")
        c._calc_epsilon = lambda *a, **k: eps
        ref = o.iteration(lem[2], eps)
        u = c._calc_control_input(lem[2])[1]
        hit = ref["S"] > 1e9
        assert 0 < hit.sum() < K
        # (a pose within f32 rounding of a circle could flip; none does with this seed)
        np.testing.assert_array_equal(c.sample_costs() > 1e9, hit)
        np.testing.assert_allclose(c.sample_costs()[~hit], ref["S"][~hit], rtol=3e-5, atol=1e-3)
        assert rmse(u, ref["u_returned"]) <= 1e-4


def test_exchange_api_errors():
    """mppi_comm_*: call-order and mode errors are reported, the handle stays usable."""
    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd import _capi as capi
    base = dict(model=capi.MODEL_DIFFDRIVE, K=64, T=20, delta_t=0.1, u_max=[1.0, 1.0], param_exploration=0.1,
                param_lambda=1.0, param_alpha=0.5, sigma=[0.1, 0.0, 0.0, 0.1], stage_cost_weight=[1, 1, 1, 0],
                terminal_cost_weight=[1, 1, 1, 0], search_window=20, filter_window=10, clamp_rollout=1)
    seq = pkg.Engine(**base)  # sequential waypoint index: no exchange
    with pytest.raises(pkg.MppiError) as ex:
        seq.comm_export(2)
    assert ex.value.code == capi.ERR_UNSUPPORTED
    e = pkg.Engine(**dict(base, waypoint_mode=capi.WAYPOINT_FROZEN, K=32, K_global=64, k_offset=0))
    with pytest.raises(pkg.MppiError) as ex:  # probe / connect before export
        e.comm_probe()
    assert ex.value.code == capi.ERR_STATE
    with pytest.raises(pkg.MppiError) as ex:
        e.comm_connect(0, [b"\0" * 64, b"\0" * 64])
    assert ex.value.code == capi.ERR_STATE
    with pytest.raises(pkg.MppiError) as ex:  # one rank is not an exchange
        e.comm_export(1)
    assert ex.value.code == capi.ERR_BAD_ARG
    h = e.comm_export(2)
    assert len(h) == 64 and e.comm_buffer()
    e.comm_close()
    e.set_ref_path(mppi_oracle.generate_point_trajectory((0, 0), (1, 1), 10))
    # without a connected exchange the handle is an ordinary (sharded: use the split step) one
    assert e.lib.mppi_comm_handle_bytes() == 64


@pytest.mark.parametrize("dual", ["0", "1"])
@pytest.mark.parametrize("variant", ["numpy", "cuda"])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_one_launch_index_resolution_and_its_fallback(monkeypatch, precision, variant, dual):
    """The sequential index resolved in one launch (the threaded index as a running maximum of the calls' first minima,
    a look-back across workgroups: fused_lookback / k_rollout_dual<..., LB>) and its fallback: on a densely sampled path a
    fast robot carries the index further than the 32 candidates reach within one iteration, the workgroup that sees it
    marks its word and the speculation rounds redo the iteration (rounds > 1); on a coarse path the candidates suffice
    (rounds == 1).  Either way index, costs and controls equal the oracle's.  `variant="cuda"`: the 10-candidate window
    (and terminal yaw wrap) of mppi_differential_drive_cuda.py:201,:239.  `dual`: one / two samples per wave."""
    import dnn_mppi_mpc_amd as pkg
    monkeypatch.setenv("MPPI_DUAL", dual)
    seen = set()
    for n_ref, speed in ((400, 4.0), (60, 1.0)):
        rng = np.random.default_rng(n_ref)
        K, T = 200, 40
        kw = dd_case(rng, K, T, n_ref, 0)
        kw.update(max_speed=speed + 1.0, delta_t=0.1, param_exploration=0.1)
        u_in = np.column_stack([np.full(T, speed), rng.normal(0, 0.05, T)])
        eps = philox.sample_epsilon(kw["sigma"], 5, 0, K, T)
        o = mppi_oracle.DiffDriveOracle(**kw)
        if variant == "cuda":
            o.SEARCH_IDX_LEN, o.WRAP_YAW_TERMINAL = 10, True
        c = pkg.MPPIAlgorithms(**kw, precision=precision, variant=variant)
        o.u_prev[:] = u_in
        c.u_prev[:] = u_in
        c._calc_epsilon = lambda *a, **k: eps
        x0 = kw["ref_path"][0] + np.array([0.05, -0.03, 0.0])
        for it in range(3):
            ref = o.iteration(x0, eps.astype(np.float64))
            u = c._calc_input_control(x0)[1]
            assert c.prev_way_point_idx == ref["idx_after"], (n_ref, it)
            seen.add(c.last_stats.rounds > 1)
            if ref["idx_start"] < n_ref - 1:  # (the index can move: the instantiation that resolves it in the launch, HYPK / LB)
                assert c._engine.rollout_kernel().endswith(", true>"), c._engine.rollout_kernel()
            tol = dict(rtol=1e-9, atol=1e-9) if precision == "f64" else dict(rtol=3e-4, atol=3e-4)
            np.testing.assert_allclose(c.sample_costs(), ref["S"], **tol)
            assert rmse(u, ref["u_returned"]) <= (1e-8 if precision == "f64" else 1e-4)
            x0 = mppi_oracle.diffdrive_plant_step(x0, ref["u0_returned"], kw["delta_t"])
    assert seen == {True, False}  # both the fallback and the pure one-launch resolution ran
