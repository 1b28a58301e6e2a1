"""GPU: the HIP path (through the C ABI) against the reference's own outputs (tests/golden) and the
oracle.  Tolerances: f64 kernels reproduce the f64 reference to rounding; the f32 kernels meet the
north-star bound -- optimal-control sequence within 1e-4 RMSE of the CPU reference."""
import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu

DD_SINGLE = [n for n in gu.names("dd_") if n != "dd_closed_loop"]
RC_SINGLE = [n for n in gu.names("rc_") if n != "rc_closed_loop"]
RMSE_TOL = 1e-4  # BASELINE.json north_star


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, float) - np.asarray(b, float)) ** 2)))


def make_dd(fx, precision, **over):
    import dnn_mppi_mpc_amd as pkg
    c = pkg.MPPIAlgorithms(**dict(fx["meta"], **over), precision=precision)
    if "u_prev_in" in fx:
        c.u_prev[:] = fx["u_prev_in"]
        c.prev_way_point_idx = int(fx["idx_before"])
    return c


def make_rc(fx, precision, **over):
    import dnn_mppi_mpc_amd as pkg
    c = pkg.MPPIRacecarController(ref_path=fx["ref_path"], **dict(fx["meta"], **over), precision=precision)
    if "u_prev_in" in fx:
        c.u_prev[:] = fx["u_prev_in"]
        c.prev_waypoints_idx = int(fx["idx_before"])
    return c


def inject(ctrl, eps):
    ctrl._calc_epsilon = lambda *a, **k: eps


@pytest.mark.parametrize("name", DD_SINGLE)
def test_diffdrive_f64_matches_reference(name):
    fx = gu.load(name)
    c = make_dd(fx, "f64")
    inject(c, gu.eps_of(fx))
    u0, u, opt, smp = c._calc_input_control(fx["x0"])
    S = c.sample_costs()
    np.testing.assert_allclose(S, fx["S"], rtol=1e-10, atol=1e-10)
    assert int(np.argmin(S)) == int(np.argmin(fx["S"]))
    np.testing.assert_allclose(c._compute_weight(), fx["w"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(u, fx["u_returned"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(u0, fx["u0_returned"], rtol=1e-8, atol=1e-10)
    assert u is c.u_prev and np.shares_memory(u0, u)  # the reference's aliasing (:165)
    assert c.prev_way_point_idx == int(fx["idx_after"])
    if "sampled_traj_list" in fx and fx["meta"]["visualze_sampled_trajs"]:
        np.testing.assert_allclose(opt, fx["optimal_traj"], rtol=1e-5, atol=2e-6)  # returned as f32
        np.testing.assert_allclose(smp, fx["sampled_traj_list"], rtol=1e-5, atol=2e-6)
    else:
        assert not opt.any() and not smp.any()


@pytest.mark.parametrize("name", DD_SINGLE)
def test_diffdrive_f32_within_north_star_tolerance(name):
    fx = gu.load(name)
    c = make_dd(fx, "f32")
    inject(c, gu.eps_of(fx))
    u0, u, _, _ = c._calc_input_control(fx["x0"])
    S = c.sample_costs()
    collided = fx["S"] > 1e9
    np.testing.assert_array_equal(S > 1e9, collided)
    np.testing.assert_allclose(S[~collided], fx["S"][~collided], rtol=2e-4, atol=2e-4)
    assert rmse(u, fx["u_returned"]) <= RMSE_TOL
    assert rmse(u0, fx["u0_returned"]) <= RMSE_TOL
    assert c.prev_way_point_idx == int(fx["idx_after"])


@pytest.mark.parametrize("precision,tol", [("f64", 1e-8), ("f32", RMSE_TOL)])
def test_diffdrive_closed_loop(precision, tol):
    """12 iterations with the reference's plant in the loop (fixture from the reference itself)."""
    from oracle import mppi_oracle
    fx = gu.load("dd_closed_loop")
    c = make_dd(fx, precision)
    state = np.zeros(3)
    for it in range(fx["x0"].shape[0]):
        inject(c, gu.eps_of(fx, it))
        # drive with the reference's recorded states so one diverging iteration cannot mask the next
        u0, u, _, _ = c._calc_input_control(fx["x0"][it])
        assert rmse(u, fx["u_returned"][it]) <= tol, it
        assert c.prev_way_point_idx == int(fx["idx_after"][it]), it
        state = mppi_oracle.diffdrive_plant_step(fx["x0"][it], u0, fx["meta"]["delta_t"])
    np.testing.assert_allclose(state, fx["final_state"], atol=10 * tol)


# Fixtures in which (nearly) every sample collides: S = n * 1e10 + tracking cost, and in the f32 reference
# ulp(1e10) = 1024 decides which samples tie for the minimum.  Only a kernel that adds in f32 in the
# reference's order can reproduce those ties, so the f64 kernels are not compared on them.
COLLISION_DOMINATED = {"rc_obs_all_collided", "rc_obs_T75"}


@pytest.mark.parametrize("name", RC_SINGLE)
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_racecar_matches_reference(name, precision):
    fx = gu.load(name)
    c = make_rc(fx, precision)
    inject(c, fx["eps"])
    u0, u, opt, smp = c._calc_control_input(fx["x0"])
    S = c.sample_costs()
    # the reference is f32 (1e10 collision penalties swallow the tracking cost: ulp(1e10) = 1024)
    np.testing.assert_allclose(S, fx["S"], rtol=2e-5, atol=1e-3)
    assert c.prev_waypoints_idx == int(fx["idx_after"])
    if precision == "f64" and name in COLLISION_DOMINATED:
        return
    assert rmse(u, fx["u_returned"]) <= RMSE_TOL
    assert rmse(u0, fx["u0_returned"]) <= RMSE_TOL
    if "sampled_traj_list" in fx and fx["meta"]["visualze_sampled_trajs"]:
        np.testing.assert_allclose(opt, fx["optimal_traj"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(smp, fx["sampled_traj_list"], rtol=1e-4, atol=1e-4)


def test_racecar_all_collided_against_f64_semantics():
    """Every sample collides: S is a multiple of 1e10 plus a tracking cost the f32 reference rounds to
    1024-sized steps.  The engine's weights must agree with an exact-arithmetic evaluation of the same S."""
    fx = gu.load("rc_obs_all_collided")
    c = make_rc(fx, "f64")
    inject(c, fx["eps"])
    c._calc_control_input(fx["x0"])
    S = c.sample_costs()
    w = c._compute_weight()
    e = np.exp(-(S - S.min()) / fx["meta"]["param_lambda"])
    np.testing.assert_allclose(w, e / e.sum(), rtol=1e-9, atol=1e-300)
    assert (S > 1e10 - 1).all()


def test_racecar_closed_loop():
    fx = gu.load("rc_closed_loop")
    c = make_rc(fx, "f32")
    for it in range(fx["x0"].shape[0]):
        inject(c, gu.eps_of(fx, it))
        u0, u, _, _ = c._calc_control_input(fx["x0"][it])
        np.testing.assert_allclose(c.sample_costs(), fx["S"][it], rtol=3e-5, atol=1e-3)
        assert rmse(u, fx["u_returned"][it]) <= RMSE_TOL, it
        assert c.prev_waypoints_idx == int(fx["idx_after"][it])


def test_racecar_raises_at_path_end():
    fx = gu.load("rc_circle")
    c = make_rc(fx, "f32")
    c.prev_waypoints_idx = 95
    end = fx["ref_path"][-1].astype(np.float64)
    with pytest.raises(IndexError):
        c._calc_control_input(end)
    assert c.prev_waypoints_idx == fx["ref_path"].shape[0] - 1
    assert not c.u_prev.any() or np.array_equal(c.u_prev, fx["u_prev_in"])  # u untouched


def test_short_horizon_raises_like_reference():
    fx = gu.load("dd_small_T10")
    with pytest.raises(ValueError):
        make_dd(fx, "f32", num_horizons_T=9)


@pytest.mark.parametrize("name", ["rc_circle", "rc_obs_default", "rc_obs_T75"])
def test_racecar_torch_variant_filter(name):
    """`variant="torch"` (mppi_race_car_torch.py): everything as in the NumPy file except the moving average, which is
    the conv1d form pinned by tests/golden/filters.npz.  Expected = the oracle (which reproduces the fixture of the
    NumPy file) with that one function swapped."""
    from oracle import mppi_oracle
    fx = gu.load(name)
    o = gu.make_racecar_oracle(fx)
    o.filter_fn = mppi_oracle.moving_average_torch
    ref = o.iteration(fx["x0"], fx["eps"])
    c = make_rc(fx, "f32", variant="torch")
    inject(c, fx["eps"])
    u0, u, _, _ = c._calc_control_input(fx["x0"])
    np.testing.assert_allclose(c.sample_costs(), fx["S"], rtol=2e-5, atol=1e-3)  # the filter does not touch S
    assert rmse(u, ref["u_returned"]) <= RMSE_TOL
    assert rmse(u0, ref["u0_returned"]) <= RMSE_TOL
    plain = make_rc(fx, "f32")
    inject(plain, fx["eps"])
    assert rmse(plain._calc_control_input(fx["x0"])[1], u) > 1e-3 or name == "rc_obs_T75"  # the variants do differ


@pytest.mark.parametrize("name", ["dd_c1_moderate", "dd_nonzero_u", "dd_clamped"])
def test_diffdrive_cuda_variant(name):
    """`variant="cuda"` (mppi_differential_drive_cuda.py = the CPU file with cp. for np., a 10-waypoint search window
    and the terminal yaw wrapped): the oracle with those two switches set.  That file cannot run here (no cupy), so
    this pins the engine to the restatement of a source diff, not to an execution of it."""
    fx = gu.load(name)
    o = gu.make_diffdrive_oracle(fx)
    o.SEARCH_IDX_LEN, o.WRAP_YAW_TERMINAL = 10, True
    x0 = fx["x0"] + np.array([0.0, 0.0, -0.9])  # a negative yaw, so that the wrap matters
    ref = o.iteration(x0, gu.eps_of(fx))
    c = make_dd(fx, "f64", variant="cuda")
    inject(c, gu.eps_of(fx))
    u0, u, _, _ = c._calc_input_control(x0)
    np.testing.assert_allclose(c.sample_costs(), ref["S"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(u, ref["u_returned"], rtol=1e-7, atol=1e-9)
    assert c.prev_way_point_idx == ref["idx_after"]
    plain = gu.make_diffdrive_oracle(fx).iteration(x0, gu.eps_of(fx))
    assert not np.allclose(plain["S"], ref["S"])  # the switches do change the costs


@pytest.mark.parametrize("precision,stol", [("f64", 3e-5), ("f32", 2e-4)])
@pytest.mark.parametrize("name", gu.names("ddtorch_"))
def test_diffdrive_torch_variant_matches_reference(name, precision, stol):
    """`variant="torch"` against controllers/mppi_differential_drive_torch.py itself (run on the CPU in f32 with its x0
    aliasing removed, oracle/gen_golden.py gen_dd_torch): beta = lambda (:187-190), no clamp in the rollout (:128),
    terminal yaw wrap (:231), conv1d moving average (:252-263).  The reference is f32, so even the f64 kernels meet it
    at f32 rounding only."""
    fx = gu.load(name)
    c = make_dd(fx, precision, variant="torch")
    inject(c, fx["eps"])
    u0, u, _, _ = c._calc_input_control(fx["x0"])
    np.testing.assert_allclose(c.sample_costs(), fx["S"], rtol=stol, atol=stol)
    assert rmse(u, fx["u_returned"]) <= RMSE_TOL
    assert rmse(u0, fx["u0_returned"]) <= RMSE_TOL
    assert c.prev_way_point_idx == int(fx["idx_after"])
    plain = make_dd(fx, precision)  # the NumPy-file semantics on the same inputs differ
    inject(plain, fx["eps"])
    assert rmse(plain._calc_input_control(fx["x0"])[1], fx["u_returned"]) > 1e-3


def test_racecar_device_plant_against_vehicle_update():
    """`mppi_run_closed_loop` on a race-car handle: the plant on the device is `Vehicle.update` (models/vehicle.py:85-114)
    and the loop is the reference controller fed the vehicle's state (tests/golden/plant_rc_vehicle.npz, `closed_*`).
    The noise is the engine's own Philox draw, which the fixture's generator restates in NumPy (same seed)."""
    import dnn_mppi_mpc_amd as pkg
    fx = gu.load("plant_rc_vehicle")
    n = fx["closed_u0"].shape[0]
    c = pkg.MPPIRacecarController(ref_path=fx["ref_path"], **fx["meta"], precision="f32", seed=int(fx["eps_seed"]))
    c._engine.set_state(fx["closed_vehicle"][0])
    trace, st = c._engine.run_closed_loop(n, trace=True)
    np.testing.assert_allclose(trace, fx["closed_u0"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(c._engine.get_state(), fx["closed_vehicle"][-1], rtol=0, atol=1e-4)
    np.testing.assert_allclose(c._engine.get_u_prev(), fx["closed_u_final"], rtol=0, atol=2e-4)
    assert st.idx_after == int(fx["closed_idx_after"][-1])
    # the driver's own loop (mppi_race_car.py:259-281): controller fed ref_path[i], vehicle integrating beside it
    d = pkg.MPPIRacecarController(ref_path=fx["ref_path"], **fx["meta"], precision="f32", seed=int(fx["eps_seed"]))
    from oracle import mppi_oracle
    m, veh = fx["meta"], fx["driver_vehicle"][0]
    for i in range(n):
        u0 = d._calc_control_input(fx["driver_x0"][i])[0].copy()
        assert rmse(u0, fx["driver_u0"][i]) <= RMSE_TOL, i
        assert d.prev_waypoints_idx == int(fx["driver_idx_after"][i])
        veh = mppi_oracle.racecar_plant_step(veh, u0, m["delta_t"], m["wheel_base"], m["max_steer_abs"], m["max_accel_abs"])
    np.testing.assert_allclose(veh, fx["driver_vehicle"][-1], rtol=0, atol=1e-4)
