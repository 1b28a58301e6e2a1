"""GPU: the batched stage methods of the controller classes -- `_state_transition` / `_F`,
`_compute_cost` / `_c`, `_terminal_cost` / `_phi`, `_is_collided`, `_get_nearest_waypoint` / `get_nearest_waypoint`,
`_moving_average_filter`, `_compute_weight`, `_g` -- through mppi_eval_* of the C ABI, each against the matching
function of the oracle (pinned to the reference's outputs) or, for the filters, against the reference's own outputs
(tests/golden/filters.npz).  Also mppi_step_device_x0 (the observed state handed over in device memory)."""
import numpy as np
import pytest

import golden_util as gu
from oracle import mppi_oracle

pytestmark = pytest.mark.gpu


def _dd(fx_name="dd_obs_m8_collide", precision="f64", **over):
    import dnn_mppi_mpc_amd as pkg
    fx = gu.load(fx_name)
    return fx, pkg.MPPIAlgorithms(**dict(fx["meta"], **over), precision=precision), gu.make_diffdrive_oracle(fx)


def _rc(fx_name="rc_obs_default", precision="f64"):
    import dnn_mppi_mpc_amd as pkg
    fx = gu.load(fx_name)
    return fx, pkg.MPPIRacecarController(ref_path=fx["ref_path"], **fx["meta"], precision=precision), gu.make_racecar_oracle(fx)


def test_diffdrive_state_transition_clamp_and_collision():
    fx, c, o = _dd()
    rng = np.random.default_rng(1)
    x = np.column_stack([rng.uniform(0, 5, 300), rng.uniform(0, 5, 300), rng.uniform(-4, 4, 300)])
    v = np.column_stack([rng.uniform(-8, 8, 300), rng.uniform(-5, 5, 300)])
    dt = fx["meta"]["delta_t"]
    want = np.column_stack([x[:, 0] + v[:, 0] * np.cos(x[:, 2]) * dt, x[:, 1] + v[:, 0] * np.sin(x[:, 2]) * dt,
                            x[:, 2] + v[:, 1] * dt])  # `_state_transition` :194-196 (the oracle's rollout body)
    np.testing.assert_allclose(c._state_transition(x, v), want, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(c._state_transition(x[7], v[7]), want[7], rtol=1e-12, atol=1e-13)  # the reference's call shape
    np.testing.assert_array_equal(c._g(v.copy()), o.clamp(v))
    vv = v[3].copy()
    assert c._g(vv) is vv and np.array_equal(vv, o.clamp(v[3]))  # in place, like :285-289
    np.testing.assert_array_equal(c._is_collided(x), o.collided(x[:, 0], x[:, 1]))
    assert 0 < c._is_collided(x).sum() < 300 and c._is_collided(x[0]) in (0.0, 1.0)


@pytest.mark.parametrize("precision,tol", [("f64", 1e-11), ("f32", 2e-5)])
def test_diffdrive_costs_thread_the_waypoint_index_like_successive_reference_calls(precision, tol):
    """`_compute_cost` / `_terminal_cost` move `prev_way_point_idx` (:228,:244): a batch of n states = n successive
    calls.  Expected: the oracle's sequential scan + its cost expression."""
    fx, c, o = _dd(precision=precision)
    rng = np.random.default_rng(2)
    s = np.linspace(0.0, 1.0, 120)
    x = np.column_stack([5 * s + rng.normal(0, 0.2, 120), 5 * s + rng.normal(0, 0.2, 120), rng.uniform(-1, 2, 120)])
    for terminal, w in ((False, o.stage_cost_weight), (True, o.terminal_cost_weight)):
        c.prev_way_point_idx = 4
        idx, p = mppi_oracle.sequential_waypoint_scan(x[:, 0], x[:, 1], o.ref_path[:, :2], 4, 20)
        R = o.ref_path[idx]
        want = (w[0] * (x[:, 0] - R[:, 0]) ** 2 + w[1] * (x[:, 1] - R[:, 1]) ** 2 + w[2] * (x[:, 2] - R[:, 2]) ** 2
                + 1.0e10 * o.collided(x[:, 0], x[:, 1]))
        got = c._terminal_cost(x) if terminal else c._compute_cost(x)
        hit = want > 1e9
        np.testing.assert_array_equal(got > 1e9, hit)
        np.testing.assert_allclose(got[~hit], want[~hit], rtol=tol, atol=tol)
        assert c.prev_way_point_idx == p > 4
    # one state at a time = the reference's call shape, same thread of the index
    c.prev_way_point_idx = 4
    one = np.array([c._compute_cost(row) for row in x[:10]])
    c.prev_way_point_idx = 4
    np.testing.assert_allclose(one, c._compute_cost(x[:10]), rtol=1e-12)


def test_diffdrive_nearest_waypoint():
    fx, c, o = _dd("dd_nonzero_u")
    rng = np.random.default_rng(3)
    xs, ys = rng.uniform(0, 4, 64), rng.uniform(-2, 0.5, 64)
    c.prev_way_point_idx = 7
    idx, rx, ry, ryaw = c._get_nearest_waypoint(xs, ys)  # update_prev_idx=False: every call from index 7
    want = np.array([o.nearest_waypoint(x, y, 7) for x, y in zip(xs, ys)])
    np.testing.assert_array_equal(idx, want)
    np.testing.assert_array_equal(np.column_stack([rx, ry, ryaw]), o.ref_path[want])
    assert c.prev_way_point_idx == 7
    i1, *_ = c._get_nearest_waypoint(xs[5], ys[5], update_prev_idx=True)
    assert i1 == want[5] == c.prev_way_point_idx


def test_racecar_stage_methods():
    fx, c, o = _rc(precision="f32")
    rng = np.random.default_rng(4)
    path = fx["ref_path"].astype(np.float64)
    x = path[rng.integers(0, 40, 200)] + rng.normal(0, [1.0, 1.0, 0.5, 1.0], (200, 4))
    v = np.column_stack([rng.uniform(-0.8, 0.8, 200), rng.uniform(-3, 3, 200)])
    x32, v32 = x.astype(np.float32), v.astype(np.float32)
    want = np.column_stack(o.step(x32[:, 0], x32[:, 1], x32[:, 2], x32[:, 3], v32[:, 0], v32[:, 1]))  # `_F` :183-197
    np.testing.assert_allclose(c._F(x, v), want, rtol=2e-6, atol=2e-6)
    np.testing.assert_array_equal(c._g(v.copy()).astype(np.float32), o.clamp(v32))
    c.prev_waypoints_idx = 3
    for fn, w in ((c._c, o.stage_cost_weight), (c._phi, o.terminal_cost_weight)):
        want = o.state_cost(x32, 3, w)  # `_c` / `_phi` + the outline collision term, index frozen at 3
        got = fn(x)
        hit = want > 1e9
        assert np.mean((got > 1e9) != hit) <= 0.01  # an outline point within an f32 ulp of a circle may flip
        same = (got > 1e9) == hit
        np.testing.assert_allclose(got[same & ~hit], want[same & ~hit], rtol=3e-5, atol=1e-3)
        assert c.prev_waypoints_idx == 3  # `update_prev_idx=False` inside the costs (:143)
    hits = c._is_collided(x)
    assert np.mean(hits != o.collided(x32[:, 0], x32[:, 1], x32[:, 2])) <= 0.01 and 0 < hits.sum() < 200
    idx, rx, ry, ryaw, rv = c.get_nearest_waypoint(x[:, 0], x[:, 1])
    np.testing.assert_array_equal(idx, o.nearest_waypoint(x32[:, 0], x32[:, 1], 3))
    np.testing.assert_array_equal(rv, fx["ref_path"][idx, 3])


@pytest.mark.parametrize("T", [10, 13, 20, 50, 75])
def test_moving_average_filters_against_the_reference_outputs(T):
    """The three `_moving_average_filter` implementations of the reference called directly (tests/golden/filters.npz)
    against the finalize kernel's filter code in the matching mode."""
    import dnn_mppi_mpc_amd as pkg
    fx = gu.load("filters")
    xx = fx[f"in_T{T}"]
    base = dict(gu.load("dd_c1_moderate")["meta"], num_horizons_T=T, num_samples_K=16)
    dd = pkg.MPPIAlgorithms(**base, precision="f64")
    np.testing.assert_allclose(dd._moving_average_filter(xx, 10), fx[f"dd_T{T}"], rtol=1e-12, atol=1e-15)
    ddt = pkg.MPPIAlgorithms(**base, precision="f64", variant="torch")
    np.testing.assert_allclose(ddt._moving_average_filter(xx, 10), fx[f"ddtorch_T{T}"], rtol=2e-6, atol=1e-7)
    rc_fx = gu.load("rc_circle")
    rkw = dict(rc_fx["meta"], horizon_step_T=T, number_of_samples_K=16)
    rc = pkg.MPPIRacecarController(ref_path=rc_fx["ref_path"], **rkw, precision="f32")
    np.testing.assert_allclose(rc._moving_average_filter(xx, 10), fx[f"rc_T{T}"], rtol=2e-6, atol=1e-7)
    rct = pkg.MPPIRacecarController(ref_path=rc_fx["ref_path"], **rkw, precision="f32", variant="torch")
    np.testing.assert_allclose(rct._moving_average_filter(xx, 10), fx[f"rctorch_T{T}"], rtol=2e-6, atol=1e-7)


def test_compute_weight_of_a_given_cost_vector():
    """`_compute_weight(S)` with the reference's signature, against the weights the reference itself returned."""
    for name in ("dd_c1_moderate", "dd_obs_m2"):
        fx, c, _ = _dd(name)
        np.testing.assert_allclose(c._compute_weight(fx["S"]), fx["w"], rtol=1e-9, atol=1e-300)
    fx, c, _ = _rc("rc_circle_gamma")
    np.testing.assert_allclose(c._compute_weight(fx["S"]), fx["w"], rtol=2e-4, atol=1e-12)  # the reference is f32


def test_step_with_the_observed_state_in_device_memory():
    import torch

    import dnn_mppi_mpc_amd as pkg
    fx = gu.load("dd_nonzero_u")
    out = []
    for dev in (False, True):
        c = pkg.MPPIAlgorithms(**fx["meta"], precision="f64")
        c.u_prev[:] = fx["u_prev_in"]
        c.prev_way_point_idx = int(fx["idx_before"])
        c._sync_state_to_device()
        eps = torch.from_numpy(fx["eps"]).cuda()
        x0 = torch.tensor(fx["x0"], dtype=torch.float64, device="cuda") if dev else fx["x0"]
        u, u0, st = c._engine.step(x0, eps)
        out.append((u, u0, st.idx_after, c.sample_costs()))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][3], out[1][3])
    assert out[0][2] == out[1][2] == int(fx["idx_after"])
    np.testing.assert_allclose(out[1][0], fx["u_returned"], rtol=1e-8, atol=1e-10)
