"""CPU: the plain-C oracle (oracle/mppi_oracle.c) against the reference's own outputs."""
import numpy as np
import pytest

import golden_util as gu
from oracle import c_oracle

DD_SINGLE = [n for n in gu.names("dd_") if n != "dd_closed_loop"]
RC_SINGLE = [n for n in gu.names("rc_") if n != "rc_closed_loop"]


@pytest.mark.parametrize("name", DD_SINGLE)
def test_c_diffdrive_matches_reference(name):
    fx = gu.load(name)
    o = c_oracle.DiffDriveC(**fx["meta"])
    o.u_prev[:] = fx["u_prev_in"]
    o.prev_way_point_idx = int(fx["idx_before"])
    out = o.iteration(fx["x0"], gu.eps_of(fx))
    # libm cos/sin vs NumPy's vector loops differ by an ulp here and there: 1e-11 on S.
    np.testing.assert_allclose(out["S"], fx["S"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(out["u_returned"], fx["u_returned"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(out["u0_returned"], fx["u0_returned"], rtol=1e-7, atol=1e-10)
    assert out["idx_after"] == int(fx["idx_after"])


def test_c_diffdrive_closed_loop():
    fx = gu.load("dd_closed_loop")
    o = c_oracle.DiffDriveC(**fx["meta"])
    for it in range(fx["x0"].shape[0]):
        out = o.iteration(fx["x0"][it], gu.eps_of(fx, it))
        np.testing.assert_allclose(out["S"], fx["S"][it], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(out["u_returned"], fx["u_returned"][it], rtol=1e-6, atol=1e-9)
        assert out["idx_after"] == int(fx["idx_after"][it])


@pytest.mark.parametrize("name", RC_SINGLE)
def test_c_racecar_matches_reference(name):
    fx = gu.load(name)
    o = c_oracle.RaceCarC(ref_path=fx["ref_path"], **fx["meta"])
    o.u_prev[:] = fx["u_prev_in"]
    o.prev_waypoints_idx = int(fx["idx_before"])
    out = o.iteration(fx["x0"], fx["eps"])
    # f32 path, libm cosf/sinf/tanf vs NumPy's f32 loops: a few ulps through T steps
    np.testing.assert_allclose(out["S"], fx["S"], rtol=2e-5)
    np.testing.assert_allclose(out["u_returned"], fx["u_returned"], rtol=2e-3, atol=2e-5)
    assert out["idx_after"] == int(fx["idx_after"])


def test_diffdrive_rejects_short_horizon():
    fx = gu.load("dd_small_T10")
    meta = dict(fx["meta"], num_horizons_T=9)
    o = c_oracle.DiffDriveC(**meta)
    with pytest.raises(ValueError):
        o.iteration(fx["x0"], np.zeros((o.K, 9, 2), np.float32))


def test_c_diffdrive_frozen_index_openmp_variant():
    """The all-core CPU baseline of bench.py (frozen waypoint index, OpenMP over the samples) against NumPy: with the
    index frozen at the x0 call's waypoint every sample searches the same window, so S is a plain function of x_T."""
    from oracle import c_oracle, mppi_oracle
    fx = gu.load("dd_c2_k4096_moderate")
    eps = gu.eps_of(fx)
    m = fx["meta"]
    o = c_oracle.DiffDriveC(**m)
    o.u_prev[:] = fx["u_prev_in"]
    o.prev_way_point_idx = int(fx["idx_before"])
    got = o.iteration(fx["x0"], eps, frozen_threads=4)
    n = mppi_oracle.DiffDriveOracle(**m)
    u, x0 = fx["u_prev_in"], fx["x0"]
    p0 = n.nearest_waypoint(x0[0], x0[1], int(fx["idx_before"]))
    K, T = n.K, n.T
    v = n.clamp(np.where((np.arange(K) < mppi_oracle.exploit_threshold(m["param_exploration"], K))[:, None, None],
                         u[None] + eps, eps.astype(np.float64)))
    X = n.rollout(x0, v)
    R, win = n.ref_path, n.ref_path[p0:p0 + 20]
    xT, yT, yawT = X[:, -1, 0], X[:, -1, 1], X[:, -1, 2]
    i = p0 + np.argmin((xT[:, None] - win[:, 0]) ** 2 + (yT[:, None] - win[:, 1]) ** 2, axis=1)
    ws, wt = n.stage_cost_weight, n.terminal_cost_weight
    q = u[T - 1] @ np.linalg.inv(n.Sigma)
    S = ((ws[0] + wt[0]) * (xT - R[i, 0]) ** 2 + (ws[1] + wt[1]) * (yT - R[i, 1]) ** 2
         + (ws[2] + wt[2]) * (yawT - R[i, 2]) ** 2 + n.param_gamma * (q[0] * v[:, -1, 0] + q[1] * v[:, -1, 1]))
    np.testing.assert_allclose(got["S"], S, rtol=1e-10, atol=1e-10)
    assert got["idx_after"] == p0
    wk = np.exp(-(S - S.min()) / m["param_exploration"])
    un = u + mppi_oracle.moving_average_diffdrive(np.einsum("k,ktd->td", wk / wk.sum(), eps.astype(np.float64)), 10)
    np.testing.assert_allclose(got["u_returned"], np.vstack([un[1:], un[-1:]]), rtol=1e-8, atol=1e-11)


def test_c_diffdrive_per_rollout_openmp_variant():
    """MPPI_WAYPOINT_PER_ROLLOUT in the C restatement (the index threads through each sample's own T + 1 cost calls,
    mppi_differential_drive.py:228,:244, and restarts at every sample; OpenMP over the samples) against NumPy's vectorised
    scan of the same rule."""
    from oracle import c_oracle, mppi_oracle
    fx = gu.load("dd_c2_k4096_moderate")
    eps = gu.eps_of(fx)
    m = fx["meta"]
    o = c_oracle.DiffDriveC(**m)
    o.u_prev[:] = fx["u_prev_in"]
    o.prev_way_point_idx = int(fx["idx_before"])
    got = o.iteration(fx["x0"], eps, per_rollout_threads=4)
    n = mppi_oracle.DiffDriveOracle(**m)
    u, x0 = fx["u_prev_in"], fx["x0"]
    p0 = n.nearest_waypoint(x0[0], x0[1], int(fx["idx_before"]))
    K, T = n.K, n.T
    v = n.clamp(np.where((np.arange(K) < mppi_oracle.exploit_threshold(m["param_exploration"], K))[:, None, None],
                         u[None] + eps, eps.astype(np.float64)))
    X = n.rollout(x0, v)
    idx = mppi_oracle.per_rollout_waypoint_scan(X, n.ref_path[:, :2], p0, 20)
    assert (idx[:, -1] > p0).any()  # the index does move inside the rollouts of this fixture
    R = n.ref_path
    i_s, i_t = idx[:, T - 1], idx[:, T]
    xT, yT, yawT = X[:, -1, 0], X[:, -1, 1], X[:, -1, 2]
    ws, wt = n.stage_cost_weight, n.terminal_cost_weight
    q = u[T - 1] @ np.linalg.inv(n.Sigma)
    S = (ws[0] * (xT - R[i_s, 0]) ** 2 + ws[1] * (yT - R[i_s, 1]) ** 2 + ws[2] * (yawT - R[i_s, 2]) ** 2
         + n.param_gamma * (q[0] * v[:, -1, 0] + q[1] * v[:, -1, 1])
         + wt[0] * (xT - R[i_t, 0]) ** 2 + wt[1] * (yT - R[i_t, 1]) ** 2 + wt[2] * (yawT - R[i_t, 2]) ** 2)
    np.testing.assert_allclose(got["S"], S, rtol=1e-10, atol=1e-10)
    assert got["idx_after"] == p0  # the x0 call's index: the rollouts' indices do not survive the sample
