"""GPU: the `pytorch_mppi`-style MPPI with built-in models (csrc/mppi_cb.hip) against its NumPy
restatement (oracle/mppi_cb_oracle.py).  PARITY UNPINNED: `pytorch_mppi` is absent from the build container and the
reference pins no version of it; the callbacks are restated from the reference's source, the loop from the published
algorithm -- these tests hold the HIP path to that restatement, not to an execution of the library."""
import numpy as np
import pytest

from oracle import mppi_cb_oracle as cbo

pytestmark = pytest.mark.gpu

CASES = {
    "static": dict(dyn="unicycle", cost="static_obstacles", nx=3, sigma=[[0.5, 0.0], [0.0, 0.3]], K=1000, T=25, lam=1.0,
                   u_min=[-2.0, -2.0], u_max=[2.0, 2.0], null=False, dt=0.05),          # test/test_mppi.py:283-307
    "moving": dict(dyn="unicycle", cost="moving_obstacles", nx=3, sigma=[[0.5, 0.0], [0.0, 0.3]], K=200, T=20, lam=1.0,
                   u_min=[-2.0, -2.0], u_max=[2.0, 2.0], null=False, dt=0.05),          # test/test_mppi_diff.py:160-185
    "skid": dict(dyn="skid_steer", cost="skid_steer", nx=5, sigma=np.eye(4).tolist(), K=500, T=30, lam=1.0,
                 u_min=[-100.0] * 4, u_max=[100.0] * 4, null=True, dt=0.02),            # test/test_mppi_diff_dyna.py:306-340
}


def _make(case):
    from dnn_mppi_mpc_amd.callback_mppi import MPPI, RunningCost
    c = CASES[case]
    rc = getattr(RunningCost, c["cost"])()
    ctrl = MPPI(c["dyn"], rc, c["nx"], np.array(c["sigma"]), num_samples=c["K"], horizon=c["T"], lambda_=c["lam"],
                u_min=c["u_min"], u_max=c["u_max"], sample_null_action=c["null"], seed=3)
    kw = dict(goal=rc.goal, q_diag=rc.q_diag, r_diag=rc.r_diag, obstacles=rc.obstacles, safety=rc.safety_distance,
              weight=rc.obstacle_weight, kind=rc.kind)
    o = cbo.MPPIOracle(c["dyn"], kw, c["nx"], np.array(c["sigma"]), c["K"], c["T"], c["lam"], c["u_min"], c["u_max"],
                       sample_null_action=c["null"], dyn_kw=dict(dt=c["dt"]))
    return c, ctrl, o, kw


@pytest.mark.parametrize("case", list(CASES))
def test_callbacks_match_the_restatement(case):
    c, ctrl, o, kw = _make(case)
    rng = np.random.default_rng(1)
    n, nu = 400, len(c["u_min"])
    s = rng.normal(0, 3, (n, c["nx"])).astype(np.float32)
    s[:40, :2] = np.array([5.0, 4.0]) + rng.normal(0, 0.3, (40, 2))  # inside the obstacles' safety distance
    a = rng.normal(0, 1.5, (n, nu)).astype(np.float32)
    np.testing.assert_allclose(ctrl._dynamics(s, a), o.dyn(s, a, dt=c["dt"]), rtol=2e-6, atol=2e-6)
    for t in (0, 7):
        want = cbo.running_cost(s, a, t, **kw)
        np.testing.assert_allclose(ctrl._running_cost(s, a, t), want, rtol=5e-6, atol=1e-4)


@pytest.mark.parametrize("case", list(CASES))
def test_command_loop_matches_the_restatement(case):
    """Five commands in closed loop with injected noise: costs, weights, nominal sequence and returned action."""
    import torch
    c, ctrl, o, kw = _make(case)
    rng = np.random.default_rng(7)
    nu = len(c["u_min"])
    L = np.linalg.cholesky(np.array(c["sigma"]))
    state = np.zeros(c["nx"], np.float32)
    if case == "skid":  # the caller starts from a random nominal sequence (test_mppi_diff_dyna.py:320)
        U0 = rng.uniform(-100, 100, (c["T"], nu)).astype(np.float32)
        ctrl.U = U0
        o.U = U0.copy()
    for it in range(5):
        noise = (rng.normal(size=(c["K"], c["T"], nu)) @ L.T).astype(np.float32)
        a = ctrl.command(state, noise=torch.from_numpy(noise).cuda())
        b = o.command(state, noise)
        scale = max(1.0, float(np.abs(o.cost_total).max()))
        np.testing.assert_allclose(ctrl.cost_total, o.cost_total, rtol=2e-4, atol=2e-4 * scale)
        assert abs(ctrl.omega.sum() - 1.0) < 1e-4
        lim = np.abs(np.array(c["u_max"])).max()
        np.testing.assert_allclose(ctrl.U, o.U, rtol=0, atol=2e-3 * lim)
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-3 * lim)
        state = o.dyn(state[None], b[None], dt=c["dt"])[0]
    traj = ctrl.get_optimal_trajectory(state)
    s = state[None]
    for t in range(c["T"]):
        s = o.dyn(s, o.U[t][None], dt=c["dt"])
        np.testing.assert_allclose(traj[t], s[0], rtol=1e-4, atol=1e-4)


def test_in_kernel_noise_has_the_requested_covariance_and_reaches_the_goal():
    """Without injected noise the draw is the in-kernel Philox sampler: the unicycle of test/test_mppi.py driven in
    closed loop moves towards its goal and keeps clear of the obstacles' centres."""
    c, ctrl, o, kw = _make("static")
    state = np.zeros(3, np.float32)
    d0 = np.hypot(6.0, 6.0)
    for _ in range(120):
        a = ctrl.command(state)
        assert np.all(np.abs(a) <= 2.0 + 1e-6)
        state = o.dyn(state[None], a[None].astype(np.float32), dt=0.05)[0]
        for ox, oy in ((5.0, 4.0), (3.5, 3.5)):
            assert np.hypot(state[0] - ox, state[1] - oy) > 0.2
    assert np.hypot(6.0 - state[0], 6.0 - state[1]) < 0.5 * d0


def test_bad_arguments():
    import dnn_mppi_mpc_amd as pkg
    from dnn_mppi_mpc_amd.callback_mppi import MPPI, RunningCost
    with pytest.raises(ValueError):
        MPPI("bicycle", RunningCost.static_obstacles(), 3, np.eye(2))
    with pytest.raises(ValueError):
        MPPI("unicycle", lambda s, a: 0, 3, np.eye(2))
    with pytest.raises(pkg.MppiError):
        MPPI("unicycle", RunningCost.static_obstacles(), 3, np.array([[1.0, 2.0], [2.0, 1.0]]))  # not positive definite
