"""NumPy restatement of the `pytorch_mppi`-style MPPI loop and of the reference's callbacks for it.  TEST INFRASTRUCTURE ONLY
(only tests/ may import this; the product never does).

PARITY UNPINNED.  The library `pytorch_mppi` is absent from the build container and the reference pins no version of it
(pyproject.toml:8-19 does not list it), so neither this restatement nor the HIP path (csrc/mppi_cb.hip) can be checked
against an execution of it; the reference's callers import it at module level, so they cannot be imported either.  The
callbacks follow the reference's source line by line (file:line below, relative to /root/reference); the loop follows
the published algorithm (Williams et al., "Information theoretic MPC for model-based reinforcement learning", ICRA
2017) in the arrangement `pytorch_mppi.MPPI.command` uses: shift U; noise ~ N(0, Sigma); clamp U + noise and take the
noise back from the clamped action; cost = sum_t running_cost + lambda * sum_t U_t^T Sigma^-1 noise_t; weights
exp(-(cost - min) / lambda) normalised; U += sum_k w_k noise_k; return U[0].  f32 like the callers.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def unicycle(states, actions, dt=0.05):
    """test/test_mppi.py:12-26 (same function in test/test_mppi_diff.py:9-23, train/bullet_mppi_differential_drive.py:16-31)."""
    x, y, th = states[:, 0], states[:, 1], states[:, 2]
    v, om = actions[:, 0], actions[:, 1]
    return np.stack([x + v * np.cos(th) * F(dt), y + v * np.sin(th) * F(dt), th + om * F(dt)], axis=1).astype(F)


def skid_steer(states, actions, dt=0.02, m=2.0, I=0.05, r=0.1, L=0.4, damping=0.1):
    """test/test_mppi_diff_dyna.py:13-40."""
    x, y, th, v, om = (states[:, i] for i in range(5))
    ffr, ffl, frr, frl = (actions[:, i] for i in range(4))
    dv = F(r / (4 * m)) * (ffr + ffl + frr + frl) - F(damping) * v
    dom = F(r / (L * I)) * ((ffr + frr) - (ffl + frl)) / F(2) - F(damping) * om
    return np.stack([x + v * np.cos(th) * F(dt), y + v * np.sin(th) * F(dt), th + om * F(dt), v + dv * F(dt),
                     om + dom * F(dt)], axis=1).astype(F)


def running_cost(states, actions, t, goal, q_diag, r_diag, obstacles, safety, weight, kind):
    """(s - goal)^T Q (s - goal) + u^T R u + weight * obstacle term.
    kind "inverse": test/test_mppi.py:30-50 (goal (6, 6, 1.57), Q diag(20, 5, 9), R diag(.1, .1), obstacles (5, 4), (3.5, 3.5),
    safety 0.8, weight 10; test/test_mppi_diff_dyna.py:44-64 with 5 states / 4 controls);
    kind "exponential": test/test_mppi_diff.py:25-52 (goal (5, 4, 1.57), Q diag(10, 5, 9), R diag(1, 10), obstacles
    (5, 5), (7, 3) moving with velocities (0, -0.05), (0, 0.05) per step, safety 1.0, weight 1)."""
    e = states - np.asarray(goal, F)
    c = (np.asarray(q_diag, F) * e * e).sum(1) + (np.asarray(r_diag, F) * actions * actions).sum(1)
    oc = np.zeros(states.shape[0], F)
    for ox, oy, vx, vy in np.asarray(obstacles, F).reshape(-1, 4):
        d = np.sqrt((states[:, 0] - (ox + vx * F(t))) ** 2 + (states[:, 1] - (oy + vy * F(t))) ** 2)
        if kind == "inverse":
            oc += np.where(d < F(safety), F(1) / (d + F(1e-6)), F(0))
        else:
            oc += np.exp(-(d - F(safety)))
    return (c + F(weight) * oc).astype(F)


class MPPIOracle:
    def __init__(self, dynamics, cost_kw, nx, noise_sigma, num_samples, horizon, lambda_, u_min, u_max, u_init=None,
                 sample_null_action=False, dyn_kw=None):
        self.dyn = {"unicycle": unicycle, "skid_steer": skid_steer}[dynamics]
        self.dyn_kw, self.cost_kw = dyn_kw or {}, cost_kw
        self.nx, self.K, self.T = nx, int(num_samples), int(horizon)
        self.sigma = np.asarray(noise_sigma, np.float64)
        self.nu = self.sigma.shape[0]
        self.lambda_ = F(lambda_)
        self.u_min, self.u_max = np.asarray(u_min, F), np.asarray(u_max, F)
        self.u_init = np.zeros(self.nu, F) if u_init is None else np.asarray(u_init, F)
        self.null = sample_null_action
        self.U = np.zeros((self.T, self.nu), F)

    def command(self, state, noise, shift=True):
        if shift:
            self.U = np.vstack([self.U[1:], self.u_init[None]]).astype(F)
        K, T = self.K, self.T
        act = self.U[None] + np.asarray(noise, F)
        if self.null:
            act[K - 1] = 0
        act = np.clip(act, self.u_min, self.u_max).astype(F)
        noise = act - self.U[None]
        action_cost = self.lambda_ * noise @ np.linalg.inv(self.sigma).astype(F)
        s = np.tile(np.asarray(state, F), (K, 1))
        cost = np.zeros(K, F)
        for t in range(T):
            s = self.dyn(s, act[:, t], **self.dyn_kw)
            cost += running_cost(s, act[:, t], t, **self.cost_kw)
        cost = cost + (self.U[None] * action_cost).sum(axis=(1, 2))
        w = np.exp(-(cost - cost.min()) / self.lambda_)
        w = w / w.sum()
        self.U = (self.U + (w[:, None, None] * noise).sum(0)).astype(F)
        self.cost_total, self.omega = cost, w
        return self.U[0].copy()
