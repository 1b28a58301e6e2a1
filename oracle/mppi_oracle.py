"""CPU restatement (NumPy) of the reference's MPPI iteration.  TEST INFRASTRUCTURE ONLY.

This module is the *checker* for the HIP engine: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``dnn-mppi-mpc_amd``) never does.

Parity status: PINNED.  Every function here is checked against outputs of the reference
itself (``tests/golden/*.npz``, written by ``oracle/gen_golden.py`` which imports
``/root/reference/controllers/mppi_*.py`` unmodified with an injected noise tensor).

All ``file:line`` citations are relative to ``/root/reference``.

Two controller families are restated, quirks included:

* ``DiffDriveOracle``  -- controllers/mppi_differential_drive.py:42-289 and
  controllers/mppi_differential_drive_obs.py:42-313 (f64).
* ``RaceCarOracle``    -- controllers/mppi_race_car.py:9-222 and
  controllers/mppi_race_car_obstacle.py:10-274 (f32 under NumPy >= 2, NEP 50).
"""
from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------------------
# Shared stages
# --------------------------------------------------------------------------------------


def exploit_threshold(param_exploration: float, K: int) -> float:
    """Real-valued split ``k < (1.0 - expl) * K`` (mppi_differential_drive.py:116)."""
    return (1.0 - param_exploration) * K


def sequential_sum(a: np.ndarray, axis: int = 0) -> np.ndarray:
    """Left-to-right sum along ``axis`` in the array's dtype.

    The reference accumulates with Python ``+=`` loops (mppi_differential_drive.py:132-135,
    :173-175); ``np.cumsum`` is sequential (``np.sum`` is pairwise), so this reproduces
    the loop's rounding.
    """
    return np.take(np.cumsum(a, axis=axis, dtype=a.dtype), -1, axis=axis)


def moving_average_diffdrive(xx: np.ndarray, window_size: int = 10) -> np.ndarray:
    """mppi_differential_drive.py:257-271 -- 'same' convolution plus the edge fix-ups,
    including the ``xx_mean[-1]`` line that is hit ``n_conv-1`` times (bug kept)."""
    b = np.ones(window_size) / window_size
    dim = xx.shape[1]
    xx_mean = np.zeros(xx.shape)
    for d in range(dim):
        xx_mean[:, d] = np.convolve(xx[:, d], b, mode="same")
        n_conv = math.ceil(window_size / 2)
        xx_mean[0, d] *= window_size / n_conv
        for i in range(1, n_conv):
            xx_mean[i, d] *= window_size / (i + n_conv)
            xx_mean[-1, d] *= window_size / (i + n_conv - (window_size % 2))
    return xx_mean


def moving_average_torch(xx: np.ndarray, window_size: int = 10) -> np.ndarray:
    """mppi_differential_drive_torch.py:252-263 == mppi_race_car_torch.py:211-222: the signal padded with copies of
    its first / last window//2 rows goes through ``conv1d(padding=window//2)`` and the FIRST T outputs are kept --
    so output n averages padded rows n-5 .. n+4 with zeros before the start, i.e. the NumPy race-car filter delayed
    by window//2 rows (pinned by tests/golden/filters.npz; f32 like the torch files)."""
    h = window_size // 2
    xx = np.asarray(xx, np.float32)
    T = xx.shape[0]
    padded = np.concatenate([np.zeros((h, xx.shape[1]), np.float32), xx[:h], xx, xx[T - h:]], axis=0)
    out = np.zeros_like(xx)
    w = np.float32(1.0 / window_size)
    for n in range(T):
        acc = np.zeros(xx.shape[1], np.float32)
        for k in range(window_size):
            acc = acc + padded[n + k] * w
        out[n] = acc
    return out


def moving_average_racecar(xx: np.ndarray, window_size: int = 10) -> np.ndarray:
    """mppi_race_car.py:211-222 -- pad with copies of the first/last 5 rows, 'same'
    convolution, slice the padding off.  dtype follows ``xx`` (f32 in the reference)."""
    ks = window_size
    kernel = np.ones(ks, dtype=np.float32) / ks
    dim = xx.shape[1]
    xx_mean = np.zeros_like(xx)
    for d in range(dim):
        xx_padded = np.concatenate([xx[: ks // 2, d], xx[:, d], xx[-ks // 2:, d]], axis=0)
        xx_mean[:, d] = np.convolve(xx_padded, kernel, mode="same")[ks // 2: -ks // 2]
    return xx_mean


def sequential_waypoint_scan(px, py, ref_xy, p, window):
    """Thread the reference's single ``prev_way_point_idx`` through ``len(px)`` calls.

    Call ``n`` does ``p <- p + argmin_{j < min(window, N-p)} d(pos_n, ref[p+j])`` (first
    minimum), exactly mppi_differential_drive.py:201-220 with ``update_prev_idx=True``.
    Returns (idx_used_by_call[n], p_final).  Vectorised by speculation: evaluate all
    pending calls under the current ``p``, commit up to the first call that moves it.
    """
    n_calls = px.shape[0]
    idx = np.empty(n_calls, dtype=np.int64)
    n0 = 0
    chunk = 32768
    while n0 < n_calls:
        n1 = min(n_calls, n0 + chunk)
        win = ref_xy[p: p + window]
        dx = px[n0:n1, None] - win[None, :, 0]
        dy = py[n0:n1, None] - win[None, :, 1]
        d = dx ** 2 + dy ** 2
        am = np.argmin(d, axis=1)
        moved = np.nonzero(am)[0]
        if moved.size == 0:
            idx[n0:n1] = p
            n0 = n1
            continue
        f = int(moved[0])
        idx[n0: n0 + f] = p
        p = p + int(am[f])
        idx[n0 + f] = p
        n0 = n0 + f + 1
    return idx, p


def per_rollout_waypoint_scan(X, ref_xy, p0, window):
    """The waypoint index of every cost call when the index threads through each sample's OWN calls and starts from ``p0``
    (the x0 call's index) at every sample -- the engine's MPPI_WAYPOINT_PER_ROLLOUT: call t of sample k does
    ``p_k <- p_k + argmin_{j < min(window, N - p_k)} d(X[k, t], ref[p_k + j])`` (first minimum, mppi_differential_drive.py:201-220
    with ``update_prev_idx=True``), the terminal call once more on the last state (:244).  X: [K, T, >=2].
    Returns idx[K, T + 1] (column T: the terminal call).  Vectorised over the samples."""
    K, T = X.shape[:2]
    n = ref_xy.shape[0]
    p = np.full(K, int(p0), dtype=np.int64)
    idx = np.empty((K, T + 1), dtype=np.int64)
    j = np.arange(window)
    for t in range(T + 1):
        tt = min(t, T - 1)
        cand = p[:, None] + j[None, :]
        ok = cand < n
        cc = np.minimum(cand, n - 1)
        d = (X[:, tt, 0:1] - ref_xy[cc, 0]) ** 2 + (X[:, tt, 1:2] - ref_xy[cc, 1]) ** 2
        d = np.where(ok, d, np.inf)
        p = p + np.argmin(d, axis=1)
        idx[:, t] = p
    return idx


# --------------------------------------------------------------------------------------
# Differential drive (f64)
# --------------------------------------------------------------------------------------


class DiffDriveOracle:
    """Restates ``MPPIAlgorithms`` (mppi_differential_drive.py:42-289; `_obs` :42-313)."""

    SEARCH_IDX_LEN = 20  # mppi_differential_drive.py:204
    # mppi_differential_drive_cuda.py is this file with np. -> cp., SEARCH_IDX_LEN = 10 (:201) and the terminal yaw
    # wrapped into [0, 2 pi) (:239) -- set both on an instance to restate it (confirmed by diffing
    # the two sources; the file itself cannot run here, cupy is absent)
    WRAP_YAW_TERMINAL = False
    # mppi_differential_drive_torch.py: the rollout does not clamp (:128), the softmin rate is lambda itself
    # (:187-190), the terminal yaw is wrapped (:231) and the moving average is the conv1d form (:252-263) -- set
    # CLAMP_ROLLOUT = False, BETA_IS_LAMBDA = True, WRAP_YAW_TERMINAL = True, FILTER = moving_average_torch
    # (pinned by tests/golden/ddtorch_*.npz: that file run on the CPU with its x0 aliasing removed)
    CLAMP_ROLLOUT = True
    BETA_IS_LAMBDA = False
    FILTER = None  # None: moving_average_diffdrive

    def __init__(self, delta_t, ref_path, max_speed, max_omega, num_samples_K, num_horizons_T,
                 param_exploration, param_lambda, param_alpha, sigma, stage_cost_weight,
                 terminal_cost_weight, obstacle_circles=None, safety_margin_rate=None,
                 visualize_optimal_traj=True, visualze_sampled_trajs=True, visualize_sampled_traj=None):
        if visualize_sampled_traj is not None:  # the torch file's spelling (:63-64)
            visualze_sampled_trajs = visualize_sampled_traj
        self.delta_t = float(delta_t)
        self.ref_path = np.asarray(ref_path, dtype=np.float64)
        self.max_speed = float(max_speed)
        self.max_omega = float(max_omega)
        self.T = int(num_horizons_T)
        self.K = int(num_samples_K)
        self.param_exploration = float(param_exploration)
        self.param_lambda = float(param_lambda)
        self.param_alpha = float(param_alpha)
        self.param_gamma = self.param_lambda * (1.0 - self.param_alpha)  # :74
        self.Sigma = np.asarray(sigma, dtype=np.float64)
        self.stage_cost_weight = np.asarray(stage_cost_weight, dtype=np.float64)
        self.terminal_cost_weight = np.asarray(terminal_cost_weight, dtype=np.float64)
        self.obstacle_circles = None if obstacle_circles is None else np.asarray(obstacle_circles, np.float64)
        self.safety_margin_rate = safety_margin_rate
        self.visualize_optimal_traj = visualize_optimal_traj
        self.visualze_sampled_trajs = visualze_sampled_trajs
        self.u_prev = np.zeros((self.T, 2))  # :82
        self.prev_way_point_idx = 0  # :85

    # -- stages -----------------------------------------------------------------------
    def nearest_waypoint(self, x, y, p):
        """:201-220 for one call; returns the new index."""
        win = self.ref_path[p: p + self.SEARCH_IDX_LEN]
        d = (x - win[:, 0]) ** 2 + (y - win[:, 1]) ** 2
        return p + int(np.argmin(d))

    def clamp(self, v):
        """`_g` :285-289 (vectorised, returns a new array)."""
        out = np.empty_like(v)
        out[..., 0] = np.clip(v[..., 0], -self.max_speed, self.max_speed)
        out[..., 1] = np.clip(v[..., 1], -self.max_omega, self.max_omega)
        return out

    def collided(self, x, y):
        """`_is_collided` mppi_differential_drive_obs.py:301-313 (vectorised)."""
        if self.obstacle_circles is None:
            return np.zeros_like(x)
        robot_radius = 0.5 * self.safety_margin_rate
        hit = np.zeros(x.shape, dtype=bool)
        for ox, oy, orad in self.obstacle_circles:
            hit |= (x - ox) ** 2 + (y - oy) ** 2 < (robot_radius + orad) ** 2
        return hit.astype(np.float64)

    def rollout(self, x0, v):
        """`_state_transition` :182-198 over all samples; v is the clamped [K,T,2]."""
        K, T = v.shape[:2]
        dt = self.delta_t
        X = np.empty((K, T, 3))
        x = np.full(K, x0[0], dtype=np.float64)
        y = np.full(K, x0[1], dtype=np.float64)
        yaw = np.full(K, x0[2], dtype=np.float64)
        for t in range(T):
            speed = v[:, t, 0]
            omega = v[:, t, 1]
            x, y, yaw = x + speed * np.cos(yaw) * dt, y + speed * np.sin(yaw) * dt, yaw + omega * dt
            X[:, t, 0], X[:, t, 1], X[:, t, 2] = x, y, yaw
        return X

    def compute_weight(self, S):
        """`_compute_weight` :167-180 (sequential eta)."""
        rho = S.min()
        beta = self.param_lambda if self.BETA_IS_LAMBDA else 1.0 / self.param_exploration
        e = np.exp(-beta * (S - rho))
        eta = sequential_sum(e)
        return (1 / eta) * e

    # -- one iteration ------------------------------------------------------------------
    def iteration(self, observed_x, epsilon):
        """`_calc_input_control` :87-165 with ``epsilon`` injected in place of
        `_calc_epsilon` :273-283.  Mutates ``u_prev`` / ``prev_way_point_idx`` like the
        reference and returns every intermediate the fixtures record."""
        K, T = self.K, self.T
        x0 = np.asarray(observed_x, dtype=np.float64)
        eps = np.asarray(epsilon, dtype=np.float64)
        u = self.u_prev.copy()
        out = {"idx_before": self.prev_way_point_idx}

        # :96-99
        p = self.nearest_waypoint(x0[0], x0[1], self.prev_way_point_idx)
        out["path_end"] = bool(p >= self.ref_path.shape[0] - 1)
        if out["path_end"]:
            p = self.ref_path.shape[0] - 1
        out["idx_start"] = p

        # :116-121
        thr = exploit_threshold(self.param_exploration, K)
        exploit = (np.arange(K) < thr)[:, None, None]
        v = np.where(exploit, u[None] + eps, eps)
        if self.CLAMP_ROLLOUT:
            v = self.clamp(v)
        X = self.rollout(x0, v)

        # :124,:126 -- waypoint state threaded through K*(T+1) calls, k-major
        px = np.concatenate([X[:, :, 0], X[:, -1:, 0]], axis=1).reshape(-1)
        py = np.concatenate([X[:, :, 1], X[:, -1:, 1]], axis=1).reshape(-1)
        idx, p = sequential_waypoint_scan(px, py, self.ref_path[:, :2], p, self.SEARCH_IDX_LEN)
        idx = idx.reshape(K, T + 1)
        i_stage, i_term = idx[:, T - 1], idx[:, T]
        xT, yT, yawT = X[:, -1, 0], X[:, -1, 1], X[:, -1, 2]
        R = self.ref_path
        w, wt = self.stage_cost_weight, self.terminal_cost_weight
        coll = self.collided(xT, yT)
        stage = (w[0] * (xT - R[i_stage, 0]) ** 2 + w[1] * (yT - R[i_stage, 1]) ** 2
                 + w[2] * (yawT - R[i_stage, 2]) ** 2)
        yaw_term = (yawT + 2.0 * np.pi) % (2.0 * np.pi) if self.WRAP_YAW_TERMINAL else yawT
        term = (wt[0] * (xT - R[i_term, 0]) ** 2 + wt[1] * (yT - R[i_term, 1]) ** 2
                + wt[2] * (yaw_term - R[i_term, 2]) ** 2)
        if self.obstacle_circles is not None:
            stage = stage + coll * 1.0e10
            term = term + coll * 1.0e10
        q = u[T - 1].T @ np.linalg.inv(self.Sigma)  # :124, last step only survives
        ctrl = self.param_gamma * (q[0] * v[:, T - 1, 0] + q[1] * v[:, T - 1, 1])
        S = (stage + ctrl) + term
        out["S"] = S
        out["idx_after"] = p
        self.prev_way_point_idx = p

        wgt = self.compute_weight(S)
        out["w"] = wgt
        w_eps = sequential_sum(wgt[:, None, None] * eps, axis=0)  # :132-135
        out["w_eps_raw"] = w_eps
        w_eps = moving_average_diffdrive(w_eps, 10) if self.FILTER is None else type(self).FILTER(w_eps, 10)  # :138
        out["w_eps_filtered"] = w_eps
        u = u + w_eps  # :141
        out["u_pre_clamp"] = u.copy()
        if self.visualze_sampled_trajs:  # :145-149 clamps every row of u in place
            u = self.clamp(u)
        out["u_pre_shift"] = u.copy()
        # :162-165 (alias => returned u is the shifted sequence, u[0] the pre-shift u[1])
        self.u_prev[:-1] = u[1:]
        self.u_prev[-1] = u[-1]
        out["u_returned"] = self.u_prev.copy()
        out["u0_returned"] = self.u_prev[0].copy()
        out["v"] = v
        out["X"] = X
        return out

    def viz_trajectories(self, x0, u_pre_shift, v):
        """:144-159 -- note the ``[t-1]`` indexing: step t is driven by control t-1 (t=0
        wraps to the last row)."""
        T = self.T
        order = (np.arange(T) - 1) % T
        opt = self.rollout(np.asarray(x0, np.float64), self.clamp(u_pre_shift[None, order]))[0]
        smp = self.rollout(np.asarray(x0, np.float64), v[:, order])
        return opt, smp


# --------------------------------------------------------------------------------------
# Race car (f32, NumPy >= 2 promotion rules)
# --------------------------------------------------------------------------------------

F32 = np.float32


class RaceCarOracle:
    """Restates ``MPPIRacecarController`` (mppi_race_car.py:9-222; `_obstacle` :10-274)."""

    SEARCH_INDEX_LEN = 200  # mppi_race_car.py:158
    filter_fn = staticmethod(lambda xx, w: moving_average_racecar(xx, w))  # the torch file: moving_average_torch

    def __init__(self, delta_t=0.05, wheel_base=2.5, max_steer_abs=0.523, max_accel_abs=2.0,
                 ref_path=None, horizon_step_T=10, number_of_samples_K=100, param_exploration=0.01,
                 param_lambda=50.0, param_alpha=1.0, sigma=None, stage_cost_weight=None,
                 terminal_cost_weight=None, obstacle_circles=None, collision_safety_margin_rat=1.5,
                 visualize_optimal_traj=True, visualze_sampled_trajs=True, raise_at_path_end=True):
        self.T = int(horizon_step_T)
        self.K = int(number_of_samples_K)
        self.param_exploration = param_exploration
        self.param_lambda = param_lambda
        self.param_alpha = param_alpha
        self.param_gamma = self.param_lambda * (1.0 - self.param_alpha)
        sigma = np.array([[0.5, 0.0], [0.0, 0.1]]) if sigma is None else sigma
        sw = np.array([50.0, 50.0, 1.0, 20.0]) if stage_cost_weight is None else stage_cost_weight
        tw = np.array([50.0, 50.0, 1.0, 20.0]) if terminal_cost_weight is None else terminal_cost_weight
        self.Sigma = np.asarray(sigma).astype(F32)
        self.stage_cost_weight = np.asarray(sw).astype(F32)
        self.terminal_cost_weight = np.asarray(tw).astype(F32)
        self.delta_t = delta_t
        self.wheel_base = wheel_base
        self.max_steer_abs = max_steer_abs
        self.max_accel_abs = max_accel_abs
        self.ref_path = np.asarray(ref_path).astype(F32)
        self.vehicle_w, self.vehicle_l = 3.0, 4.0  # mppi_race_car_obstacle.py:53-54
        self.obstacle_circles = None if obstacle_circles is None else np.asarray(obstacle_circles, np.float64)
        self.collision_safety_margin_rate = collision_safety_margin_rat
        self.visualize_optimal_traj = visualize_optimal_traj
        self.visualze_sampled_trajs = visualze_sampled_trajs
        self.raise_at_path_end = raise_at_path_end
        self.u_prev = np.zeros((self.T, 2), dtype=F32)
        self.prev_waypoints_idx = 0

    def nearest_waypoint(self, x, y, p):
        """`get_nearest_waypoint` mppi_race_car.py:157-174, vectorised over calls."""
        win = self.ref_path[p: p + self.SEARCH_INDEX_LEN]
        dx = np.asarray(x, F32)[..., None] - win[:, 0]
        dy = np.asarray(y, F32)[..., None] - win[:, 1]
        d = dx ** 2 + dy ** 2
        return p + np.argmin(d, axis=-1)

    def clamp(self, v):
        out = np.empty_like(v)
        out[..., 0] = np.clip(v[..., 0], -self.max_steer_abs, self.max_steer_abs)
        out[..., 1] = np.clip(v[..., 1], -self.max_accel_abs, self.max_accel_abs)
        return out

    def step(self, x, y, yaw, vel, steer, accel):
        """`_F` mppi_race_car.py:183-197."""
        l, dt = self.wheel_base, self.delta_t
        new_x = x + vel * np.cos(yaw) * dt
        new_y = y + vel * np.sin(yaw) * dt
        new_yaw = yaw + vel / l * np.tan(steer) * dt
        new_v = vel + accel * dt
        return new_x, new_y, new_yaw, new_v

    def rollout(self, x0, v):
        K, T = v.shape[:2]
        X = np.empty((K, T, 4), dtype=F32)
        x = np.full(K, x0[0], F32)
        y = np.full(K, x0[1], F32)
        yaw = np.full(K, x0[2], F32)
        vel = np.full(K, x0[3], F32)
        for t in range(T):
            x, y, yaw, vel = self.step(x, y, yaw, vel, v[:, t, 0], v[:, t, 1])
            X[:, t, 0], X[:, t, 1], X[:, t, 2], X[:, t, 3] = x, y, yaw, vel
        return X

    def collided(self, x, y, yaw):
        """`_is_collided` + `_affine_transform` mppi_race_car_obstacle.py:241-274: outline
        points in f32, circle test in f64 (the circles stay a f64 array, :57)."""
        if self.obstacle_circles is None:
            return np.zeros(x.shape, dtype=F32)
        vw = self.vehicle_w * self.collision_safety_margin_rate
        vl = self.vehicle_l * self.collision_safety_margin_rate
        sx = [-0.5 * vl, -0.5 * vl, 0.0, +0.5 * vl, +0.5 * vl, +0.5 * vl, 0.0, -0.5 * vl, -0.5 * vl]
        sy = [0.0, +0.5 * vw, +0.5 * vw, +0.5 * vw, 0.0, -0.5 * vw, -0.5 * vw, -0.5 * vw, 0.0]
        c, s = np.cos(yaw), np.sin(yaw)
        hit = np.zeros(x.shape, dtype=bool)
        for a, b in zip(sx, sy):
            qx = F32(a) * c - F32(b) * s + x
            qy = F32(a) * s + F32(b) * c + y
            for ox, oy, orad in self.obstacle_circles:
                hit |= (qx.astype(np.float64) - ox) ** 2 + (qy.astype(np.float64) - oy) ** 2 < orad ** 2
        return hit.astype(F32)

    def state_cost(self, X, p, weight):
        """`_c` / `_phi` mppi_race_car.py:137-155 (obstacle variant :147-171) for states
        X[...,4] with the waypoint search frozen at ``p``."""
        x, y, yaw, vel = X[..., 0], X[..., 1], X[..., 2], X[..., 3]
        two_pi = F32(2.0 * np.pi)
        yaw_w = (yaw + two_pi) % two_pi
        i = self.nearest_waypoint(x, y, p)
        R = self.ref_path
        c = (weight[0] * (x - R[i, 0]) ** 2 + weight[1] * (y - R[i, 1]) ** 2
             + weight[2] * (yaw_w - R[i, 2]) ** 2 + weight[3] * (vel - R[i, 3]) ** 2)
        if self.obstacle_circles is not None:
            c = c + self.collided(x, y, yaw) * F32(1.0e10)
        return c.astype(F32)

    def compute_weight(self, S):
        """`_compute_weight` mppi_race_car.py:199-209."""
        rho = S.min()
        e = np.exp(F32(-1.0 / self.param_lambda) * (S - rho))
        eta = e.sum()
        return (F32(1.0) / eta) * e

    def iteration(self, observed_x, epsilon):
        """`_calc_control_input` mppi_race_car.py:55-121 / obstacle :65-131."""
        K, T = self.K, self.T
        u = self.u_prev.copy()
        x0 = np.asarray(observed_x).astype(F32)
        eps = np.asarray(epsilon).astype(F32)
        out = {"idx_before": self.prev_waypoints_idx}
        p = int(self.nearest_waypoint(x0[0], x0[1], self.prev_waypoints_idx))
        self.prev_waypoints_idx = p
        out["idx_start"] = out["idx_after"] = p
        out["path_end"] = bool(p >= self.ref_path.shape[0] - 1)
        if out["path_end"] and self.raise_at_path_end:
            raise IndexError("[ERROR] Reached the end of the reference path.")

        thr = exploit_threshold(self.param_exploration, K)
        exploit = (np.arange(K) < thr)[:, None, None]
        v = self.clamp(np.where(exploit, u[None] + eps, eps)).astype(F32)
        X = self.rollout(x0, v)
        c = self.state_cost(X, p, self.stage_cost_weight)  # [K,T]
        sinv = np.linalg.inv(self.Sigma)
        q = np.matmul(sinv, v[..., None])[..., 0]  # inv(Sigma) @ v  -> [K,T,2]
        ctrl = F32(self.param_gamma) * (u[None, :, 0] * q[..., 0] + u[None, :, 1] * q[..., 1])
        S = sequential_sum((c + ctrl).astype(F32), axis=1)  # S[k] += ... in t order (:84)
        S = (S + self.state_cost(X[:, -1], p, self.terminal_cost_weight)).astype(F32)
        out["S"] = S
        w = self.compute_weight(S)
        out["w"] = w
        w_eps = sequential_sum((w[:, None, None] * eps).astype(F32), axis=0)
        out["w_eps_raw"] = w_eps
        w_eps = self.filter_fn(w_eps, 10)
        out["w_eps_filtered"] = w_eps
        u = (u + w_eps).astype(F32)
        out["u_pre_clamp"] = u.copy()
        if self.visualize_optimal_traj:  # :102-106 clamps u in place through `_g(u[t-1])`
            u = self.clamp(u)
        out["u_pre_shift"] = u.copy()
        self.u_prev[:-1] = u[1:]
        self.u_prev[-1] = u[-1]
        out["u_returned"] = self.u_prev.copy()
        out["u0_returned"] = self.u_prev[0].copy()
        out["v"] = v
        out["X"] = X
        return out

    def viz_trajectories(self, x0, u_pre_shift, v):
        T = self.T
        order = (np.arange(T) - 1) % T
        x0 = np.asarray(x0).astype(F32)
        opt = self.rollout(x0, self.clamp(u_pre_shift[None, order]))[0]
        smp = self.rollout(x0, v[:, order])
        return opt, smp


# --------------------------------------------------------------------------------------
# Plants and path generators (drivers)
# --------------------------------------------------------------------------------------


def diffdrive_plant_step(state, u, dt):
    """`DifferentialDrive.update_state` mppi_differential_drive.py:33-40."""
    x, y, yaw = state
    return np.array([x + u[0] * np.cos(yaw) * dt, y + u[0] * np.sin(yaw) * dt, yaw + u[1] * dt])


def racecar_plant_step(state, u, dt, wheel_base=2.5, max_steer_abs=0.523, max_accel_abs=2.0):
    """`Vehicle.update` models/vehicle.py:85-114: the kinematic bicycle with the controls clipped first; f64 state
    (pinned by tests/golden/plant_rc_vehicle.npz)."""
    x, y, yaw, v = np.asarray(state, np.float64)
    steer = np.clip(u[0], -max_steer_abs, max_steer_abs)
    accel = np.clip(u[1], -max_accel_abs, max_accel_abs)
    return np.array([x + v * np.cos(yaw) * dt, y + v * np.sin(yaw) * dt, yaw + v / wheel_base * np.tan(steer) * dt,
                     v + accel * dt])


def generate_point_trajectory(start_point, end_point, num_points=100):
    """mppi_differential_drive.py:385-389."""
    x = np.linspace(start_point[0], end_point[0], num_points)
    y = np.linspace(start_point[1], end_point[1], num_points)
    yaw = np.arctan2(end_point[1] - start_point[1], end_point[0] - start_point[0]) * np.ones(num_points)
    return np.array([x, y, yaw]).T


def generate_lemniscate_racecar(num_points, radius):
    """mppi_race_car_obstacle.py:288-299 (f32 linspace)."""
    t = np.linspace(0, 2 * np.pi, num_points, dtype=np.float32)
    a = radius
    x = a * np.cos(t) / (1 + np.sin(t) ** 2)
    y = a * np.sin(t) * np.cos(t) / (1 + np.sin(t) ** 2)
    yaw = np.arctan2(np.gradient(y), np.gradient(x))
    v = np.ones_like(t) * 5.0
    return np.stack([x, y, yaw, v], axis=1)


# --------------------------------------------------------------------------------------
# K-sharding: per-shard softmin record and its merge, f64
# --------------------------------------------------------------------------------------


def softmin_partial(S, eps, beta):
    """Record {rho, eta, eta2, W[T,2]} of one shard: W = sum_k exp(-beta (S_k - rho)) eps_k."""
    S = np.asarray(S, np.float64)
    eps = np.asarray(eps, np.float64)
    rho = S.min()
    e = np.exp(-beta * (S - rho))
    W = (e[:, None, None] * eps).sum(axis=0)
    return np.concatenate([[rho, e.sum(), (e * e).sum()], W.reshape(-1)])


def merge_partials(records, beta):
    """Merge shard records with the rescale trick; returns (rho, eta, ess, w_eps[T,2]) equal to the
    unsharded `_compute_weight` + weighted sum (mppi_race_car.py:199-209, :93-95)."""
    r = np.asarray(records, np.float64)
    rho = r[:, 0].min()
    s = np.exp(-beta * (r[:, 0] - rho))
    eta = (s * r[:, 1]).sum()
    eta2 = (s * s * r[:, 2]).sum()
    W = (s[:, None] * r[:, 3:]).sum(axis=0)
    return rho, eta, eta * eta / eta2, (W / eta).reshape(-1, 2)


# --------------------------------------------------------------------------------------
# BASELINE config 5: the diff-drive MPPI loop with learned residual dynamics (build-defined composition,
# SURVEY.md section 3.3): x_{t+1} = x_t + dt * (f(x_t, v_t) + MLP([x_t, v_t])), f = [v cos(yaw), v sin(yaw), w]
# (test/bullet_differential_drive_dnn.py:79-92); MLP = train/train_diff_mlp.py:13-36
# (Linear(5,512) -> 3 x [Linear(512,512) + tanh] -> Linear(512,3); the first layer has no activation).
# --------------------------------------------------------------------------------------


def mlp_forward(weights, z):
    """`MultiLayerPerceptron.forward` train/train_diff_mlp.py:31-36 in f64.  weights: dict with the
    checkpoint's key layout (input_layer.*, hidden_layer.{0,1,2}.*, out_layer.*); z: [..., 5]."""
    w = {k: np.asarray(v, np.float64) for k, v in weights.items()}
    h = z @ w["input_layer.weight"].T + w["input_layer.bias"]
    i = 0
    while f"hidden_layer.{i}.weight" in w:
        h = np.tanh(h @ w[f"hidden_layer.{i}.weight"].T + w[f"hidden_layer.{i}.bias"])
        i += 1
    return h @ w["out_layer.weight"].T + w["out_layer.bias"]


def random_mlp_weights(seed=0, hidden=512, n_hidden=3, out_scale=0.05):
    """Random weights of the reference architecture, magnitudes like saved_models/mlp_diff_300x100_3l.pth
    (SURVEY.md App. C); the checkpoint itself cannot travel to the GPU box."""
    rng = np.random.default_rng(seed)
    w = {"input_layer.weight": rng.normal(0, 0.2, (hidden, 5)).astype(np.float32),
         "input_layer.bias": rng.normal(0, 0.15, hidden).astype(np.float32)}
    for i in range(n_hidden):
        w[f"hidden_layer.{i}.weight"] = rng.normal(0, 0.045, (hidden, hidden)).astype(np.float32)
        w[f"hidden_layer.{i}.bias"] = rng.normal(0, 0.02, hidden).astype(np.float32)
    w["out_layer.weight"] = rng.normal(0, out_scale / np.sqrt(hidden) * 4, (3, hidden)).astype(np.float32)
    w["out_layer.bias"] = rng.normal(0, 0.01, 3).astype(np.float32)
    return w


class DiffDriveMlpOracle(DiffDriveOracle):
    """DiffDriveOracle with `_state_transition` replaced by the residual model; cost, waypoint index,
    weights, filter and shift are the reference's (inherited)."""

    def __init__(self, *a, mlp_weights=None, **kw):
        super().__init__(*a, **kw)
        self.mlp_weights = mlp_weights

    def rollout(self, x0, v):
        K, T = v.shape[:2]
        dt = self.delta_t
        X = np.empty((K, T, 3))
        s = np.tile(np.asarray(x0, np.float64), (K, 1))
        for t in range(T):
            z = np.concatenate([s, v[:, t]], axis=1)
            r = mlp_forward(self.mlp_weights, z)
            f = np.stack([v[:, t, 0] * np.cos(s[:, 2]), v[:, t, 0] * np.sin(s[:, 2]), v[:, t, 1]], axis=1)
            s = s + dt * (f + r)
            X[:, t] = s
        return X
