"""ctypes loader for oracle/mppi_oracle.c (test infrastructure; see that file's header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libmppi_oracle.so")


class OracleCfg(C.Structure):
    _fields_ = [("K", C.c_int), ("T", C.c_int), ("n_ref", C.c_int), ("n_obs", C.c_int),
                ("clamp_u_after_update", C.c_int), ("reserved", C.c_int),
                ("delta_t", C.c_double), ("u_max0", C.c_double), ("u_max1", C.c_double), ("wheel_base", C.c_double),
                ("param_exploration", C.c_double), ("param_lambda", C.c_double), ("param_alpha", C.c_double),
                ("sigma", C.c_double * 4), ("stage_w", C.c_double * 4), ("term_w", C.c_double * 4),
                ("safety_margin", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "mppi_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _cfg(K, T, ref, obs, clamp, dt, umax, wheel_base, expl, lam, alpha, sigma, sw, tw, margin):
    c = OracleCfg()
    c.K, c.T, c.n_ref, c.n_obs = int(K), int(T), int(ref.shape[0]), 0 if obs is None else int(obs.shape[0])
    c.clamp_u_after_update = int(bool(clamp))
    c.delta_t, c.u_max0, c.u_max1, c.wheel_base = float(dt), float(umax[0]), float(umax[1]), float(wheel_base)
    c.param_exploration, c.param_lambda, c.param_alpha = float(expl), float(lam), float(alpha)
    c.sigma[:] = np.asarray(sigma, float).reshape(-1).tolist()
    sw, tw = np.asarray(sw, float), np.asarray(tw, float)
    for i in range(sw.size):
        c.stage_w[i], c.term_w[i] = sw[i], tw[i]
    c.safety_margin = 0.0 if margin is None else float(margin)
    return c


class DiffDriveC:
    """Same constructor keywords as the reference's MPPIAlgorithms (f64)."""

    def __init__(self, delta_t, ref_path, max_speed, max_omega, num_samples_K, num_horizons_T, param_exploration,
                 param_lambda, param_alpha, sigma, stage_cost_weight, terminal_cost_weight, obstacle_circles=None,
                 safety_margin_rate=None, visualize_optimal_traj=True, visualze_sampled_trajs=True):
        self.ref = np.ascontiguousarray(ref_path, np.float64)
        self.obs = None if obstacle_circles is None else np.ascontiguousarray(obstacle_circles, np.float64)
        self.K, self.T = int(num_samples_K), int(num_horizons_T)
        self.cfg = _cfg(self.K, self.T, self.ref, self.obs, visualze_sampled_trajs, delta_t, (max_speed, max_omega),
                        0.0, param_exploration, param_lambda, param_alpha, sigma, stage_cost_weight,
                        terminal_cost_weight, safety_margin_rate)
        self.u_prev = np.zeros((self.T, 2))
        self.prev_way_point_idx = 0

    def iteration(self, x0, eps, frozen_threads=0, per_rollout_threads=0):
        """``frozen_threads`` > 0: the frozen-waypoint-index variant on that many OpenMP threads (samples independent);
        ``per_rollout_threads`` > 0: the index threads through each sample's own calls and restarts at every sample
        (samples independent too); both 0: the reference's sequential index, one core."""
        eps = np.ascontiguousarray(eps, np.float32)
        assert eps.shape == (self.K, self.T, 2)
        x0 = np.ascontiguousarray(x0, np.float64)
        S, u0, stats, idx = np.empty(self.K), np.empty(2), np.empty(4), C.c_int(self.prev_way_point_idx)
        obs = self.obs if self.obs is not None else np.zeros(3)
        args = (C.byref(self.cfg), _p(self.ref, C.c_double), _p(obs, C.c_double), _p(x0, C.c_double),
                _p(eps, C.c_float), _p(self.u_prev, C.c_double), C.byref(idx), _p(S, C.c_double), _p(u0, C.c_double),
                _p(stats, C.c_double))
        if per_rollout_threads > 0:
            rc = lib().oracle_diffdrive_iteration_independent(*args, C.c_int(int(per_rollout_threads)), C.c_int(1))
        elif frozen_threads > 0:
            rc = lib().oracle_diffdrive_iteration_frozen(*args, C.c_int(int(frozen_threads)))
        else:
            rc = lib().oracle_diffdrive_iteration(*args)
        if rc != 0:
            raise ValueError("oracle_diffdrive_iteration failed (T < 10?)")
        self.prev_way_point_idx = idx.value
        return {"S": S, "u_returned": self.u_prev.copy(), "u0_returned": u0, "idx_after": idx.value,
                "rho": stats[0], "eta": stats[1], "idx_start": int(stats[2]), "path_end": bool(stats[3])}


class RaceCarC:
    """Same constructor keywords as the reference's MPPIRacecarController (f32)."""

    def __init__(self, delta_t=0.05, wheel_base=2.5, max_steer_abs=0.523, max_accel_abs=2.0, ref_path=None,
                 horizon_step_T=10, number_of_samples_K=100, param_exploration=0.01, param_lambda=50.0,
                 param_alpha=1.0, sigma=((0.5, 0.0), (0.0, 0.1)), stage_cost_weight=(50.0, 50.0, 1.0, 20.0),
                 terminal_cost_weight=(50.0, 50.0, 1.0, 20.0), obstacle_circles=None,
                 collision_safety_margin_rat=1.5, visualize_optimal_traj=True, visualze_sampled_trajs=True):
        self.ref = np.ascontiguousarray(ref_path, np.float32)
        self.obs = None if obstacle_circles is None else np.ascontiguousarray(obstacle_circles, np.float64)
        self.K, self.T = int(number_of_samples_K), int(horizon_step_T)
        self.cfg = _cfg(self.K, self.T, self.ref, self.obs, visualize_optimal_traj, delta_t,
                        (max_steer_abs, max_accel_abs), wheel_base, param_exploration, param_lambda, param_alpha,
                        np.asarray(sigma, np.float32), np.asarray(stage_cost_weight, np.float32),
                        np.asarray(terminal_cost_weight, np.float32), collision_safety_margin_rat)
        self.u_prev = np.zeros((self.T, 2), np.float32)
        self.prev_waypoints_idx = 0

    def iteration(self, x0, eps):
        eps = np.ascontiguousarray(eps, np.float32)
        assert eps.shape == (self.K, self.T, 2)
        x0 = np.ascontiguousarray(x0, np.float32)
        S, u0, stats = np.empty(self.K, np.float32), np.empty(2, np.float32), np.empty(4)
        idx = C.c_int(self.prev_waypoints_idx)
        obs = self.obs if self.obs is not None else np.zeros(3)
        rc = lib().oracle_racecar_iteration(C.byref(self.cfg), _p(self.ref, C.c_float), _p(obs, C.c_double),
                                            _p(x0, C.c_float), _p(eps, C.c_float), _p(self.u_prev, C.c_float),
                                            C.byref(idx), _p(S, C.c_float), _p(u0, C.c_float), _p(stats, C.c_double))
        if rc != 0:
            raise ValueError("oracle_racecar_iteration failed (T < 5?)")
        self.prev_waypoints_idx = idx.value
        return {"S": S, "u_returned": self.u_prev.copy(), "u0_returned": u0, "idx_after": idx.value,
                "rho": stats[0], "eta": stats[1], "idx_start": int(stats[2]), "path_end": bool(stats[3])}
