"""`pytorch_mppi.MPPI` call surface with BUILT-IN dynamics / running-cost models.

The reference's test/test_mppi.py, test/test_mppi_diff.py, test/test_mppi_diff_dyna.py and
train/bullet_mppi_differential_drive.py drive ``pytorch_mppi.MPPI(dynamics, running_cost, nx, noise_sigma, num_samples=...,
horizon=..., lambda_=..., u_min=..., u_max=...)`` with Python callbacks.  Here ``dynamics`` and ``running_cost`` NAME a
model evaluated on the GPU (libmppi_hip.so, ``mppi_cb_*``):

    from dnn_mppi_mpc_amd.callback_mppi import MPPI, RunningCost
    ctrl = MPPI("unicycle", RunningCost.static_obstacles(), 3, noise_sigma, num_samples=1000, horizon=25, lambda_=1.0,
                u_min=[-2, -2], u_max=[2, 2])
    action = ctrl.command(state)

``command`` / ``U`` / ``cost_total`` / ``get_trajectories``-style access mirror the library's.  The loop follows the
published algorithm; the library is absent here and un-pinned by the reference, so that part is parity-unpinned
(include/mppi_hip.h).  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi

DYNAMICS = {"unicycle": (0, 3, 2), "skid_steer": (1, 5, 4)}


class MppiCbConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("dynamics", C.c_int32), ("obstacle_kind", C.c_int32),
                ("K", C.c_int32), ("T", C.c_int32), ("n_obs", C.c_int32), ("sample_null_action", C.c_int32),
                ("dt", C.c_double), ("lambda_", C.c_double), ("noise_sigma", C.c_double * 16),
                ("u_min", C.c_double * 4), ("u_max", C.c_double * 4), ("u_init", C.c_double * 4),
                ("goal", C.c_double * 5), ("q_diag", C.c_double * 5), ("r_diag", C.c_double * 4),
                ("obstacles", C.c_double * 64), ("safety_distance", C.c_double), ("obstacle_weight", C.c_double),
                ("skid_params", C.c_double * 5), ("seed", C.c_uint64)]


class RunningCost:
    """(s - goal)^T diag(Q) (s - goal) + u^T diag(R) u + weight * sum over obstacles of the obstacle term."""

    def __init__(self, goal, q_diag, r_diag, obstacles, safety_distance, obstacle_weight, kind):
        self.goal, self.q_diag, self.r_diag = map(lambda a: np.asarray(a, float), (goal, q_diag, r_diag))
        self.obstacles = np.asarray(obstacles, float).reshape(-1, 4)  # x, y, vx, vy (per horizon step)
        self.safety_distance, self.obstacle_weight, self.kind = float(safety_distance), float(obstacle_weight), kind

    @classmethod
    def static_obstacles(cls):
        """test/test_mppi.py:30-50."""
        return cls([6.0, 6.0, 1.57], [20, 5, 9], [0.1, 0.1], [[5.0, 4.0, 0, 0], [3.5, 3.5, 0, 0]], 0.8, 10.0, "inverse")

    @classmethod
    def moving_obstacles(cls):
        """test/test_mppi_diff.py:25-52."""
        return cls([5.0, 4.0, 1.57], [10.0, 5.0, 9.0], [1.0, 10.0], [[5.0, 5.0, 0.0, -0.05], [7.0, 3.0, 0.0, 0.05]], 1.0, 1.0,
                   "exponential")

    @classmethod
    def skid_steer(cls):
        """test/test_mppi_diff_dyna.py:44-64."""
        return cls([6.0, 6.0, 1.57, 2.0, 0.0], [200, 200, 20, 10, 20], [0.001] * 4, [[5.0, 4.0, 0, 0], [3.5, 3.5, 0, 0]], 0.8,
                   10.0, "inverse")


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float64)


class MPPI:
    def __init__(self, dynamics, running_cost, nx, noise_sigma, num_samples=100, horizon=15, device=0,
                 terminal_state_cost=None, lambda_=1.0, noise_mu=None, u_min=None, u_max=None, u_init=None, U_init=None,
                 u_scale=1, u_per_command=1, step_dependent_dynamics=False, rollout_samples=1, rollout_var_cost=0,
                 rollout_var_discount=0.95, sample_null_action=False, noise_abs_cost=False, dt=None, seed=0):
        if dynamics not in DYNAMICS:
            raise ValueError(f"dynamics must name a built-in model: {sorted(DYNAMICS)}")
        if not isinstance(running_cost, RunningCost):
            raise ValueError("running_cost must be a RunningCost (built-in quadratic + obstacle model)")
        if terminal_state_cost is not None or noise_mu is not None or u_scale != 1 or u_per_command != 1 or \
                rollout_samples != 1 or noise_abs_cost:
            raise NotImplementedError("only the options the reference's callers use are built")
        dyn, self.nx, self.nu = DYNAMICS[dynamics]
        if int(nx) != self.nx:
            raise ValueError(f"{dynamics} has nx = {self.nx}")
        sigma = _np(noise_sigma).reshape(self.nu, self.nu)
        self.K, self.T = int(num_samples), int(horizon)
        self.lib = capi.load_library()
        c = MppiCbConfig()
        c.struct_size = C.sizeof(MppiCbConfig)
        if hasattr(device, "index"):
            device = device.index or 0
        c.device, c.dynamics, c.K, c.T = int(device) if not isinstance(device, str) else 0, dyn, self.K, self.T
        c.obstacle_kind = 0 if running_cost.kind == "inverse" else 1
        c.n_obs, c.sample_null_action = running_cost.obstacles.shape[0], int(bool(sample_null_action))
        c.dt = float(dt) if dt is not None else (0.05 if dynamics == "unicycle" else 0.02)  # the callers' constants
        c.lambda_ = float(lambda_)
        for i in range(self.nu):
            for j in range(self.nu):
                c.noise_sigma[4 * i + j] = sigma[i, j]
        lo = _np(u_min) if u_min is not None else np.full(self.nu, -1e30)
        hi = _np(u_max) if u_max is not None else np.full(self.nu, 1e30)
        ui = _np(u_init) if u_init is not None else np.zeros(self.nu)
        for i in range(self.nu):
            c.u_min[i], c.u_max[i], c.u_init[i], c.r_diag[i] = lo[i], hi[i], ui[i], running_cost.r_diag[i]
        for i in range(self.nx):
            c.goal[i], c.q_diag[i] = running_cost.goal[i], running_cost.q_diag[i]
        for m, row in enumerate(running_cost.obstacles):
            for q in range(4):
                c.obstacles[4 * m + q] = row[q]
        c.safety_distance, c.obstacle_weight = running_cost.safety_distance, running_cost.obstacle_weight
        for i, v in enumerate((2.0, 0.05, 0.1, 0.4, 0.1)):  # test/test_mppi_diff_dyna.py:15-19, :29-30
            c.skid_params[i] = v
        c.seed = int(seed)
        self._h = C.c_void_p()
        rc = self.lib.mppi_cb_create(C.byref(c), C.byref(self._h))
        if rc:
            msg = self.lib.mppi_cb_last_error(None)
            self._h = None
            raise capi.MppiError(rc, msg.decode() if msg else "mppi_cb_create failed")
        self.cfg = c
        if U_init is not None:
            self.U = U_init

    def _ck(self, rc):
        if rc:
            msg = self.lib.mppi_cb_last_error(self._h)
            raise capi.MppiError(rc, msg.decode() if msg else "unknown error")

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.mppi_cb_destroy(self._h)
            self._h = None

    @property
    def U(self):
        u = np.empty((self.T, self.nu))
        self._ck(self.lib.mppi_cb_get_nominal(self._h, u.ctypes.data_as(C.POINTER(C.c_double))))
        return u

    @U.setter
    def U(self, value):
        u = np.ascontiguousarray(_np(value).reshape(self.T, self.nu))
        self._ck(self.lib.mppi_cb_set_nominal(self._h, u.ctypes.data_as(C.POINTER(C.c_double))))

    def command(self, state, shift_nominal_trajectory=True, noise=None):
        """`MPPI.command(state)` -> the first action of the updated nominal sequence.  ``noise``: optional CUDA float32
        tensor [K, T, nu] in place of the in-kernel Philox draw."""
        s = np.ascontiguousarray(_np(state).reshape(self.nx))
        eps = None
        if noise is not None:
            if not noise.is_cuda or not noise.is_contiguous() or tuple(noise.shape) != (self.K, self.T, self.nu):
                raise ValueError(f"noise must be a contiguous CUDA float32 tensor [{self.K}, {self.T}, {self.nu}]")
            eps = C.c_void_p(noise.data_ptr())
        a = np.empty(self.nu)
        self._ck(self.lib.mppi_cb_command(self._h, s.ctypes.data_as(C.POINTER(C.c_double)), eps, int(bool(shift_nominal_trajectory)),
                                          a.ctypes.data_as(C.POINTER(C.c_double)), None))
        return a

    def _costs(self, which):
        out = np.empty(self.K)
        p = out.ctypes.data_as(C.POINTER(C.c_double))
        self._ck(self.lib.mppi_cb_get_costs(self._h, p if which == 0 else None, p if which == 1 else None))
        return out

    @property
    def cost_total(self):
        return self._costs(0)

    @property
    def omega(self):
        return self._costs(1)

    def _eval(self, what, states, actions, t):
        s = np.ascontiguousarray(_np(states).reshape(-1, self.nx))
        a = np.ascontiguousarray(_np(actions).reshape(-1, self.nu))
        out = np.empty((s.shape[0], self.nx) if what == 0 else s.shape[0])
        D = C.POINTER(C.c_double)
        self._ck(self.lib.mppi_cb_eval(self._h, what, s.ctypes.data_as(D), a.ctypes.data_as(D), int(t), s.shape[0],
                                       out.ctypes.data_as(D)))
        return out

    def _dynamics(self, state, u, t=0):
        return self._eval(0, state, u, t)

    def _running_cost(self, state, u, t=0):
        return self._eval(1, state, u, t)

    def get_optimal_trajectory(self, state):
        """The states the nominal sequence drives through (the `optimal_traj` of the callers' get_trajectories)."""
        s = np.ascontiguousarray(_np(state).reshape(self.nx))
        tr = np.empty((self.T, self.nx))
        D = C.POINTER(C.c_double)
        self._ck(self.lib.mppi_cb_nominal_trajectory(self._h, s.ctypes.data_as(D), tr.ctypes.data_as(D)))
        return tr
