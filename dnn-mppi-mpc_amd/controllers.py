"""Host-side mirror of the reference's MPPI controller classes.

Same class names, constructor keywords, method names, return tuples, mutable attributes and
error behaviour as

* ``MPPIAlgorithms``          controllers/mppi_differential_drive.py:42-289 (+ ``_obs.py``,
  and the ``_cuda.py`` / ``_torch.py`` variants this engine replaces)
* ``MPPIRacecarController``   controllers/mppi_race_car.py:9-256 (+ ``_obstacle.py``,
  ``_cupy.py`` / ``_torch.py``)

but the body of ``_calc_input_control`` / ``_calc_control_input`` (sample -> rollout -> cost
-> softmin weight -> reduce -> filter -> shift) runs in ``libmppi_hip.so`` on an MI355X.
There is no CPU fallback: constructing a controller without the library or a GPU raises.
(file:line citations are relative to the reference repository.)
"""
from __future__ import annotations

import numpy as np

from . import _capi as capi
from .distributed import exchange_partials, shard_range
from .engine import Engine


def _num(x):
    """Python float from a float / NumPy scalar / 0-d tensor (mppi_differential_drive_torch.py:380-391
    passes every scalar as a 0-d tensor)."""
    if hasattr(x, "detach"):
        x = x.detach().cpu()
    return float(x)


def _int(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu()
    return int(x)


def _arr(x, dtype=np.float64):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.array(x, dtype=dtype)


class _ControllerBase:
    """State shared by both families: engine handle, aliased u_prev, waypoint index."""

    _idx_name = "prev_way_point_idx"
    _ref_dtype = np.float64

    def _finish_init(self, cfg, ref_path, obstacle_circles, precision, device, seed, process_group):
        cfg.update(precision=capi.PREC_F64 if precision in ("f64", "float64") else capi.PREC_F32,
                   device=int(device), seed=int(seed), collision_penalty=1.0e10, filter_window=10)
        self._pg = process_group
        self._sharded = process_group is not None  # split step + all-gather, even on a 1-rank group
        self._world, self._rank = 1, 0
        if process_group is not None:
            import torch.distributed as dist
            self._world, self._rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        K = cfg["K"]
        if self._world > 1:  # contiguous K/world shard per rank, SURVEY.md section 8e
            k0, k_local = shard_range(K, self._rank, self._world)
            cfg.update(K=k_local, K_global=K, k_offset=k0)
        self._engine = Engine(**cfg)
        self._ref_path = None
        self.ref_path = ref_path
        self._obstacles = None
        if obstacle_circles is not None:
            self.obstacle_circles = obstacle_circles
        self._u_host = np.zeros((self.T, self.dim_u))  # `u_prev`, mppi_differential_drive.py:82
        self._u_dev_copy = self._u_host.copy()
        self._idx_dev = 0
        setattr(self, "_idx_host", 0)
        self._zero_opt = None
        self._zero_smp = None
        self._partial = None
        self._gathered = None
        self.last_stats = None
        self.exchange = "collective" if self._sharded else "none"
        if self._sharded:
            self._setup_exchange()

    def _setup_exchange(self):
        """K sharded over several GPUs: the carrier of the one exchange per iteration, agreed by all ranks --
        "p2p": the per-rank record travels inside the finalize kernel (IPC-mapped buffers + flags, include/mppi_hip.h
        `mppi_comm_connect`), no host call or collective launch per iteration; else "rccl": one ncclAllGather per
        iteration enqueued by the library itself (`mppi_comm_init`, NCCL backend only: RCCL needs one GPU per rank);
        else "collective": the split step around `torch.distributed.all_gather_into_tensor`.
        ``MPPI_EXCHANGE=rccl|collective`` starts further down the list."""
        import os

        import torch
        import torch.distributed as dist
        want = os.environ.get("MPPI_EXCHANGE", "p2p").lower()
        if want not in ("p2p", "rccl", "collective"):
            raise ValueError("MPPI_EXCHANGE must be 'p2p', 'rccl' or 'collective'")
        eng = self._engine
        ok, handle = want == "p2p" and self._world > 1, b""
        if ok:
            try:
                handle = eng.comm_export(self._world)
            except Exception:  # noqa: BLE001 - any failure here means "use the collective"
                ok = False
        cpu = dist.get_backend(self._pg) != "nccl"
        flag_dev = "cpu" if cpu else f"cuda:{eng.cfg.device}"

        def all_ok(v):  # every rank takes the same branch
            t = torch.tensor([1 if v else 0], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self._pg)
            return bool(t.item())

        ok = all_ok(ok)
        if ok:
            handles = [None] * self._world
            if cpu:
                dist.all_gather_object(handles, handle, group=self._pg)
            else:  # (the object collectives of the NCCL backend stage through the CURRENT device: make it this rank's)
                with torch.cuda.device(eng.cfg.device):
                    dist.all_gather_object(handles, handle, group=self._pg)
            try:
                eng.comm_connect(self._rank, handles)
            except Exception:  # noqa: BLE001
                ok = False
            ok = all_ok(ok)
        if ok:
            if cpu:
                dist.barrier(group=self._pg)
            else:
                with torch.cuda.device(eng.cfg.device):
                    dist.barrier(group=self._pg)
            try:
                for _ in range(4):  # both slots, each reused once: records AND flags must arrive fresh
                    eng.comm_probe()
            except Exception:  # noqa: BLE001
                ok = False
            ok = all_ok(ok)
        if ok:
            self.exchange = "p2p"
            return
        if handle:
            eng.comm_close()
        if want == "collective" or cpu:
            return
        # RCCL inside the library: every rank must be able to load librccl before any of them enters the (blocking)
        # communicator set-up
        uid = b""
        try:
            uid = eng.comm_unique_id()
            ok = True
        except Exception:  # noqa: BLE001
            ok = False
        if not all_ok(ok):
            return
        box = [uid if self._rank == 0 else None]
        with torch.cuda.device(eng.cfg.device):
            dist.broadcast_object_list(box, src=dist.get_global_rank(self._pg, 0), group=self._pg)
        try:
            eng.comm_init(box[0], self._rank, self._world)
            ok = True
        except Exception:  # noqa: BLE001
            ok = False
        if all_ok(ok):
            self.exchange = "rccl"
        else:
            eng.comm_close()

    # -- mutable attributes of the reference ---------------------------------------------------------
    @property
    def ref_path(self):
        return self._ref_path

    @ref_path.setter
    def ref_path(self, value):  # callers re-assign it (mppi_race_car.py:267)
        self._ref_path = _arr(value).astype(self._ref_dtype)  # race car: `ref_path.astype(np.float32)`
        self._engine.set_ref_path(self._ref_path.astype(np.float64))

    @property
    def obstacle_circles(self):
        return self._obstacles

    @obstacle_circles.setter
    def obstacle_circles(self, value):
        self._obstacles = _arr(value).reshape(-1, 3)
        self._engine.set_obstacles(self._obstacles)

    @property
    def u_prev(self):
        """The nominal control sequence.  Like the reference's attribute it is one array object that
        the controller keeps mutating, and the sequence returned by the iteration aliases it (:165)."""
        return self._u_host

    @u_prev.setter
    def u_prev(self, value):
        self._u_host[...] = _arr(value)

    def _get_idx(self):
        return self._idx_host

    def _set_idx(self, v):
        self._idx_host = int(v)

    def _sync_state_to_device(self):
        if not np.array_equal(self._u_host, self._u_dev_copy):  # user wrote into u_prev
            self._engine.set_u_prev(self._u_host)
            self._u_dev_copy[...] = self._u_host
        if self._idx_host != self._idx_dev:
            self._engine.set_waypoint_idx(self._idx_host)
            self._idx_dev = self._idx_host

    def restart_episode(self, x0):
        """Back to the initial condition of the reference driver (a fresh controller and the robot at ``x0``,
        mppi_differential_drive.py:393-441): nominal controls zero, waypoint index 0, and ``x0`` as the state of the
        device-resident plant of ``run_closed_loop*``.  The noise counter keeps running (fresh noise every episode)."""
        self._u_host[...] = 0.0
        self._u_dev_copy[...] = 0.0
        self._idx_host = self._idx_dev = 0
        self._engine.set_u_prev(self._u_host)
        self._engine.set_waypoint_idx(0)
        self._engine.set_state(_arr(x0))

    # -- stage S1 ---------------------------------------------------------------------------------------
    def _calc_epsilon(self, sigma, size_sample, size_time_step, size_dim_u):
        """`_calc_epsilon` (mppi_differential_drive.py:273-283): the engine's Philox sampler for the
        current iteration as a CUDA float32 tensor [K,T,2].  Assign/override this method to inject noise
        (NumPy array or tensor); left alone, the iteration draws the same numbers in-kernel and never
        materialises them."""
        sigma = _arr(sigma)
        if sigma.shape[0] != sigma.shape[1] or sigma.shape[0] != size_dim_u or size_dim_u < 1:
            print("[ERROR] sigma must be a square matrix with the size of size_dim_u.")
            raise ValueError
        return self._engine.sample_epsilon(self._engine.stats.iteration)

    def _epsilon_is_overridden(self):
        return "_calc_epsilon" in self.__dict__ or type(self)._calc_epsilon is not _ControllerBase._calc_epsilon

    def _device_eps(self):
        if not self._epsilon_is_overridden():
            return None
        import torch
        eps = self._calc_epsilon(self.Sigma, self._engine.K if self._world == 1 else self.K, self.T, self.dim_u)
        if not hasattr(eps, "is_cuda"):
            eps = torch.from_numpy(np.ascontiguousarray(eps, dtype=np.float32))
        eps = eps.to(device=f"cuda:{self._engine.cfg.device}", dtype=torch.float32)
        if self._world > 1 and eps.shape[0] == self.K:  # a global tensor was injected: take this rank's rows
            o = int(self._engine.cfg.k_offset)
            eps = eps[o:o + self._engine.K]
        return eps.contiguous()

    # -- the iteration -------------------------------------------------------------------------------------
    def _iterate(self, observed_x):
        x0 = _arr(observed_x).reshape(-1)
        self._sync_state_to_device()
        eps = self._device_eps()
        if not self._sharded:
            u, u0, st = self._engine.step(x0, eps)
        else:
            u, u0, st = self._sharded_step(x0, eps)
        self._u_host[...] = u  # in place: the returned sequence aliases u_prev
        self._u_dev_copy[...] = u
        self._idx_host = self._idx_dev = int(st.idx_after)
        self.last_stats = st
        self._last_eps = eps  # keep the tensor alive for the viz rollouts
        return st

    def _exchange_buffers(self):
        import torch
        if self._partial is None:
            dev = f"cuda:{self._engine.cfg.device}"
            n = self._engine.partial_len()
            self._partial = torch.empty(n, dtype=torch.float64, device=dev)
            self._gathered = torch.empty(self._world * n, dtype=torch.float64, device=dev)
        return self._partial, self._gathered

    def _all_gather_partials(self):
        """The one exchange step of an iteration (RCCL all-gather of 3+2T doubles per rank)."""
        part, gath = self._exchange_buffers()
        return exchange_partials(part, self._world, self._pg, out=gath)

    def _sharded_step(self, x0, eps):
        """K sharded over the ranks of ``process_group``: one all-gather per iteration, every rank finishes
        the iteration identically."""
        import torch
        if self.exchange in ("p2p", "rccl"):  # the library exchanges the records itself
            return self._engine.step(x0, eps)
        part, _ = self._exchange_buffers()
        stream = torch.cuda.current_stream()
        self._engine.step_begin(x0, eps, part, stream)
        gath = self._all_gather_partials()
        return self._engine.step_end(gath, self._world, stream)

    def run_closed_loop_sharded(self, n_iters):
        """n iterations with the driver's plant on the device and K sharded over the ranks; nothing but
        the all-gather leaves the GPU, one synchronisation at the end."""
        import ctypes as C

        import torch
        import torch.distributed as dist
        if self.exchange in ("p2p", "rccl"):  # nothing leaves the GPUs until the last iteration is done
            _, st = self._engine.run_closed_loop(int(n_iters))
            u = self._engine.get_u_prev()
            self._u_host[...] = u
            self._u_dev_copy[...] = u
            self._idx_host = self._idx_dev = int(st.idx_after)
            self.last_stats = st
            return st
        part, gath = self._exchange_buffers()
        stream = torch.cuda.current_stream()
        eng, lib = self._engine, self._engine.lib
        h, sp = eng._h, C.c_void_p(stream.cuda_stream)
        pp, gp = C.c_void_p(part.data_ptr()), C.c_void_p(gath.data_ptr())
        nccl = dist.get_backend(self._pg) == "nccl"
        for _ in range(int(n_iters)):  # per iteration: two ABI calls and the one collective
            rc = lib.mppi_step_begin(h, None, None, pp, sp)
            if rc:
                eng._ck(rc)
            if nccl:
                dist.all_gather_into_tensor(gath, part, group=self._pg)
            else:
                exchange_partials(part, self._world, self._pg, out=gath)
            rc = lib.mppi_step_end_async(h, gp, self._world, sp)
            if rc:
                eng._ck(rc)
        u, u0, st = eng.sync_result(stream)
        self._u_host[...] = u
        self._u_dev_copy[...] = u
        self._idx_host = self._idx_dev = int(st.idx_after)
        self.last_stats = st
        return st

    def _viz(self, want_opt, want_smp):
        nx = self.dim_x
        if self._zero_opt is None:
            self._zero_opt = np.zeros((self.T, nx))
            self._zero_smp = np.zeros((self._engine.K, self.T, nx))
        opt_t, smp_t = (None, None)
        if want_opt or want_smp:
            opt_t, smp_t = self._engine.rollout_viz(want_opt, want_smp)
        opt = opt_t.cpu().numpy().astype(np.float64) if opt_t is not None else self._zero_opt
        smp = smp_t.cpu().numpy().astype(np.float64) if smp_t is not None else self._zero_smp
        return opt, smp

    # -- the stage methods of the reference classes, batched -------------------------------
    # A single state / control (the reference's call shape) returns what the reference returns; an [n, .] array
    # evaluates n calls in one launch.  Methods that move the waypoint index in the reference (`_compute_cost`,
    # `_terminal_cost` of the diff-drive: `update_prev_idx=True`, mppi_differential_drive.py:228,:244) thread it through
    # the n calls in order and leave it in `prev_way_point_idx`, as n successive reference calls would.
    @staticmethod
    def _as_rows(a, ncol):
        a = _arr(a)
        single = a.ndim == 1
        a = a.reshape(1, -1) if single else a
        if a.ndim != 2 or a.shape[1] != ncol:
            raise ValueError(f"expected {ncol} values per row")
        return a, single

    def _transition(self, x_t, v_t):
        x, single = self._as_rows(x_t, self.dim_x)
        v, _ = self._as_rows(v_t, self.dim_u)
        self._sync_state_to_device()
        out = self._engine.eval_state_transition(x, v)
        return out[0] if single else out

    def _stage_or_terminal_cost(self, x_t, terminal, update):
        x, single = self._as_rows(x_t, self.dim_x)
        self._sync_state_to_device()
        cost, _, p = self._engine.eval_cost(x, self._idx_host, terminal=terminal, update_prev_idx=update)
        if update:
            self._idx_host = int(p)
        return float(cost[0]) if single else cost

    def _is_collided(self, x_t):
        """`_is_collided` (mppi_differential_drive_obs.py:301-313 / mppi_race_car_obstacle.py:255-274): 1.0 / 0.0."""
        x, single = self._as_rows(x_t, self.dim_x)
        hit = self._engine.eval_is_collided(x)
        return float(hit[0]) if single else hit

    def _nearest(self, x, y, update_prev_idx):
        xa, ya = np.atleast_1d(_arr(x)).reshape(-1), np.atleast_1d(_arr(y)).reshape(-1)
        single = np.ndim(x) == 0
        rows = np.zeros((xa.size, self.dim_x))
        rows[:, 0], rows[:, 1] = xa, ya
        self._sync_state_to_device()
        idx, p = self._engine.eval_nearest_waypoint(rows, self._idx_host, update_prev_idx)
        if update_prev_idx:
            self._idx_host = int(p)
        ref = self._ref_path[idx]
        cols = [idx] + [ref[:, j] for j in range(ref.shape[1])]
        return tuple(c[0] for c in cols) if single else tuple(cols)

    def _moving_average_filter(self, xx, window_size=10):
        """`_moving_average_filter` (mppi_differential_drive.py:257-271 / mppi_race_car.py:211-222 / the torch files'
        conv1d form) of a [T, 2] signal, evaluated by the finalize kernel's filter code."""
        if int(window_size) != 10:
            raise ValueError("the engine is built for the reference's window_size = 10")
        return self._engine.eval_moving_average(_arr(xx)).astype(self._ref_dtype)

    def _compute_weight(self, S=None):
        """`_compute_weight` (:167-180 / mppi_race_car.py:199-209): of ``S`` when given (the reference's signature), else of
        the last iteration's costs; evaluated on the GPU."""
        if S is None:
            return self._engine.weights()
        return self._engine.eval_weights(_arr(S))

    def _g(self, v):
        """`_g` (:285-289 / mppi_race_car.py:176-181): clamps in place like the reference and returns ``v``."""
        a, _ = self._as_rows(v, self.dim_u)
        out = self._engine.eval_clamp(a).reshape(np.shape(v))
        if isinstance(v, np.ndarray):
            v[...] = out
            return v
        return out

    def sample_costs(self):
        """S[K] of the last iteration (`S`, :103)."""
        return self._engine.costs()


class MPPIAlgorithms(_ControllerBase):
    """Differential-drive MPPI (controllers/mppi_differential_drive.py:42-289).

    ``obstacle_circles`` / ``safety_margin_rate`` select the `_obs` variant (:58-59 there).
    ``variant``: "numpy" (the CPU file, default), "cuda" (SEARCH_IDX_LEN 10 + terminal yaw wrap,
    mppi_differential_drive_cuda.py:201,:239) or "torch" (no clamp in the rollout, beta = lambda,
    terminal yaw wrap, mppi_differential_drive_torch.py:128,:187,:231).
    """

    def __init__(self, delta_t, ref_path, max_speed, max_omega, num_samples_K, num_horizons_T, param_exploration,
                 param_lambda, param_alpha, sigma, stage_cost_weight, terminal_cost_weight, obstacle_circles=None,
                 safety_margin_rate=None, visualize_optimal_traj=True, visualze_sampled_trajs=True,
                 visualize_sampled_traj=None, *, variant="numpy", precision="f32", device=0, seed=0,
                 process_group=None, waypoint_mode=None, learned_dynamics=None, learned_scalers=None):
        if visualize_sampled_traj is not None:  # the torch variant's spelling (:63-64)
            visualze_sampled_trajs = visualize_sampled_traj
        self.delta_t = _num(delta_t)
        self.max_speed = _num(max_speed)
        self.max_omega = _num(max_omega)
        self.dim_x, self.dim_u = 3, 2
        self.T = _int(num_horizons_T)
        self.K = _int(num_samples_K)
        self.param_exploration = _num(param_exploration)
        self.param_lambda = _num(param_lambda)
        self.param_alpha = _num(param_alpha)
        self.param_gamma = self.param_lambda * (1.0 - self.param_alpha)
        self.Sigma = _arr(sigma)
        self.stage_cost_weight = _arr(stage_cost_weight)
        self.terminal_cost_weight = _arr(terminal_cost_weight)
        self.visualize_optimal_traj = bool(visualize_optimal_traj)
        self.visualze_sampled_trajs = bool(visualze_sampled_trajs)
        self.safefy_margin_rate = None if safety_margin_rate is None else _num(safety_margin_rate)
        if variant not in ("numpy", "cuda", "torch"):
            raise ValueError("variant must be 'numpy', 'cuda' or 'torch'")
        if self.T < 10:  # np.convolve(..., 'same') returns 10 samples and the assignment fails (:264)
            raise ValueError(f"could not broadcast input array from shape (10,) into shape ({self.T},)")
        if waypoint_mode is None:
            # K sharded over ranks: the reference's one index cannot travel from rank to rank inside an iteration; the index
            # still threads through every sample's own calls (:228, :244) and restarts at each sample
            waypoint_mode = "sequential" if process_group is None else "per_rollout"
        if waypoint_mode not in ("sequential", "frozen", "per_rollout"):
            raise ValueError("waypoint_mode must be 'sequential', 'per_rollout' or 'frozen'")
        if learned_dynamics is not None and hasattr(learned_dynamics, "state_dict"):
            learned_dynamics = learned_dynamics.state_dict()  # a torch module (train/train_diff_mlp.py:13-36)
        cfg = dict(
            model=capi.MODEL_DIFFDRIVE if learned_dynamics is None else capi.MODEL_DIFFDRIVE_MLP,
            K=self.K, T=self.T, delta_t=self.delta_t,
            u_max=[self.max_speed, self.max_omega], wheel_base=0.0,
            param_exploration=self.param_exploration, param_lambda=self.param_lambda, param_alpha=self.param_alpha,
            sigma=self.Sigma, stage_cost_weight=self.stage_cost_weight, terminal_cost_weight=self.terminal_cost_weight,
            beta_mode=capi.BETA_LAMBDA if variant == "torch" else capi.BETA_INV_EXPLORATION,
            accumulate_stage_cost=0,  # `S[k] =`, :124
            waypoint_mode={"sequential": capi.WAYPOINT_SEQUENTIAL, "frozen": capi.WAYPOINT_FROZEN,
                           "per_rollout": capi.WAYPOINT_PER_ROLLOUT}[waypoint_mode],
            search_window=10 if variant == "cuda" else 20,
            wrap_yaw_stage=0, wrap_yaw_terminal=0 if variant == "numpy" else 1,
            clamp_rollout=0 if variant == "torch" else 1,
            clamp_u_after_update=int(self.visualze_sampled_trajs),  # :145-149
            filter_mode=capi.FILTER_TORCH if variant == "torch" else capi.FILTER_DIFFDRIVE,  # _torch.py:252-263
            obstacle_model=capi.OBSTACLE_CIRCLE if obstacle_circles is not None else capi.OBSTACLE_NONE,
            raise_at_path_end=0,
            safety_margin=0.0 if safety_margin_rate is None else self.safefy_margin_rate,
            vehicle_w=0.0, vehicle_l=0.0,
        )
        self._finish_init(cfg, ref_path, obstacle_circles, precision, device, seed, process_group)
        self._learned = learned_dynamics is not None
        if self._learned:
            self._engine.set_mlp(learned_dynamics, learned_scalers)

    prev_way_point_idx = property(_ControllerBase._get_idx, _ControllerBase._set_idx)

    def _calc_input_control(self, observed_x):
        """One MPPI iteration (:87-165).  Returns ``(u[0], u, optimal_traj, sampled_traj_list)`` with the
        reference's aliasing: ``u`` is ``self.u_prev`` after the shift, ``u[0]`` its first row."""
        st = self._iterate(observed_x)
        if st.path_end:
            print("[ERROR] Reached the end of the reference path.")  # :98
        # both viz rollouts hang off `visualze_sampled_trajs` in the reference (:145,:154)
        want = self.visualze_sampled_trajs  # (learned dynamics too: the same loop with the transition swapped)
        opt, smp = self._viz(want, want)
        return self._u_host[0], self._u_host, opt, smp

    def _state_transition(self, x_t, v_t):
        """`_state_transition` (:182-198): Euler step of the unicycle -- of the residual model x + dt (f + MLP([x, v])) when
        one is loaded (test/bullet_differential_drive_dnn.py:79-92)."""
        return self._transition(x_t, v_t)

    def _compute_cost(self, x_t):
        """`_compute_cost` (:222-236; `_obs.py:228-244` adds the collision term); moves `prev_way_point_idx` (:228)."""
        return self._stage_or_terminal_cost(x_t, terminal=False, update=True)

    def _terminal_cost(self, x_T):
        """`_terminal_cost` (:239-249); moves `prev_way_point_idx` (:244)."""
        return self._stage_or_terminal_cost(x_T, terminal=True, update=True)

    def _get_nearest_waypoint(self, x, y, update_prev_idx=False):
        """`_get_nearest_waypoint` (:201-220) -> (nearest_idx, ref_x, ref_y, ref_yaw)."""
        return self._nearest(x, y, update_prev_idx)


class MPPIRacecarController(_ControllerBase):
    """Kinematic-bicycle MPPI (controllers/mppi_race_car.py:9-256); ``obstacle_circles`` selects the
    `_obstacle` variant, whose path-end behaviour is print-only (mppi_race_car_obstacle.py:73-74)."""

    _idx_name = "prev_waypoints_idx"
    _ref_dtype = np.float32  # mppi_race_car.py:48

    def __init__(self, delta_t=0.05, wheel_base=2.5, max_steer_abs=0.523, max_accel_abs=2.000,
                 ref_path=((0.0, 0.0, 0.0, 1.0), (10.0, 0.0, 0.0, 1.0)), horizon_step_T=10, number_of_samples_K=100,
                 param_exploration=0.01, param_lambda=50.0, param_alpha=1.0, sigma=((0.5, 0.0), (0.0, 0.1)),
                 stage_cost_weight=(50.0, 50.0, 1.0, 20.0), terminal_cost_weight=(50.0, 50.0, 1.0, 20.0),
                 obstacle_circles=None, collision_safety_margin_rat=1.5, visualize_optimal_traj=True,
                 visualze_sampled_trajs=True, *, variant="numpy", precision="f32", device=0, seed=0, process_group=None):
        if variant not in ("numpy", "cupy", "torch"):
            raise ValueError("variant must be 'numpy', 'cupy' or 'torch'")
        # mppi_race_car_cupy.py is the NumPy file with cp. for np.; mppi_race_car_torch.py differs in its moving
        # average (conv1d over the padded signal, first T outputs: the NumPy filter delayed by window/2 rows)
        self.variant = variant
        self.dim_x, self.dim_u = 4, 2
        self.T = _int(horizon_step_T)
        self.K = _int(number_of_samples_K)
        self.param_exploration = _num(param_exploration)
        self.param_lambda = _num(param_lambda)
        self.param_alpha = _num(param_alpha)
        self.param_gamma = self.param_lambda * (1.0 - self.param_alpha)
        self.Sigma = _arr(sigma, np.float32)
        self.stage_cost_weight = _arr(stage_cost_weight, np.float32)
        self.terminal_cost_weight = _arr(terminal_cost_weight, np.float32)
        self.visualize_optimal_traj = bool(visualize_optimal_traj)
        self.visualze_sampled_trajs = bool(visualze_sampled_trajs)
        self.delta_t = _num(delta_t)
        self.wheel_base = _num(wheel_base)
        self.max_steer_abs = _num(max_steer_abs)
        self.max_accel_abs = _num(max_accel_abs)
        self.vehicle_w, self.vehicle_l = 3.0, 4.0  # mppi_race_car_obstacle.py:53-54
        self.collision_safety_margin_rate = _num(collision_safety_margin_rat)
        self._has_obstacles = obstacle_circles is not None
        if self.T < 5:  # the padded 'same' convolution cannot be sliced back to T rows (mppi_race_car.py:220)
            raise ValueError(f"could not broadcast input array into shape ({self.T},)")
        cfg = dict(
            model=capi.MODEL_RACECAR, K=self.K, T=self.T, delta_t=self.delta_t,
            u_max=[self.max_steer_abs, self.max_accel_abs], wheel_base=self.wheel_base,
            param_exploration=self.param_exploration, param_lambda=self.param_lambda, param_alpha=self.param_alpha,
            sigma=self.Sigma, stage_cost_weight=self.stage_cost_weight, terminal_cost_weight=self.terminal_cost_weight,
            beta_mode=capi.BETA_INV_LAMBDA,  # mppi_race_car.py:205
            accumulate_stage_cost=1,         # `S[k] +=`, :84
            waypoint_mode=capi.WAYPOINT_FROZEN, search_window=200,  # :143,:158
            wrap_yaw_stage=1, wrap_yaw_terminal=1,  # :141,:150
            clamp_rollout=1, clamp_u_after_update=int(self.visualize_optimal_traj),  # :102-106
            filter_mode=capi.FILTER_TORCH if variant == "torch" else capi.FILTER_RACECAR,  # _torch.py:211-222
            obstacle_model=capi.OBSTACLE_OUTLINE if self._has_obstacles else capi.OBSTACLE_NONE,
            raise_at_path_end=0 if self._has_obstacles else 1,  # mppi_race_car.py:63-65 vs _obstacle.py:73-74
            safety_margin=self.collision_safety_margin_rate, vehicle_w=self.vehicle_w, vehicle_l=self.vehicle_l,
        )
        self._finish_init(cfg, ref_path, obstacle_circles, precision, device, seed, process_group)

    prev_waypoints_idx = property(_ControllerBase._get_idx, _ControllerBase._set_idx)

    def _calc_control_input(self, observed_x):
        """One MPPI iteration (mppi_race_car.py:55-121); raises IndexError at the path end like :63-65."""
        try:
            st = self._iterate(observed_x)
        except capi.MppiError as e:
            if e.code == capi.ERR_PATH_END:
                self._idx_host = self._idx_dev = self._engine.get_waypoint_idx()
                print("[ERROR] Reached the end of the reference path.")
                raise IndexError from None
            raise
        if st.path_end:
            print("[ERROR] Reached the end of the reference path.")
        opt, smp = self._viz(self.visualize_optimal_traj, self.visualze_sampled_trajs)
        return self._u_host[0], self._u_host, opt.astype(np.float32), smp.astype(np.float32)

    def _F(self, x_t, v_t):
        """`_F` (mppi_race_car.py:183-197): Euler step of the kinematic bicycle, controls [steer, accel]."""
        return self._transition(x_t, v_t).astype(np.float32)

    def _c(self, x_t):
        """`_c` (mppi_race_car.py:137-146; `_obstacle.py:147-158` adds the collision term); the index stays (:143)."""
        return self._stage_or_terminal_cost(x_t, terminal=False, update=False)

    def _phi(self, x_T):
        """`_phi` (mppi_race_car.py:148-155)."""
        return self._stage_or_terminal_cost(x_T, terminal=True, update=False)

    def get_nearest_waypoint(self, x, y, update_prev_idx=False):
        """`get_nearest_waypoint` (mppi_race_car.py:157-174) -> (nearest_idx, ref_x, ref_y, ref_yaw, ref_v)."""
        return self._nearest(x, y, update_prev_idx)
