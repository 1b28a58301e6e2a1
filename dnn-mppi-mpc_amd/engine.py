"""Thin object wrapper over the C ABI handle (one engine = one controller on one GPU)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _dev_ptr(t):
    """Device pointer of a contiguous CUDA tensor (zero-copy hand-off to the C ABI)."""
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("expected a contiguous CUDA tensor")
    return C.c_void_p(t.data_ptr())


def _stream_ptr(stream):
    if stream is None:
        return None
    return C.c_void_p(getattr(stream, "cuda_stream", stream))


class Engine:
    """Owns an ``mppi_handle``.  All arithmetic happens in libmppi_hip.so on the GPU."""

    def __init__(self, **cfg):
        self.lib = capi.load_library()
        c = capi.MppiConfig()
        c.struct_size = C.sizeof(capi.MppiConfig)
        for k, v in cfg.items():
            if k in ("u_max", "sigma", "stage_cost_weight", "terminal_cost_weight"):
                arr = np.asarray(v, dtype=np.float64).reshape(-1)
                field = getattr(c, k)
                for i in range(len(field)):
                    field[i] = float(arr[i]) if i < arr.size else 0.0
            else:
                setattr(c, k, v)
        self.cfg = c
        self.K, self.T = int(c.K), int(c.T)
        self.nx = 4 if c.model == capi.MODEL_RACECAR else 3
        # several independent problems in one handle: the state / control / cost accessors then carry a leading
        # [n_agents] axis and run_closed_loop advances all of them in one launch per stage
        self.n_agents = max(1, int(c.n_agents))
        self._lead = (self.n_agents,) if self.n_agents > 1 else ()
        self._h = capi._H()
        rc = self.lib.mppi_create(C.byref(c), C.byref(self._h))
        if rc != capi.OK:
            self._h = None
            capi.check(self.lib, None, rc)
        self.stats = capi.MppiStats()
        # buffers and ctypes arguments of the per-iteration call, built once (every `.ctypes.data_as` costs ~1 us)
        self._x0_buf = np.zeros(self.nx)
        self._u_buf, self._u0_buf = np.empty((self.T, 2)), np.empty(2)
        self._step_args = (_dp(self._x0_buf), _dp(self._u_buf), _dp(self._u0_buf), C.byref(self.stats))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mppi_destroy(self._h)
            self._h = None

    __del__ = close

    def _ck(self, rc):
        capi.check(self.lib, self._h, rc)

    # -- configuration / state ----------------------------------------------------------------
    def set_ref_path(self, path):
        p = np.ascontiguousarray(path, dtype=np.float64)
        if p.ndim != 2:
            raise ValueError("ref_path must be 2-D")
        self._ck(self.lib.mppi_set_ref_path(self._h, _dp(p), p.shape[0], p.shape[1]))

    def set_obstacles(self, circles):
        c = np.ascontiguousarray(circles, dtype=np.float64).reshape(-1, 3)
        self._ck(self.lib.mppi_set_obstacles(self._h, _dp(c), c.shape[0]))

    def set_mlp(self, weights, scalers=None):
        """Residual-model weights in the checkpoint's key layout (``state_dict`` of the reference's
        ``MultiLayerPerceptron``, train/train_diff_mlp.py:13-36); tensors or arrays.

        ``scalers``: optional dict with the ``StandardScaler`` statistics the model was trained with
        (train/train_diff_mlp.py:72-86): ``in_mean``/``in_scale`` (5: state then control) and ``out_mean``/``out_scale``
        (3).  The affine maps are folded into the first and last Linear on the host (`mppi_set_mlp_scaled`), so the kernel
        is unchanged: MLP((z - m_in) / s_in) * s_out + m_out."""
        def arr(k):
            v = weights[k]
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            return np.ascontiguousarray(v, dtype=np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        n_hidden = sum(1 for k in weights if k.startswith("hidden_layer.") and k.endswith(".weight"))
        w_in, b_in = arr("input_layer.weight"), arr("input_layer.bias")
        wh = [arr(f"hidden_layer.{i}.weight") for i in range(n_hidden)]
        bh = [arr(f"hidden_layer.{i}.bias") for i in range(n_hidden)]
        w_out, b_out = arr("out_layer.weight"), arr("out_layer.bias")
        hidden = w_in.shape[0]
        if w_in.shape != (hidden, 5) or w_out.shape != (3, hidden) or any(w.shape != (hidden, hidden) for w in wh):
            raise ValueError("unexpected MLP shapes (expected Linear(5,H) -> n x Linear(H,H) -> Linear(H,3))")
        PP = C.POINTER(C.c_float) * max(1, n_hidden)
        if scalers is None:
            self._ck(self.lib.mppi_set_mlp(self._h, hidden, n_hidden, fp(w_in), fp(b_in), PP(*[fp(w) for w in wh]),
                                           PP(*[fp(b) for b in bh]), fp(w_out), fp(b_out)))
            return
        st = {k: np.ascontiguousarray(scalers[k], dtype=np.float64) for k in ("in_mean", "in_scale", "out_mean", "out_scale")}
        if st["in_mean"].shape != (5,) or st["in_scale"].shape != (5,) or st["out_mean"].shape != (3,) or st["out_scale"].shape != (3,):
            raise ValueError("scalers: in_mean / in_scale have 5 entries (state, control), out_mean / out_scale 3")
        self._ck(self.lib.mppi_set_mlp_scaled(self._h, hidden, n_hidden, fp(w_in), fp(b_in), PP(*[fp(w) for w in wh]),
                                              PP(*[fp(b) for b in bh]), fp(w_out), fp(b_out), _dp(st["in_mean"]),
                                              _dp(st["in_scale"]), _dp(st["out_mean"]), _dp(st["out_scale"])))

    def set_u_prev(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        if u.shape != self._lead + (self.T, 2):
            raise ValueError(f"u_prev must be {list(self._lead + (self.T, 2))}")
        self._ck(self.lib.mppi_set_u_prev(self._h, _dp(u)))

    def get_u_prev(self):
        u = np.empty(self._lead + (self.T, 2))
        self._ck(self.lib.mppi_get_u_prev(self._h, _dp(u)))
        return u

    def set_waypoint_idx(self, idx):
        self._ck(self.lib.mppi_set_waypoint_idx(self._h, int(idx)))

    def get_waypoint_idx(self):
        v = C.c_int32()
        self._ck(self.lib.mppi_get_waypoint_idx(self._h, C.byref(v)))
        return v.value

    def set_iteration(self, it):
        self._ck(self.lib.mppi_set_iteration(self._h, int(it)))

    def set_state(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape != self._lead + (self.nx,):
            raise ValueError(f"state must be {list(self._lead + (self.nx,))}")
        self._ck(self.lib.mppi_set_state(self._h, _dp(x)))

    def get_state(self):
        x = np.empty(self._lead + (self.nx,))
        self._ck(self.lib.mppi_get_state(self._h, _dp(x)))
        return x

    # -- the iteration ----------------------------------------------------------------------------
    def step(self, x0, eps=None, stream=None):
        """One MPPI iteration.  ``eps``: CUDA float32 tensor [K,T,2] or None (on-device Philox).  ``x0``: host array, or a
        float64 CUDA tensor [nx] handed over zero-copy (mppi_step_device_x0).  Returns (u[T,2] shifted, u0[2], stats)."""
        if hasattr(x0, "is_cuda") and x0.is_cuda:
            import torch
            if x0.dtype != torch.float64 or tuple(x0.shape) != (self.nx,) or not x0.is_contiguous():
                raise ValueError(f"a device x0 must be a contiguous float64 tensor with {self.nx} entries")
            if eps is not None:
                self._check_eps(eps)
            _, up, u0p, stp = self._step_args
            rc = self.lib.mppi_step_device_x0(self._h, C.c_void_p(x0.data_ptr()), _dev_ptr(eps), up, u0p, stp,
                                              _stream_ptr(stream))
            if rc:
                self._ck(rc)
            return self._u_buf.copy(), self._u0_buf.copy(), self.stats
        if np.shape(x0) != (self.nx,):
            raise ValueError(f"observed_x must have {self.nx} entries")
        self._x0_buf[:] = x0
        if eps is not None:
            self._check_eps(eps)
        xp, up, u0p, stp = self._step_args
        rc = self.lib.mppi_step(self._h, xp, _dev_ptr(eps), up, u0p, stp, _stream_ptr(stream))
        if rc:
            self._ck(rc)
        return self._u_buf.copy(), self._u0_buf.copy(), self.stats

    def _check_eps(self, eps):
        if eps is not None:
            import torch
            if eps.dtype != torch.float32 or tuple(eps.shape) != (self.K, self.T, 2):
                raise ValueError(f"eps must be float32 [{self.K}, {self.T}, 2]")

    def partial_len(self):
        n = C.c_int32()
        self._ck(self.lib.mppi_partial_len(self._h, C.byref(n)))
        return n.value

    def step_begin(self, x0, eps, partial, stream=None):
        """x0 None: continue from the state on the device (closed loop)."""
        xp = None
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64)
            xp = _dp(x0)
        self._check_eps(eps)
        self._ck(self.lib.mppi_step_begin(self._h, xp, _dev_ptr(eps), _dev_ptr(partial), _stream_ptr(stream)))

    def step_end_async(self, partials, nranks, stream=None):
        self._ck(self.lib.mppi_step_end_async(self._h, _dev_ptr(partials), int(nranks), _stream_ptr(stream)))

    def sync_result(self, stream=None):
        u, u0 = np.empty((self.T, 2)), np.empty(2)
        self._ck(self.lib.mppi_sync_result(self._h, _dp(u), _dp(u0), C.byref(self.stats), _stream_ptr(stream)))
        return u, u0, self.stats

    # -- peer-to-peer exchange of the per-rank record (include/mppi_hip.h, mppi_comm_*) -------------------
    def comm_export(self, nranks):
        """Creates this rank's exchange buffer; returns its IPC handle (bytes) for the other ranks."""
        buf = C.create_string_buffer(self.lib.mppi_comm_handle_bytes())
        self._ck(self.lib.mppi_comm_export(self._h, int(nranks), buf))
        return buf.raw

    def comm_buffer(self):
        p = C.c_void_p()
        self._ck(self.lib.mppi_comm_buffer(self._h, C.byref(p)))
        return p.value

    def comm_connect(self, rank, handles, local_ptrs=None):
        """``handles``: one IPC handle (bytes) per rank, in rank order (own entry ignored); ``local_ptrs``:
        optional device addresses of peers living in this process (entries that are None use the handle)."""
        n = len(handles)
        blob = C.create_string_buffer(b"".join(bytes(hd) for hd in handles))
        lp = None
        if local_ptrs is not None:
            lp = (C.c_void_p * n)(*[C.c_void_p(p) if p else C.c_void_p(None) for p in local_ptrs])
        self._ck(self.lib.mppi_comm_connect(self._h, int(rank), n, blob, lp))

    def comm_probe(self, stream=None):
        self._ck(self.lib.mppi_comm_probe(self._h, _stream_ptr(stream)))

    def comm_close(self):
        self._ck(self.lib.mppi_comm_close(self._h))

    # -- the same exchange carried by RCCL inside the library (include/mppi_hip.h, mppi_comm_init) ----------
    def comm_unique_id(self):
        """ncclGetUniqueId through the library (loads librccl.so.1 on first use); bytes for the other ranks."""
        buf = C.create_string_buffer(self.lib.mppi_comm_unique_id_bytes())
        rc = self.lib.mppi_comm_unique_id(buf)
        if rc != capi.OK:
            raise capi.MppiError(rc, (self.lib.mppi_last_error(None) or b"").decode() or "mppi_comm_unique_id failed")
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        """ncclCommInitRank on the handle's device: collective, every rank calls it with rank 0's id.  From then on
        ``step`` and ``run_closed_loop`` carry ONE ncclAllGather per iteration inside the library."""
        buf = C.create_string_buffer(bytes(unique_id), len(unique_id))
        self._ck(self.lib.mppi_comm_init(self._h, buf, int(rank), int(nranks)))

    def step_end(self, partials, nranks, stream=None):
        u, u0 = np.empty((self.T, 2)), np.empty(2)
        self._ck(self.lib.mppi_step_end(self._h, _dev_ptr(partials), int(nranks), _dp(u), _dp(u0),
                                        C.byref(self.stats), _stream_ptr(stream)))
        return u, u0, self.stats

    def costs(self):
        S = np.empty(self._lead + (self.K,))
        self._ck(self.lib.mppi_get_costs(self._h, _dp(S)))
        return S

    def weights(self):
        w = np.empty(self.K)
        self._ck(self.lib.mppi_get_weights(self._h, _dp(w)))
        return w

    def sample_epsilon(self, iteration, out=None, stream=None):
        """The sampler's noise of `iteration` as a tensor: float32 [K,T,2] ([n_agents,K,T,2] for a batched handle)."""
        import torch
        shape = self._lead + (self.K, self.T, 2)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=f"cuda:{self.cfg.device}")
        elif out.dtype != torch.float32 or tuple(out.shape) != shape:
            raise ValueError(f"out must be float32 {list(shape)}")
        self._ck(self.lib.mppi_sample_epsilon(self._h, int(iteration), _dev_ptr(out), _stream_ptr(stream)))
        return out

    def set_noise_ring(self, ring):
        """`_calc_epsilon` materialised for the closed loop: ``ring`` = CUDA float32 [n_slots, (n_agents,) K, T, 2], n_slots a
        power of two; iteration i reads slot i mod n_slots.  ``None`` returns to the in-kernel sampler.  The engine keeps a
        reference to the tensor."""
        if ring is None:
            self._ck(self.lib.mppi_set_noise_ring(self._h, None, 0))
            self._noise_ring = None
            return
        import torch
        want = self._lead + (self.K, self.T, 2)
        if ring.dtype != torch.float32 or tuple(ring.shape[1:]) != want:
            raise ValueError(f"the noise ring must be float32 [n_slots, {', '.join(map(str, want))}]")
        self._ck(self.lib.mppi_set_noise_ring(self._h, _dev_ptr(ring), int(ring.shape[0])))
        self._noise_ring = ring

    def rollout_viz(self, want_optimal=True, want_sampled=True, stream=None):
        import torch
        dev = f"cuda:{self.cfg.device}"
        opt = torch.empty((self.T, self.nx), dtype=torch.float32, device=dev) if want_optimal else None
        smp = torch.empty((self.K, self.T, self.nx), dtype=torch.float32, device=dev) if want_sampled else None
        self._ck(self.lib.mppi_rollout_viz(self._h, _dev_ptr(opt), _dev_ptr(smp), _stream_ptr(stream)))
        return opt, smp

    # -- batched stage methods (include/mppi_hip.h, mppi_eval_*) -------------------------------------------
    @staticmethod
    def _rows(a, ncol):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != ncol:
            raise ValueError(f"expected an [n, {ncol}] array")
        return a

    def eval_state_transition(self, x, v):
        x, v = self._rows(x, self.nx), self._rows(v, 2)
        if v.shape[0] != x.shape[0]:
            raise ValueError("x and v must have the same number of rows")
        out = np.empty_like(x)
        self._ck(self.lib.mppi_eval_state_transition(self._h, _dp(x), _dp(v), x.shape[0], _dp(out)))
        return out

    def eval_clamp(self, v):
        v = self._rows(v, 2)
        out = np.empty_like(v)
        self._ck(self.lib.mppi_eval_clamp(self._h, _dp(v), v.shape[0], _dp(out)))
        return out

    def eval_is_collided(self, x):
        x = self._rows(x, self.nx)
        out = np.empty(x.shape[0])
        self._ck(self.lib.mppi_eval_is_collided(self._h, _dp(x), x.shape[0], _dp(out)))
        return out

    def eval_nearest_waypoint(self, x, prev_idx, update_prev_idx=False):
        """Returns (idx[n], prev_idx after the calls)."""
        x = self._rows(x, self.nx)
        p, idx = C.c_int32(int(prev_idx)), np.empty(x.shape[0], dtype=np.int32)
        self._ck(self.lib.mppi_eval_nearest_waypoint(self._h, _dp(x), x.shape[0], C.byref(p), int(bool(update_prev_idx)),
                                                     idx.ctypes.data_as(C.POINTER(C.c_int32))))
        return idx, p.value

    def eval_cost(self, x, prev_idx, terminal=False, update_prev_idx=False):
        """Returns (cost[n], idx[n], prev_idx after the calls)."""
        x = self._rows(x, self.nx)
        p, idx, cost = C.c_int32(int(prev_idx)), np.empty(x.shape[0], dtype=np.int32), np.empty(x.shape[0])
        self._ck(self.lib.mppi_eval_cost(self._h, int(bool(terminal)), _dp(x), x.shape[0], C.byref(p),
                                         int(bool(update_prev_idx)), _dp(cost), idx.ctypes.data_as(C.POINTER(C.c_int32))))
        return cost, idx, p.value

    def eval_moving_average(self, xx):
        xx = np.ascontiguousarray(xx, dtype=np.float64)
        if xx.shape != (self.T, 2):
            raise ValueError(f"the filter input must be [{self.T}, 2]")
        out = np.empty_like(xx)
        self._ck(self.lib.mppi_eval_moving_average(self._h, _dp(xx), _dp(out)))
        return out

    def eval_weights(self, S):
        S = np.ascontiguousarray(S, dtype=np.float64).reshape(-1)
        w = np.empty_like(S)
        self._ck(self.lib.mppi_eval_weights(self._h, _dp(S), S.size, _dp(w)))
        return w

    def run_closed_loop(self, n_iters, trace=False, stream=None):
        tr = np.empty((n_iters, 2)) if trace else None
        self._ck(self.lib.mppi_run_closed_loop(self._h, int(n_iters), _dp(tr) if trace else None,
                                               C.byref(self.stats), _stream_ptr(stream)))
        return tr, self.stats

    def enable_timing(self, on=True):
        self._ck(self.lib.mppi_enable_timing(self._h, int(bool(on))))

    def set_rollout_repeats(self, n):
        self._ck(self.lib.mppi_set_rollout_repeats(self._h, int(n)))

    def counters(self):
        out = (C.c_int64 * 3)()
        self._ck(self.lib.mppi_get_counters(self._h, out))
        layout = C.c_int32(-1)
        if hasattr(self.lib, "mppi_get_rollout_layout"):  # (absent from an older diagnostic build under MPPI_LIB)
            self._ck(self.lib.mppi_get_rollout_layout(self._h, C.byref(layout)))
        return {"iterations": out[0], "rollout_launches": out[1], "finalize_launches": out[2], "rollout_layout": layout.value}

    def rollout_kernel(self):
        """Name of the kernel instantiation the last rollout-class launch took, as rocprofv3 prints it."""
        if not hasattr(self.lib, "mppi_get_rollout_kernel"):  # (an older diagnostic build under MPPI_LIB)
            return ""
        buf = C.create_string_buffer(192)
        self._ck(self.lib.mppi_get_rollout_kernel(self._h, buf, 192))
        return buf.value.decode()

    def host_timing(self):
        """Seconds this handle's closed-loop calls spent enqueueing launches / inside the calls (cumulative)."""
        out = (C.c_double * 2)()
        self._ck(self.lib.mppi_get_host_timing(self._h, out))
        return {"enqueue_s": out[0], "loop_s": out[1]}

    def time_rollout_launch(self, n_slots=500, extra=2, stream=None):
        """GPU-side launch-to-launch duration of the rollout kernel (microseconds) and the in-graph iteration period, from
        graph replays (`mppi_time_rollout_launch`): independent of how fast the host enqueues."""
        out = (C.c_double * 2)()
        self._ck(self.lib.mppi_time_rollout_launch(self._h, int(n_slots), int(extra), _stream_ptr(stream), out))
        return {"rollout_us": out[0], "graph_iteration_us": out[1]}

    def last_kernel_ms(self):
        out = (C.c_float * 4)()
        self._ck(self.lib.mppi_last_kernel_ms(self._h, out))
        return {"rollout": out[0], "reduce": out[1], "finalize": out[2], "event_pair_overhead": out[3]}
