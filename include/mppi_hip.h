/*
 * mppi_hip.h -- C ABI of libmppi_hip.so, the MI355X (gfx950) MPPI rollout engine.
 *
 * The reference (SokhengDin/DNN-MPPI-MPC) has no FFI: its boundary for this path is the
 * Python class surface of controllers/mppi_differential_drive*.py and
 * controllers/mppi_race_car*.py.  This header is the C ABI placed underneath that surface
 * (SURVEY.md section 8b); every entry point cites the reference code it stands in for.
 * `file:line` citations are relative to the reference repository root.
 *
 * Conventions
 *   - every function returns 0 on success or a negative mppi_status; text via
 *     mppi_last_error().  No exception crosses the ABI.
 *   - "host" pointers are ordinary CPU memory; "device" pointers are HIP device memory on
 *     the handle's GPU (e.g. torch.Tensor.data_ptr() of a contiguous CUDA tensor --
 *     accepted zero-copy).  The caller owns every buffer it passes.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - a handle is not thread-safe: one handle per host thread / GPU.
 */
#ifndef MPPI_HIP_H
#define MPPI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPPI_ABI_VERSION 4

typedef enum {
    MPPI_OK = 0,
    MPPI_ERR_BAD_ARG = -1,    /* null pointer, bad enum, inconsistent sizes */
    MPPI_ERR_SHAPE = -2,      /* T below the filter window, K < 1, path not set ... */
    MPPI_ERR_NO_DEVICE = -3,  /* no HIP device / wrong architecture */
    MPPI_ERR_HIP = -4,        /* a HIP runtime call failed */
    MPPI_ERR_PATH_END = -5,   /* race car: nearest waypoint is the last one (mppi_race_car.py:63-65) */
    MPPI_ERR_UNSUPPORTED = -6,/* e.g. sequential waypoint mode with K sharded over ranks */
    MPPI_ERR_STATE = -7,      /* call sequence error (step_end without step_begin ...) */
    MPPI_ERR_COMM = -8        /* peer-to-peer exchange: a rank did not arrive within the timeout */
} mppi_status;

/* dynamics: controllers/mppi_differential_drive.py:182-198 / controllers/mppi_race_car.py:183-197 */
typedef enum {
    MPPI_MODEL_DIFFDRIVE = 0,
    MPPI_MODEL_RACECAR = 1,
    /* unicycle + learned residual, x' = x + dt (f(x,v) + MLP([x,v])) (test/bullet_differential_drive_dnn.py:79-92);
     * needs mppi_set_mlp; fp32 only (f32 MFMA) */
    MPPI_MODEL_DIFFDRIVE_MLP = 2
} mppi_model;
/* arithmetic type of the rollout/cost kernels (the noise tensor is always f32) */
typedef enum { MPPI_PREC_F32 = 0, MPPI_PREC_F64 = 1 } mppi_precision;
/* waypoint search state, SURVEY.md App. A.3:
 *  SEQUENTIAL = one index threaded through all K*(T+1) cost calls in k-major order
 *               (mppi_differential_drive.py:228,:244, update_prev_idx=True)
 *  FROZEN     = index fixed at the nearest waypoint of x0 during rollouts
 *               (mppi_race_car.py:143,:152, default update_prev_idx=False)
 *  PER_ROLLOUT = the index threads through a sample's OWN T stage calls and its terminal call exactly as the reference
 *               threads it (:228, :244: every call searches from where the previous one ended) but starts again from the
 *               x0 call's index at every sample -- the part of the reference's bookkeeping that survives when samples
 *               must be independent (K sharded over GPUs, several agents per handle, all host cores in the CPU baseline);
 *               prev_way_point_idx after the iteration = the x0 call's index, as in FROZEN.  Diff-drive models only (the
 *               race-car files have no such bookkeeping: MPPI_ERR_UNSUPPORTED) */
typedef enum { MPPI_WAYPOINT_SEQUENTIAL = 0, MPPI_WAYPOINT_FROZEN = 1, MPPI_WAYPOINT_PER_ROLLOUT = 2 } mppi_waypoint_mode;
/* softmin rate beta in w = exp(-beta (S - rho)):
 *  INV_EXPLORATION 1/param_exploration (mppi_differential_drive.py:175)
 *  INV_LAMBDA      1/param_lambda      (mppi_race_car.py:205)
 *  LAMBDA          param_lambda        (mppi_differential_drive_torch.py:187) */
typedef enum { MPPI_BETA_INV_EXPLORATION = 0, MPPI_BETA_INV_LAMBDA = 1, MPPI_BETA_LAMBDA = 2 } mppi_beta_mode;
/* smoothing of the weighted noise:
 *  DIFFDRIVE  np.convolve 'same' + edge factors incl. the repeated last-row factor
 *             (mppi_differential_drive.py:257-271)
 *  RACECAR    pad with first/last window/2 rows, convolve, unpad (mppi_race_car.py:211-222)
 *  TORCH      the torch files' variant of RACECAR: same padding, `conv1d(padding=window/2)` and the FIRST T
 *             outputs, i.e. output n averages padded rows n-5 .. n+4 with zeros before the start
 *             (mppi_differential_drive_torch.py:252-263, mppi_race_car_torch.py:211-222) */
typedef enum {
    MPPI_FILTER_DIFFDRIVE = 0, MPPI_FILTER_RACECAR = 1, MPPI_FILTER_NONE = 2, MPPI_FILTER_TORCH = 3
} mppi_filter_mode;
/* collision term (adds collision_penalty per colliding stage/terminal state):
 *  CIRCLE   robot disc of radius 0.5*margin vs circles (mppi_differential_drive_obs.py:301-313)
 *  OUTLINE  9 outline points of the (l*m) x (w*m) box vs circles (mppi_race_car_obstacle.py:241-274) */
typedef enum { MPPI_OBSTACLE_NONE = 0, MPPI_OBSTACLE_CIRCLE = 1, MPPI_OBSTACLE_OUTLINE = 2 } mppi_obstacle_model;

/*
 * Mirrors the constructor keywords of MPPIAlgorithms (mppi_differential_drive.py:44-60,
 * _obs.py:44-62) and MPPIRacecarController (mppi_race_car.py:10-27, _obstacle.py:11-30),
 * plus one switch per behavioural difference between the reference's variants
 * (SURVEY.md App. B).  Zero-initialise, set struct_size = sizeof(mppi_config).
 */
typedef struct {
    int32_t struct_size;
    int32_t device;              /* HIP device ordinal */
    int32_t model;               /* mppi_model */
    int32_t precision;           /* mppi_precision */
    int32_t K;                   /* samples evaluated by THIS handle (num_samples_K / number_of_samples_K) */
    int32_t T;                   /* horizon (num_horizons_T / horizon_step_T) */
    int32_t K_global;            /* total samples over all ranks (0 => K) */
    int32_t k_offset;            /* global index of this handle's first sample */
    double delta_t;
    double u_max[2];             /* diff: max_speed, max_omega; race: max_steer_abs, max_accel_abs */
    double wheel_base;           /* race car only */
    double param_exploration, param_lambda, param_alpha;
    double sigma[4];             /* row-major 2x2 */
    double stage_cost_weight[4]; /* diff: x,y,yaw; race: x,y,yaw,v */
    double terminal_cost_weight[4];
    int32_t beta_mode;           /* mppi_beta_mode */
    int32_t accumulate_stage_cost; /* 0: S[k] = ... (mppi_differential_drive.py:124); 1: S[k] += ... (mppi_race_car.py:84) */
    int32_t waypoint_mode;       /* mppi_waypoint_mode */
    int32_t search_window;       /* 20 (:204), 200 (mppi_race_car.py:158), 10 (_cuda.py:201) */
    int32_t wrap_yaw_stage;      /* (yaw + 2pi) % 2pi before the stage cost (mppi_race_car.py:141) */
    int32_t wrap_yaw_terminal;   /* same for the terminal cost (mppi_race_car.py:150, _cuda.py:239) */
    int32_t clamp_rollout;       /* `_g` inside the rollout (off only in mppi_differential_drive_torch.py:128) */
    int32_t clamp_u_after_update;/* in-place clamp of u by the visualisation loop (mppi_differential_drive.py:145-149) */
    int32_t filter_mode;         /* mppi_filter_mode */
    int32_t filter_window;       /* 10 */
    int32_t obstacle_model;      /* mppi_obstacle_model */
    int32_t raise_at_path_end;   /* 1: mppi_step returns MPPI_ERR_PATH_END (mppi_race_car.py:63-65) */
    double safety_margin;        /* diff: safety_margin_rate; race: collision_safety_margin_rate */
    double vehicle_w, vehicle_l; /* race outline, 3.0 / 4.0 (mppi_race_car_obstacle.py:53-54) */
    double collision_penalty;    /* 1.0e10 */
    uint64_t seed;               /* Philox key for the on-device sampler */
    /* Several independent MPPI problems ("agents") in one handle and one launch per stage:
     * same parameters, reference path and obstacles, separate state, nominal controls, waypoint index and noise.
     * 0/1 = one agent (every entry point).  > 1: needs MPPI_WAYPOINT_FROZEN, T <= 128, K <= 8192 and no sharding;
     * mppi_set_state / mppi_get_state / mppi_set_u_prev / mppi_get_u_prev / mppi_get_costs then take [n_agents][...]
     * arrays and mppi_run_closed_loop advances all agents (stats: agent 0; an agent that reaches the end of its path
     * stops the call with MPPI_ERR_PATH_END); mppi_set_waypoint_idx / mppi_set_iteration apply to every agent; the
     * host-in-the-loop and split steps, the visualisation rollouts and the exchange are single-agent. */
    int32_t n_agents;
    /* fourth word of the sampler's Philox counter (agent a of a batched handle draws with noise_stream + a, so a
     * single-agent handle with noise_stream = a reproduces its noise) */
    int32_t noise_stream;
} mppi_config;

/* per-iteration diagnostics (the reference only prints) */
typedef struct {
    double rho;           /* min_k S */
    double eta;           /* sum_k exp(-beta (S_k - rho)) */
    double ess;           /* effective sample size (sum w)^2 / sum w^2 */
    int32_t idx_start;    /* waypoint index after the x0 call (mppi_differential_drive.py:96) */
    int32_t idx_after;    /* prev_way_point_idx after the iteration */
    int32_t path_end;     /* idx_start reached the last waypoint (:97-99) */
    int32_t rounds;       /* speculation rounds the sequential waypoint mode needed (>= 1) */
    int64_t iteration;    /* iterations completed by this handle */
    int32_t n_collided;   /* samples of the last iteration whose cost carries a collision penalty (`_is_collided`,
                             _obs.py:301-313 / mppi_race_car_obstacle.py:255-274); 0 without obstacles; -1 when the records of
                             other ranks were merged (K sharded: they carry no count) */
    int32_t reserved;
    double iter_us;       /* host wall time per iteration of the call that filled this struct (mppi_step: the call;
                             mppi_run_closed_loop: the call divided by its iterations), microseconds */
    double kernel_us;     /* rollout + merge + finalize kernel time per iteration by HIP events on the launch stream: the
                             average over the window since mppi_enable_timing(h, 1), as of this call; 0 while timing is off */
} mppi_stats;

typedef struct mppi_handle mppi_handle;

/* library level */
int mppi_abi_version(void);
const char *mppi_last_error(const mppi_handle *h); /* h may be NULL: last create() failure */
int mppi_device_count(void);

/* MPPIAlgorithms.__init__ / MPPIRacecarController.__init__ (mppi_differential_drive.py:44-85,
 * mppi_race_car.py:10-52): validates, allocates the workspaces, u_prev = 0, waypoint idx = 0 */
int mppi_create(const mppi_config *cfg, mppi_handle **out);
int mppi_destroy(mppi_handle *h);

/* `self.ref_path` (host, row-major [n, ncols], ncols = 3 (x,y,yaw) or 4 (x,y,yaw,v));
 * re-settable like the attribute (mppi_race_car.py:267) */
int mppi_set_ref_path(mppi_handle *h, const double *path, int32_t n, int32_t ncols);
/* `self.obstacle_circles` (host, [m,3] = x,y,r) */
int mppi_set_obstacles(mppi_handle *h, const double *xyr, int32_t m);
/*
 * Weights of the residual model `MultiLayerPerceptron` (train/train_diff_mlp.py:13-36), host float arrays in the
 * checkpoint's own layout (saved_models/mlp_diff_300x100_3l.pth): input_layer.weight [512,5], .bias [512];
 * hidden_layer.{0..n_hidden-1}.weight [512,512], .bias [512]; out_layer.weight [3,512], .bias [3].
 * hidden must be 512; n_hidden 3 (the architecture the reference trains, mlp_diff_300x100_3l*.pth) or 2 (its older
 * checkpoints mlp_diff.pth, mlp_diff_300x100.pth, mlp_diff_300x100_v2.pth: hidden_layer.{0,1}).
 * Numeric range: the default kernel carries weights, inputs and activations as pairs of f16 numbers.  Inputs [x, y, yaw, v, w]
 * and first-layer pre-activations of ANY finite magnitude are handled (per-sample power-of-two scales inside the kernel); a
 * WEIGHT beyond +-65504 cannot be, so such a model is served by the f32-input MFMA kernel instead (about 3x slower, same
 * results to the tolerance of the tests); mppi_last_error then says so and mppi_get_rollout_kernel returns "k_rollout_mlp(".
 */
int mppi_set_mlp(mppi_handle *h, int32_t hidden, int32_t n_hidden, const float *w_in, const float *b_in,
                 const float *const *w_hidden, const float *const *b_hidden, const float *w_out, const float *b_out);
/*
 * The same with the `StandardScaler` statistics the model was trained with (train/train_diff_mlp.py:72-86; values in
 * SURVEY.md App. C): the network then sees (z - in_mean) / in_scale, z = [x, y, yaw, v, w], and its output is taken as
 * y * out_scale + out_mean.  The two affine maps are folded into the first and the last Linear in f64 on the host
 * (W_in / in_scale, b_in - (W_in / in_scale) in_mean; out_scale W_out, out_scale b_out + out_mean), so the kernels are
 * the same.  in_mean / in_scale: 5 doubles, out_mean / out_scale: 3 doubles; a NULL pair means identity.
 */
int mppi_set_mlp_scaled(mppi_handle *h, int32_t hidden, int32_t n_hidden, const float *w_in, const float *b_in,
                        const float *const *w_hidden, const float *const *b_hidden, const float *w_out, const float *b_out,
                        const double *in_mean, const double *in_scale, const double *out_mean, const double *out_scale);
/* mutable controller state: `u_prev[T,2]` and `prev_way_point_idx` / `prev_waypoints_idx`
 * (mppi_differential_drive.py:82,:85); host pointers */
int mppi_set_u_prev(mppi_handle *h, const double *u);
int mppi_get_u_prev(mppi_handle *h, double *u);
int mppi_set_waypoint_idx(mppi_handle *h, int32_t idx);
int mppi_get_waypoint_idx(mppi_handle *h, int32_t *idx);

/*
 * One `_calc_input_control` / `_calc_control_input` (mppi_differential_drive.py:87-165,
 * mppi_race_car.py:55-121): sample -> rollout -> cost -> softmin weight -> reduce ->
 * filter -> update -> shift.
 *   x0      host, nx doubles (3 diff / 4 race)                                  [observed_x]
 *   eps     device, float[K,T,2] C-order, or NULL => Philox(seed, iteration)    [`_calc_epsilon`]
 *   u_out   host, double[T,2]: the returned (shifted) sequence                  [`u`, :165]
 *   u0_out  host, double[2]:  the returned first control (pre-shift u[1])       [`u[0]`, :165]
 *   stats   host, nullable
 * Synchronises `stream` before returning (the outputs are host memory).
 */
int mppi_step(mppi_handle *h, const double *x0, const float *eps, double *u_out, double *u0_out,
              mppi_stats *stats, void *stream);

/* The same with the observed state in DEVICE memory (nx doubles, e.g. a float64 CUDA tensor's data_ptr()): a state
 * estimator that already lives on the GPU hands it over without a host round trip. */
int mppi_step_device_x0(mppi_handle *h, const double *x0_device, const float *eps, double *u_out, double *u0_out,
                        mppi_stats *stats, void *stream);

/*
 * Split form for K sharded over ranks.  `begin` runs sample ->
 * rollout -> cost -> local softmin partial and writes this rank's record
 * {rho_g, eta_g, eta2_g, W_g[T,2]} (3 + 2T doubles, mppi_partial_len) to `partial` (device).
 * The caller all-gathers the records over its communicator (RCCL); `end` merges `nranks`
 * records (device, double[nranks, 3+2T]) and finishes the iteration identically on every
 * rank.  x0 == NULL in `begin` takes the state already on the device (closed loop);
 * `end_async` then advances it with the plant and returns without synchronising, and
 * `mppi_sync_result` fetches the outputs of the last finished iteration.
 */
int mppi_partial_len(const mppi_handle *h, int32_t *n_doubles);
int mppi_step_begin(mppi_handle *h, const double *x0, const float *eps, double *partial, void *stream);
int mppi_step_end(mppi_handle *h, const double *partials, int32_t nranks, double *u_out, double *u0_out,
                  mppi_stats *stats, void *stream);
int mppi_step_end_async(mppi_handle *h, const double *partials, int32_t nranks, void *stream);
int mppi_sync_result(mppi_handle *h, double *u_out, double *u0_out, mppi_stats *stats, void *stream);

/*
 * Peer-to-peer form of the same exchange ("direct P2P stores into peers' buffers +
 * flag polling over xGMI" when the collective's latency dominates an iteration).  Every rank owns one
 * exchange buffer in fine-grained device memory; `export` creates it and returns its IPC handle
 * (mppi_comm_handle_bytes() bytes), the caller passes the handles of all ranks around once (any host
 * channel, e.g. torch.distributed.all_gather_object) and `connect` maps them (entries of `local_ptrs`
 * that are non-NULL are used as they are: peers living in the same process, see `mppi_comm_buffer`; such a
 * pointer must be device memory, and when it lives on another GPU `connect` enables peer access to that GPU --
 * MPPI_ERR_UNSUPPORTED when it is not device memory or the two GPUs cannot reach each other).
 * From then on mppi_step and mppi_run_closed_loop exchange the per-rank record inside the finalize
 * kernel: store into every peer's buffer, raise a flag, wait for all flags -- no host call and no
 * collective launch per iteration.  Every rank must make the same sequence of step / closed-loop /
 * probe calls.  A rank that does not arrive within MPPI_EXCHANGE_TIMEOUT_MS (default 3000) makes the
 * call fail with MPPI_ERR_COMM on every rank; `probe` runs one flag round to check the wiring.
 * Needs MPPI_WAYPOINT_FROZEN (as the split form does).
 */
int mppi_comm_handle_bytes(void);
int mppi_comm_export(mppi_handle *h, int32_t nranks, void *handle_out);
int mppi_comm_buffer(mppi_handle *h, void **device_ptr);
int mppi_comm_connect(mppi_handle *h, int32_t rank, int32_t nranks, const void *handles /* [nranks][handle_bytes] */,
                      void *const *local_ptrs /* nullable [nranks] */);
int mppi_comm_probe(mppi_handle *h, void *stream);
int mppi_comm_close(mppi_handle *h);

/*
 * The same exchange carried by RCCL INSIDE the library (`mppi_comm_init(h, ncclUniqueId, rank,
 * nranks)`): rank 0 obtains an id with `mppi_comm_unique_id` (= ncclGetUniqueId, mppi_comm_unique_id_bytes() = 128 bytes),
 * passes it to every rank over any host channel, and every rank calls `mppi_comm_init` (= ncclCommInitRank on the handle's
 * device; collective, blocks until all ranks have called).  From then on mppi_step and mppi_run_closed_loop on that handle
 * enqueue ONE ncclAllGather of the per-rank record (3 + 2T doubles) per iteration between the rollout and the finalize
 * launches, on the caller's stream: a C caller needs no torch.distributed.  A peer-to-peer connection (mppi_comm_connect)
 * takes precedence over it when both exist.  librccl.so.1 is loaded on first use (dlopen), not linked.  Needs
 * MPPI_WAYPOINT_FROZEN; every rank must make the same sequence of calls; `mppi_comm_close` releases the communicator.
 */
int mppi_comm_unique_id_bytes(void);
int mppi_comm_unique_id(void *id_out);
int mppi_comm_init(mppi_handle *h, const void *unique_id, int32_t rank, int32_t nranks);

/* S[K] of the last iteration (`S`, :103) and its weights (`_compute_weight`, :167-180); host doubles */
int mppi_get_costs(mppi_handle *h, double *S);
int mppi_get_weights(mppi_handle *h, double *w);
/* the noise the sampler produces for `iteration` (`_calc_epsilon`, :273-283): device float[K,T,2]
 * ([n_agents,K,T,2] for a batched handle: agent a's tensor follows agent a-1's) */
int mppi_sample_epsilon(mppi_handle *h, int64_t iteration, float *eps_out, void *stream);
/*
 * `_calc_epsilon` materialised for the calls that take no tensor of their own (mppi_run_closed_loop, and mppi_step /
 * mppi_step_begin with eps == NULL): a ring of `n_slots` noise tensors in DEVICE memory, float[n_slots][n_agents][K][T][2];
 * iteration i (the handle's iteration counter, see mppi_set_iteration) reads slot i mod n_slots, picked inside the kernels,
 * so a closed loop on the device replays a recorded noise sequence -- and the rollout reads its noise as coalesced rows from
 * HBM (the reference's second pass over eps, :132-135, stays in registers).  n_slots must be a power of two; the caller owns
 * the ring and keeps it alive; eps_ring == NULL or n_slots == 0 returns to the in-kernel Philox sampler.  A ring filled by
 * mppi_sample_epsilon(h, i, ring + i * n_agents * K * T * 2) for i = 0 .. n_slots-1 reproduces the sampler's run over those
 * iterations bit for bit.
 */
int mppi_set_noise_ring(mppi_handle *h, const float *eps_ring, int32_t n_slots);
/* iteration counter that keys the sampler (checkpoint / resume) */
int mppi_set_iteration(mppi_handle *h, int64_t iteration);

/*
 * The visualisation rollouts of the last iteration (mppi_differential_drive.py:144-159):
 * optimal_traj[T,nx] from the updated u and sampled_traj[K,T,nx] from the clamped v, both
 * with the reference's `[t-1]` control indexing.  Device float buffers, either nullable.
 * MPPI_MODEL_DIFFDRIVE_MLP: the same loop with the learned transition (the rollout kernel's network code, states stored
 * instead of costs).
 */
int mppi_rollout_viz(mppi_handle *h, float *optimal_traj, float *sampled_traj, void *stream);

/*
 * Closed loop on the device: the plant of the reference's driver
 * (DifferentialDrive.update_state mppi_differential_drive.py:33-40 / Vehicle.update
 * models/vehicle.py:85-114) advances the state with the returned control after every
 * iteration, with no host round trip.  `mppi_set_state` seeds it; `mppi_run_closed_loop`
 * runs `n_iters` complete iterations (eps from the sampler) and synchronises once.
 */
int mppi_set_state(mppi_handle *h, const double *x);
int mppi_get_state(mppi_handle *h, double *x);
int mppi_run_closed_loop(mppi_handle *h, int32_t n_iters, double *u0_trace /* host, nullable [n_iters,2] */,
                         mppi_stats *stats, void *stream);

/*
 * The stage methods of the controller classes as batched entry points, evaluated by the device
 * code the rollout kernels use, in the handle's precision.  Host arrays of n items in and out.
 *   mppi_eval_state_transition  `_state_transition` (mppi_differential_drive.py:182-198) / `_F` (mppi_race_car.py:183-197):
 *                               x[n,nx], v[n,2] -> x_next[n,nx]   (no clamping: that is `_g`, mppi_eval_clamp)
 *   mppi_eval_clamp             `_g` (:285-289 / mppi_race_car.py:176-181): v[n,2] -> clamped [n,2]
 *   mppi_eval_is_collided       `_is_collided` (_obs.py:301-313 / mppi_race_car_obstacle.py:255-274): x[n,nx] -> {0,1}[n]
 *   mppi_eval_nearest_waypoint  `_get_nearest_waypoint` (:201-220) / `get_nearest_waypoint` (mppi_race_car.py:157-174)
 *                               for the positions x[n,nx] (columns 0,1), searched from *prev_idx.  update_prev_idx = 1:
 *                               the index threads through the n calls in order, as n successive calls of the reference
 *                               method would, and *prev_idx is left at the last one; 0: every call starts at *prev_idx
 *   mppi_eval_cost              `_compute_cost` / `_c` (terminal = 0), `_terminal_cost` / `_phi` (terminal = 1): the
 *                               weighted tracking error against the nearest waypoint (searched as above) + the
 *                               collision penalty; idx_out (nullable) receives the waypoint index of every call
 *   mppi_eval_moving_average    `_moving_average_filter` in the handle's filter mode: xx[T,2] -> [T,2]
 *   mppi_eval_weights           `_compute_weight` of a given cost vector S[n] with the handle's softmin rate
 */
int mppi_eval_state_transition(mppi_handle *h, const double *x, const double *v, int32_t n, double *x_next);
int mppi_eval_clamp(mppi_handle *h, const double *v, int32_t n, double *out);
int mppi_eval_is_collided(mppi_handle *h, const double *x, int32_t n, double *out);
int mppi_eval_nearest_waypoint(mppi_handle *h, const double *x, int32_t n, int32_t *prev_idx, int32_t update_prev_idx,
                               int32_t *idx_out);
int mppi_eval_cost(mppi_handle *h, int32_t terminal, const double *x, int32_t n, int32_t *prev_idx, int32_t update_prev_idx,
                   double *cost, int32_t *idx_out);
int mppi_eval_moving_average(mppi_handle *h, const double *xx, double *out);
int mppi_eval_weights(mppi_handle *h, const double *S, int32_t n, double *w);

/* Kernel durations by HIP events on the launch stream.  While enabled, every launch group of every
 * iteration is bracketed by an event pair, plus one empty pair that calibrates the cost of the
 * bracketing itself.  mppi_last_kernel_ms: averages (ms) over the window since timing was enabled,
 * calibration subtracted: out[0] rollout+cost(+fused softmin partial), out[1] separate reduce / merge
 * launches (0 when fused), out[2] finalise, out[3] the empty-pair calibration itself. */
int mppi_last_kernel_ms(mppi_handle *h, float *out4);
int mppi_enable_timing(mppi_handle *h, int32_t on);
/* Measurement aid: launch the rollout kernel `n` >= 1 times per iteration (it is idempotent -- same
 * state in, same S / partial records out).  The growth of the iteration period per extra launch is that
 * kernel's launch-to-launch duration on the stream, free of any event overhead (bench.py). */
int mppi_set_rollout_repeats(mppi_handle *h, int32_t n);
/* Launch bookkeeping since mppi_create (host side): out3 = {iterations completed, rollout-class kernel launches
 * (the repeats of mppi_set_rollout_repeats included), finalize launches}.  An iteration of the sequential waypoint
 * mode may take several of each (speculation rounds); bench.py divides a measured time by these. */
int mppi_get_counters(mppi_handle *h, int64_t *out3);

/* Host side of mppi_run_closed_loop since mppi_create (seconds, cumulative): out2 = {time spent ENQUEUEING the launches,
 * time inside the calls}.  A loop the host cannot feed fast enough shows enqueue ~ whole call; bench.py reports the ratio
 * (`host_enqueue_share`), because the iteration's 9 us leave a host that needs ~3 us per launch little slack. */
int mppi_get_host_timing(const mppi_handle *h, double *out2);

/* Launch-to-launch duration of the rollout kernel on the GPU with the host out of the loop (bench.py's `roofline.kernel_us`):
 * `n_slots` closed-loop iterations are captured into a HIP graph as they are and with the (idempotent) rollout kernel
 * launched 1 + `extra` times per iteration, each replayed four times between two events; us_out2 = {microseconds one more
 * launch of the kernel costs, microseconds per iteration of the plain graph}.  The iterations are real ones (the state
 * advances).  Needs a waypoint index that cannot move (frozen mode, or the end of the path) and no rank exchange. */
int mppi_time_rollout_launch(mppi_handle *h, int32_t n_slots, int32_t extra, void *stream, double *us_out2);

/* Which fused rollout kernel serves the handle (fixed at mppi_create from K x n_agents and T; diagnostic, for tests and
 * profiles): low two bits 0 = one sample per wave, lanes over the horizon; 1 = two samples per wave, two steps per lane
 * (T <= 64 and >= 8192 samples per launch); 2 = one sample per wave, two steps per lane (64 < T <= 128); 3 = two samples per
 * wave, three steps per lane (the race car as the reference runs it, f32, 64 < T <= 96);
 * +4 = every (half-)wave rolls out two samples in sequence.  -1: the handle does not use the fused kernels. */
int mppi_get_rollout_layout(const mppi_handle *h, int32_t *layout);
/* The kernel instantiation the handle's last rollout-class launch took, spelled as rocprofv3 prints it (e.g.
 * "k_rollout_fused<float, 0, 1, false, 2, false>", "k_rollout_dual<float, 1, 1, false, 2, true>"): bench.py looks the
 * launch's counter figures up under this name in profiles/.  A learned-dynamics handle answers as soon as mppi_set_mlp has
 * run: "k_rollout_mlp_h3<false, 8, 2, 3>" (operands as two f16 halves, the default) or "k_rollout_mlp(" (f32-input MFMA: chosen by
 * MPPI_MLP_F32=1, or by mppi_set_mlp itself when a weight exceeds the f16 range). */
int mppi_get_rollout_kernel(const mppi_handle *h, char *buf, int32_t n);

/*
 * `pytorch_mppi`-style MPPI with built-in dynamics / running-cost models: the callbacks the
 * reference's test/test_mppi.py, test/test_mppi_diff.py, test/test_mppi_diff_dyna.py and
 * train/bullet_mppi_differential_drive.py hand to `pytorch_mppi.MPPI`, as selectable device functions.  `mppi_cb_command`
 * is `MPPI.command(state)`: shift the nominal sequence, sample, roll out, weigh, update, return the first action.
 * The library itself is absent from the build container and un-pinned by the reference: the loop follows the
 * published algorithm and is PARITY UNPINNED (csrc/mppi_cb.hip, oracle/mppi_cb_oracle.py).  fp32 like the callers.
 */
typedef enum {
    MPPI_CB_DYN_UNICYCLE = 0,   /* test/test_mppi.py:12-26: [x, y, theta], [v, omega] */
    MPPI_CB_DYN_SKID_STEER = 1  /* test/test_mppi_diff_dyna.py:13-40: [x, y, theta, v, omega], 4 wheel forces */
} mppi_cb_dynamics;
typedef enum {
    MPPI_CB_OBS_INVERSE = 0,     /* 1 / (d + 1e-6) inside the safety distance (test/test_mppi.py:40-50) */
    MPPI_CB_OBS_EXPONENTIAL = 1  /* exp(-(d - safety)), obstacles moving with the step index (test/test_mppi_diff.py:25-52) */
} mppi_cb_obstacle_cost;
typedef struct {
    int32_t struct_size, device;
    int32_t dynamics;            /* mppi_cb_dynamics: fixes nx (3 / 5) and nu (2 / 4) */
    int32_t obstacle_kind;       /* mppi_cb_obstacle_cost */
    int32_t K, T;                /* num_samples, horizon (T * nu <= 256) */
    int32_t n_obs;               /* <= 16 */
    int32_t sample_null_action;  /* the last sample applies the zero action (test_mppi_diff_dyna.py:338) */
    double dt, lambda_;
    double noise_sigma[16];      /* row-major, leading nu x nu block of a 4 x 4 */
    double u_min[4], u_max[4], u_init[4];
    double goal[5], q_diag[5], r_diag[4];
    double obstacles[64];        /* [n_obs][4] = x, y, vx, vy (per step of the horizon) */
    double safety_distance, obstacle_weight;
    double skid_params[5];       /* m, I, r, L, damping (2.0, 0.05, 0.1, 0.4, 0.1 in the reference) */
    uint64_t seed;
} mppi_cb_config;
typedef struct mppi_cb_handle mppi_cb_handle;
const char *mppi_cb_last_error(const mppi_cb_handle *h);
int mppi_cb_create(const mppi_cb_config *cfg, mppi_cb_handle **out);
int mppi_cb_destroy(mppi_cb_handle *h);
/* the nominal control sequence `U` [T, nu] (host) */
int mppi_cb_set_nominal(mppi_cb_handle *h, const double *U);
int mppi_cb_get_nominal(mppi_cb_handle *h, double *U);
/* state: host [nx]; eps: device float [K, T, nu] or NULL (Philox); action_out: host [nu] */
int mppi_cb_command(mppi_cb_handle *h, const double *state, const float *eps, int32_t shift_nominal_trajectory,
                    double *action_out, void *stream);
/* `cost_total` [K] and the weights `omega` [K] of the last command (either nullable) */
int mppi_cb_get_costs(mppi_cb_handle *h, double *cost_total, double *omega);
/* the callbacks themselves, batched: what = 0 `dynamics(states, actions, t)` -> [n, nx]; 1 `running_cost` -> [n] */
int mppi_cb_eval(mppi_cb_handle *h, int32_t what, const double *states, const double *actions, int32_t t, int32_t n,
                 double *out);
/* the trajectory U drives from `state`: [T, nx] (the optimal trajectory of the callers' get_trajectories) */
int mppi_cb_nominal_trajectory(mppi_cb_handle *h, const double *state, double *traj);

#ifdef __cplusplus
}
#endif
#endif /* MPPI_HIP_H */
