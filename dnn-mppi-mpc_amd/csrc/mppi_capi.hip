// C ABI of libmppi_hip.so (include/mppi_hip.h): handle management, parameter packing and
// the launch sequence of one MPPI iteration.  No algorithmic arithmetic happens on the host.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <time.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/mppi_hip.h"
#include "mppi_kernels.h"

using namespace mppi;

static thread_local std::string g_create_error;

static double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct RcclUniqueId { char internal[128]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES), passed by value to ncclCommInitRank

struct mppi_handle {
    mppi_config cfg;
    bool f64 = false;
    int nx = 3, n_ref = 0, n_obs = 0, n_blocks = 0, traj_per_block = 0;
    bool fused = false;       // rollout + softmin partial in one launch (T <= 128)
    int n_part = 0;           // records the rollout/reduce stage leaves in d_partials
    int B = 1;                // agents (mppi_config.n_agents); per-agent buffers are B consecutive copies
    int slots = 0;            // records per agent in d_partials / d_heads (n_part + zero padding)
    void *d_partials2 = nullptr;    // second level for large K (records merged 64:1)
    void *d_heads = nullptr, *d_heads2 = nullptr;  // compact {rho, eta, eta2, 0} of d_partials / d_partials2
    float *d_mlp = nullptr;         // packed residual-model weights (config 5)
    unsigned short *d_mlp16 = nullptr;  // the same as f16 hi / lo planes (k_rollout_mlp_h3)
    // one-launch resolution of the sequential waypoint index (LB_CAND in mppi_kernels.h)
    bool hyp = false;
    unsigned *d_hyp_slots = nullptr;  // one look-back word per workgroup (LB_COPIES copies)
    unsigned lb_seq = 0;              // tag of the last rollout / finalize launch pair that used them
    std::vector<double> ref_host;   // [n_ref][4] as the kernels see it (rounded to the handle's precision)
    StepResult *res_mapped = nullptr;  // device-side address of the pinned host result (polled completion)
    long long seq = 0;
    bool poll = true, idx_valid = true, by_args_ok = true;
    int layout = 0;  // rollout_layout(K, T): which fused rollout kernel serves this handle
    const char *rollout_kernel = "";  // the instantiation the last rollout-class launch took (mppi_get_rollout_kernel)
    MlpParams mlp;
    bool mlp_set = false;
    void *d_ref = nullptr, *d_obs = nullptr, *d_u = nullptr, *d_uhist = nullptr, *d_S = nullptr;
    int *d_pout = nullptr;
    void *d_partials = nullptr;     // block records in the handle's precision
    double *d_w = nullptr, *d_trace = nullptr;
    long long trace_cap = 0;
    DevState *d_st = nullptr;
    StepResult *d_res = nullptr, *h_res = nullptr;
    size_t res_bytes = 0;
    const float *last_eps = nullptr;
    const float *noise_ring = nullptr;  // mppi_set_noise_ring: [noise_slots][n_agents][K][T][2] device floats (caller's)
    int noise_slots = 0;
    bool last_philox = true, begun = false, timing = false, dev_loop_primed = false, slot_timed = false;
    long long iter = 0;
    int idx = 0, rollout_repeats = 1;
    std::vector<hipEvent_t> ev;  // pairs around the rollout / reduce / finalize kernels
    size_t ev_used = 0;
    hipEvent_t ev_step[2] = {nullptr, nullptr};
    float last_ms[4] = {0, 0, 0, 0};
    double last_iter_us = 0.0;  // host wall time per iteration of the last step / closed-loop call (mppi_stats::iter_us)
    // peer-to-peer exchange (mppi_comm_*)
    char *xbuf = nullptr;            // this rank's exchange buffer (fine-grained device memory)
    size_t xbuf_bytes = 0;
    int x_rank = 0, x_nranks = 0, x_export_nranks = 0;
    std::vector<void *> x_opened;    // peers' buffers mapped through IPC handles
    char **d_xpeers = nullptr;       // device array [x_nranks]
    int *d_xerr = nullptr, *d_xok = nullptr;
    long long xseq = 0, x_timeout = 300000000LL;
    // RCCL carrier inside the library (mppi_comm_init): communicator, this rank's record and the gathered records
    void *rccl_comm = nullptr;
    int rccl_rank = 0, rccl_nranks = 0;
    double *d_rccl_part = nullptr, *d_rccl_gath = nullptr;
    long long n_rollout_launches = 0, n_finalize_launches = 0;  // mppi_get_counters
    double t_enqueue_s = 0.0, t_loop_s = 0.0;  // mppi_get_host_timing: closed-loop calls, enqueueing / whole call
    // closed-loop iterations replayed from a HIP graph once the waypoint index rests (closed_loop_impl)
    bool graph_on = false;
    std::vector<char> graph_key;      // the kernel arguments the cached graph was captured with
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};  // two instances, launched in turn (see closed_loop_impl)
    hipEvent_t graph_done[2] = {nullptr, nullptr};
    hipStream_t graph_stream = nullptr;
    hipEvent_t graph_ev_in = nullptr, graph_ev_out = nullptr;
    std::string err;
};

#define FAIL(h, code, ...)                                   \
    do {                                                     \
        char _b[512];                                        \
        snprintf(_b, sizeof(_b), __VA_ARGS__);               \
        if (h) (h)->err = _b; else g_create_error = _b;      \
        return (code);                                       \
    } while (0)

#define HIPCHECK(h, call)                                                                          \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess) FAIL(h, MPPI_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); \
    } while (0)

extern "C" int mppi_abi_version(void) { return MPPI_ABI_VERSION; }

extern "C" const char *mppi_last_error(const mppi_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int mppi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static size_t rsz(const mppi_handle *h) { return h->f64 ? sizeof(double) : sizeof(float); }

// host double[] -> device array in the kernel precision
static int upload_real(mppi_handle *h, void *dst, const double *src, size_t n) {
    if (h->f64) {
        HIPCHECK(h, hipMemcpy(dst, src, n * sizeof(double), hipMemcpyHostToDevice));
    } else {
        std::vector<float> tmp(n);
        for (size_t i = 0; i < n; ++i) tmp[i] = (float)src[i];
        HIPCHECK(h, hipMemcpy(dst, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    return MPPI_OK;
}

static int download_real(mppi_handle *h, double *dst, const void *src, size_t n) {
    if (h->f64) {
        HIPCHECK(h, hipMemcpy(dst, src, n * sizeof(double), hipMemcpyDeviceToHost));
    } else {
        std::vector<float> tmp(n);
        HIPCHECK(h, hipMemcpy(tmp.data(), src, n * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) dst[i] = (double)tmp[i];
    }
    return MPPI_OK;
}

extern "C" int mppi_create(const mppi_config *cfg, mppi_handle **out) {
    if (!cfg || !out) FAIL((mppi_handle *)nullptr, MPPI_ERR_BAD_ARG, "mppi_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(mppi_config))
        FAIL((mppi_handle *)nullptr, MPPI_ERR_BAD_ARG, "mppi_create: struct_size %d != %zu (ABI mismatch)",
             cfg->struct_size, sizeof(mppi_config));
    mppi_config c = *cfg;
    if (c.K_global == 0) c.K_global = c.K;
    if (c.K > (1 << 20) || c.T > 2048)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_SHAPE, "K=%d / T=%d beyond the supported 2^20 samples / 2048 steps", c.K, c.T);
    if (c.K < 1 || c.T < 1 || c.K_global < c.K || c.k_offset < 0 || c.k_offset + c.K > c.K_global)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_SHAPE, "mppi_create: bad K=%d T=%d K_global=%d k_offset=%d", c.K, c.T,
             c.K_global, c.k_offset);
    if (c.model != MPPI_MODEL_DIFFDRIVE && c.model != MPPI_MODEL_RACECAR && c.model != MPPI_MODEL_DIFFDRIVE_MLP)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_BAD_ARG, "mppi_create: unknown model %d", c.model);
    if (c.model == MPPI_MODEL_DIFFDRIVE_MLP && c.precision != MPPI_PREC_F32)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_UNSUPPORTED, "the learned-dynamics rollout runs on the f32 MFMA path only");
    if (c.precision != MPPI_PREC_F32 && c.precision != MPPI_PREC_F64)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_BAD_ARG, "mppi_create: unknown precision %d", c.precision);
    if (c.waypoint_mode != MPPI_WAYPOINT_SEQUENTIAL && c.waypoint_mode != MPPI_WAYPOINT_FROZEN && c.waypoint_mode != MPPI_WAYPOINT_PER_ROLLOUT)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_BAD_ARG, "mppi_create: unknown waypoint_mode %d", c.waypoint_mode);
    if (c.waypoint_mode == MPPI_WAYPOINT_PER_ROLLOUT && c.model == MPPI_MODEL_RACECAR)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_UNSUPPORTED,
             "MPPI_WAYPOINT_PER_ROLLOUT restates the diff-drive files' index bookkeeping (mppi_differential_drive.py:228,:244); "
             "the race-car files search from the x0 call's index (MPPI_WAYPOINT_FROZEN)");
    if (c.filter_window < 1) c.filter_window = 10;
    if (c.search_window < 1)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_BAD_ARG, "mppi_create: search_window must be >= 1");
    // the reference's filters fail on short horizons (np.convolve 'same' returns max(M, N) samples)
    if (c.filter_mode == MPPI_FILTER_DIFFDRIVE && c.T < c.filter_window)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_SHAPE, "horizon T=%d is shorter than the moving-average window %d", c.T,
             c.filter_window);
    if ((c.filter_mode == MPPI_FILTER_RACECAR || c.filter_mode == MPPI_FILTER_TORCH) && c.T < c.filter_window / 2)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_SHAPE, "horizon T=%d is shorter than half the filter window %d", c.T,
             c.filter_window);
    if (c.waypoint_mode == MPPI_WAYPOINT_SEQUENTIAL && c.K_global != c.K)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_UNSUPPORTED,
             "the sequential waypoint index threads through all samples in order and cannot be sharded; "
             "use MPPI_WAYPOINT_PER_ROLLOUT or MPPI_WAYPOINT_FROZEN with K_global > K");
    if (c.n_agents < 1) c.n_agents = 1;
    if (c.n_agents > 1) {
        if (c.n_agents > 4096) FAIL((mppi_handle *)nullptr, MPPI_ERR_SHAPE, "mppi_create: n_agents %d > 4096", c.n_agents);
        if (c.waypoint_mode == MPPI_WAYPOINT_SEQUENTIAL || c.K_global != c.K || c.model == MPPI_MODEL_DIFFDRIVE_MLP ||
            !fused_supported(c.T) || fused_blocks(c.K, c.T, rollout_layout(c.K, c.T, c.n_agents, c.model == MPPI_MODEL_RACECAR ? MODEL_RACE : MODEL_DIFF, c.precision == MPPI_PREC_F64, c.waypoint_mode == MPPI_WAYPOINT_PER_ROLLOUT)) > 512)
            FAIL((mppi_handle *)nullptr, MPPI_ERR_UNSUPPORTED,
                 "several agents per handle need MPPI_WAYPOINT_FROZEN or _PER_ROLLOUT, an analytic model, T <= 128, at most 512 "
                 "rollout workgroups (K <= 8192) and no sharding");
    }
    const double det = c.sigma[0] * c.sigma[3] - c.sigma[1] * c.sigma[2];
    if (!(c.sigma[0] > 0) || !(det > 0))
        FAIL((mppi_handle *)nullptr, MPPI_ERR_BAD_ARG, "sigma must be a symmetric positive definite 2x2 matrix");
    if (!(c.param_exploration >= 0) || !(c.param_lambda > 0))
        FAIL((mppi_handle *)nullptr, MPPI_ERR_BAD_ARG, "param_exploration must be >= 0 and param_lambda > 0");
    if (c.beta_mode == MPPI_BETA_INV_EXPLORATION && !(c.param_exploration > 0))
        FAIL((mppi_handle *)nullptr, MPPI_ERR_BAD_ARG, "beta = 1/param_exploration needs param_exploration > 0");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_NO_DEVICE, "no HIP device is visible: libmppi_hip.so needs an MI355X");
    if (c.device < 0 || c.device >= ndev)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_NO_DEVICE, "device %d out of range (%d visible)", c.device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c.device) != hipSuccess)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        FAIL((mppi_handle *)nullptr, MPPI_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
             c.device, prop.gcnArchName);

    if (c.n_agents > 1 && getenv("MPPI_FORCE_UNFUSED"))
        FAIL((mppi_handle *)nullptr, MPPI_ERR_UNSUPPORTED, "several agents per handle need the fused rollout kernels");
    mppi_handle *h = new mppi_handle();
    h->cfg = c;
    h->f64 = c.precision == MPPI_PREC_F64;
    h->nx = c.model == MPPI_MODEL_RACECAR ? 4 : 3;
    int tpb = 0;
    if (const char *e = getenv("MPPI_TRAJ_PER_BLOCK")) tpb = atoi(e);
    if (tpb < 1) tpb = (c.K + 127) / 128;
    tpb = ((tpb + 3) / 4) * 4;
    if (tpb > 2048) tpb = 2048;
    h->traj_per_block = tpb;
    h->n_blocks = reduce_blocks(c.K, tpb);
    h->fused = fused_supported(c.T) && !getenv("MPPI_FORCE_UNFUSED");
    h->graph_on = getenv("MPPI_GRAPH") && atoi(getenv("MPPI_GRAPH")) != 0;  // (opt-in: see ensure_graph)
    // (k_rollout_tri: the race car as the reference runs it -- frozen index, `S[k] +=` -- one agent, f32)
    const bool tri_ok = c.model == MPPI_MODEL_RACECAR && !h->f64 && c.n_agents == 1 && c.waypoint_mode == MPPI_WAYPOINT_FROZEN &&
                        c.accumulate_stage_cost;
    h->layout = rollout_layout(c.K, c.T, c.n_agents, c.model == MPPI_MODEL_RACECAR ? MODEL_RACE : MODEL_DIFF, h->f64,
                               c.waypoint_mode == MPPI_WAYPOINT_PER_ROLLOUT, tri_ok);
    h->n_part = h->fused ? fused_blocks(c.K, c.T, h->layout) : h->n_blocks;
    if (c.model == MPPI_MODEL_DIFFDRIVE_MLP) h->n_part = mlp_blocks(c.K, 64);  // (mppi_set_mlp sets it again for the kernel that serves the model)
    h->res_bytes = sizeof(StepResult) + sizeof(double) * 2 * c.T;
    auto fail = [&](hipError_t e, const char *what) {
        g_create_error = std::string(what) + " failed: " + hipGetErrorString(e);
        mppi_destroy(h);
        return (int)MPPI_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(c.device)) != hipSuccess) return fail(e, "hipSetDevice");
    const size_t r = rsz(h), B = (size_t)c.n_agents;
    h->B = c.n_agents;
    if ((e = hipMalloc(&h->d_u, B * r * 2 * c.T)) != hipSuccess) return fail(e, "hipMalloc(u)");
    if ((e = hipMalloc(&h->d_uhist, B * r * 4 * c.T)) != hipSuccess) return fail(e, "hipMalloc(u history)");
    if ((e = hipMalloc(&h->d_S, B * r * c.K)) != hipSuccess) return fail(e, "hipMalloc(S)");
    if ((e = hipMalloc((void **)&h->d_pout, B * sizeof(int) * c.K)) != hipSuccess) return fail(e, "hipMalloc(pout)");
    const size_t rec_bytes = sizeof(double) * (size_t)record_len(c.T, 8);  // enough for either precision
    // zero-filled and padded by 256 records: the merge kernels read 256 slots unconditionally
    // (the streaming kernel that serves noise tensors in the two-samples-per-wave layout leaves up to one record per 32
    // samples whatever the handle's own count is: room for the larger of the two)
    const size_t n_rec_max = std::max<size_t>((size_t)h->n_part, (h->layout & LAYOUT_KIND) == LAYOUT_DUAL ? ((size_t)c.K + 31) / 32 : 0);
    const size_t slots = n_rec_max + 256, n1 = B * slots, n2 = n_rec_max / 64 + 2 + 256;
    h->slots = (int)slots;
    if ((e = hipMalloc(&h->d_partials, rec_bytes * n1)) != hipSuccess) return fail(e, "hipMalloc(partials)");
    if ((e = hipMalloc(&h->d_partials2, rec_bytes * n2)) != hipSuccess) return fail(e, "hipMalloc(partials2)");
    if ((e = hipMemset(h->d_partials, 0, rec_bytes * n1)) != hipSuccess) return fail(e, "hipMemset");
    if ((e = hipMemset(h->d_partials2, 0, rec_bytes * n2)) != hipSuccess) return fail(e, "hipMemset");
    if ((e = hipMalloc(&h->d_heads, 32 * n1)) != hipSuccess) return fail(e, "hipMalloc(heads)");
    if ((e = hipMalloc(&h->d_heads2, 32 * n2)) != hipSuccess) return fail(e, "hipMalloc(heads2)");
    if ((e = hipMemset(h->d_heads, 0, 32 * n1)) != hipSuccess) return fail(e, "hipMemset");
    if ((e = hipMemset(h->d_heads2, 0, 32 * n2)) != hipSuccess) return fail(e, "hipMemset");
    // The sequential index in one launch (look-back, LB_CAND in mppi_kernels.h): horizons of one 64-step pass in the
    // one-sample-per-wave layout or the two-samples-per-wave layout with one pass per workgroup, the reference's 20- / 10-
    // candidate windows, `S[k] =`, one agent, at most 512 workgroups (K <= 16384: configs 2 and 3).  Anything else keeps the
    // speculation rounds alone (they also serve as this path's fallback).  MPPI_NO_HYP=1 switches it off for A/B runs.
    const bool lb_layout = (h->layout & LAYOUT_KIND) == LAYOUT_FUSED || h->layout == LAYOUT_DUAL;  // (one pass per workgroup)
    h->hyp = h->fused && lb_layout && c.T <= 64 && c.model == MPPI_MODEL_DIFFDRIVE &&
             c.waypoint_mode == MPPI_WAYPOINT_SEQUENTIAL && !c.accumulate_stage_cost && (c.search_window == HYP_WINDOW || c.search_window == HYP_WINDOW_CUDA) &&
             c.n_agents == 1 && h->n_part <= HYP_MAX_BLOCKS && !getenv("MPPI_NO_HYP");
    if (h->hyp) {
        const size_t lb_bytes = sizeof(unsigned) * (size_t)LB_COPIES * LB_COPY_STRIDE;
        if ((e = hipMalloc((void **)&h->d_hyp_slots, lb_bytes)) != hipSuccess) return fail(e, "hipMalloc(look-back words)");
        if ((e = hipMemset(h->d_hyp_slots, 0, lb_bytes)) != hipSuccess) return fail(e, "hipMemset");
    }
    h->res_bytes = (h->res_bytes + 15) & ~(size_t)15;  // (the agents' results are stored back to back)
    if ((e = hipMalloc((void **)&h->d_st, B * sizeof(DevState))) != hipSuccess) return fail(e, "hipMalloc(state)");
    if ((e = hipMalloc((void **)&h->d_res, B * h->res_bytes)) != hipSuccess) return fail(e, "hipMalloc(result)");
    if ((e = hipMemset(h->d_res, 0, B * h->res_bytes)) != hipSuccess) return fail(e, "hipMemset");
    if ((e = hipHostMalloc((void **)&h->h_res, B * h->res_bytes, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess)
        return fail(e, "hipHostMalloc(result)");
    if ((e = hipHostGetDevicePointer((void **)&h->res_mapped, h->h_res, 0)) != hipSuccess)
        return fail(e, "hipHostGetDevicePointer(result)");
    h->poll = !getenv("MPPI_NO_POLL");
    h->by_args_ok = !getenv("MPPI_NO_ARGS");
    if ((e = hipMemset(h->d_u, 0, B * r * 2 * c.T)) != hipSuccess) return fail(e, "hipMemset");    // u_prev = 0 (:82)
    if ((e = hipMemset(h->d_uhist, 0, B * r * 4 * c.T)) != hipSuccess) return fail(e, "hipMemset");
    if ((e = hipMemset(h->d_S, 0, B * r * c.K)) != hipSuccess) return fail(e, "hipMemset");
    if ((e = hipMemset(h->d_pout, 0, B * sizeof(int) * c.K)) != hipSuccess) return fail(e, "hipMemset");
    DevState st0;
    memset(&st0, 0, sizeof(st0));  // prev_way_point_idx = 0 (:85)
    st0.first_k = NO_TRIGGER;
    for (size_t a = 0; a < B; ++a)
        if ((e = hipMemcpy(h->d_st + a, &st0, sizeof(st0), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
    memset(h->h_res, 0, B * h->res_bytes);
    if ((e = hipEventCreate(&h->ev_step[0])) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = hipEventCreate(&h->ev_step[1])) != hipSuccess) return fail(e, "hipEventCreate");
    *out = h;
    return MPPI_OK;
}

extern "C" int mppi_comm_close(mppi_handle *h);

extern "C" int mppi_destroy(mppi_handle *h) {
    if (!h) return MPPI_OK;
    hipSetDevice(h->cfg.device);
    if (h->xbuf || h->rccl_comm) mppi_comm_close(h);
    void *bufs[] = {h->d_ref, h->d_obs, h->d_u, h->d_uhist, h->d_S, h->d_pout, h->d_partials, h->d_partials2, h->d_mlp,
                    h->d_w,   h->d_trace, h->d_st, h->d_res, h->d_heads, h->d_heads2, h->d_hyp_slots, h->d_mlp16};
    for (void *b : bufs)
        if (b) hipFree(b);
    if (h->h_res) hipHostFree(h->h_res);
    for (int i = 0; i < 2; ++i) {
        if (h->graph_exec[i]) hipGraphExecDestroy(h->graph_exec[i]);
        if (h->graph_done[i]) hipEventDestroy(h->graph_done[i]);
    }
    if (h->graph_ev_in) hipEventDestroy(h->graph_ev_in);
    if (h->graph_ev_out) hipEventDestroy(h->graph_ev_out);
    if (h->graph_stream) hipStreamDestroy(h->graph_stream);
    for (hipEvent_t e : h->ev) hipEventDestroy(e);
    for (hipEvent_t e : h->ev_step)
        if (e) hipEventDestroy(e);
    delete h;
    return MPPI_OK;
}

extern "C" int mppi_set_ref_path(mppi_handle *h, const double *path, int32_t n, int32_t ncols) {
    if (!h || !path) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_set_ref_path: null argument");
    const int need = h->cfg.model == MPPI_MODEL_RACECAR ? 4 : 3;
    if (n < 1 || ncols < need || ncols > 4)
        FAIL(h, MPPI_ERR_SHAPE, "ref_path must be [n>=1, %d] (got [%d, %d])", need, n, ncols);
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    std::vector<double> packed((size_t)n * 4, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < ncols; ++j) packed[(size_t)i * 4 + j] = path[(size_t)i * ncols + j];
    HIPCHECK(h, hipDeviceSynchronize());
    if (h->d_ref) HIPCHECK(h, hipFree(h->d_ref));
    h->d_ref = nullptr;
    HIPCHECK(h, hipMalloc(&h->d_ref, rsz(h) * 4 * n));
    h->n_ref = n;
    h->dev_loop_primed = false;
    h->ref_host = packed;
    if (!h->f64)
        for (double &v : h->ref_host) v = (double)(float)v;
    return upload_real(h, h->d_ref, packed.data(), packed.size());
}

extern "C" int mppi_set_obstacles(mppi_handle *h, const double *xyr, int32_t m) {
    if (!h || (m > 0 && !xyr)) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_set_obstacles: null argument");
    if (m < 0) FAIL(h, MPPI_ERR_SHAPE, "mppi_set_obstacles: m < 0");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());
    if (h->d_obs) HIPCHECK(h, hipFree(h->d_obs));
    h->d_obs = nullptr;
    h->n_obs = m;
    if (m == 0) return MPPI_OK;
    std::vector<double> packed((size_t)m * 4, 0.0);
    for (int i = 0; i < m; ++i) {
        const double r = xyr[3 * i + 2];
        // mppi_differential_drive_obs.py:304-311: (0.5*margin + r)^2 ; mppi_race_car_obstacle.py:272: r^2
        const double thr = h->cfg.obstacle_model == MPPI_OBSTACLE_CIRCLE ? 0.5 * h->cfg.safety_margin + r : r;
        packed[4 * i] = xyr[3 * i];
        packed[4 * i + 1] = xyr[3 * i + 1];
        packed[4 * i + 2] = thr * thr;
    }
    HIPCHECK(h, hipMalloc(&h->d_obs, rsz(h) * 4 * m));
    return upload_real(h, h->d_obs, packed.data(), packed.size());
}

extern "C" int mppi_set_mlp(mppi_handle *h, int32_t hidden, int32_t n_hidden, const float *w_in, const float *b_in,
                            const float *const *w_hidden, const float *const *b_hidden, const float *w_out,
                            const float *b_out) {
    if (!h || !w_in || !b_in || !w_hidden || !b_hidden || !w_out || !b_out)
        FAIL(h, MPPI_ERR_BAD_ARG, "mppi_set_mlp: null argument");
    if (h->cfg.model != MPPI_MODEL_DIFFDRIVE_MLP) FAIL(h, MPPI_ERR_STATE, "mppi_set_mlp needs MPPI_MODEL_DIFFDRIVE_MLP");
    if (hidden != 512 || (n_hidden != 3 && n_hidden != 2))
        FAIL(h, MPPI_ERR_SHAPE, "mppi_set_mlp: Linear(5,512) -> n x [Linear(512,512), tanh] -> Linear(512,3) with n = 3 or 2 is built "
                                "(got hidden = %d, n = %d)", hidden, n_hidden);
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());
    const size_t n_in = 16 * 1 * 64 * 4, n_h = 16 * 64 * 64 * 4, total = n_in + 512 + 3 * (n_h + 512) + 3 * 512;
    std::vector<float> host(total);
    size_t o = 0;
    const size_t o_win = o; pack_linear(w_in, 5, host.data() + o); o += n_in;
    const size_t o_bin = o; memcpy(host.data() + o, b_in, 512 * sizeof(float)); o += 512;
    size_t o_wh[3], o_bh[3];
    for (int l = 0; l < 3; ++l) {  // (a two-layer model leaves the third slot unused)
        o_wh[l] = o; o_bh[l] = o + n_h;
        if (l < n_hidden) {
            if (!w_hidden[l] || !b_hidden[l]) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_set_mlp: null hidden layer %d", l);
            pack_linear(w_hidden[l], 512, host.data() + o);
            memcpy(host.data() + o + n_h, b_hidden[l], 512 * sizeof(float));
        }
        o += n_h + 512;
    }
    const size_t o_wo = o; memcpy(host.data() + o, w_out, 3 * 512 * sizeof(float)); o += 3 * 512;
    if (!h->d_mlp) HIPCHECK(h, hipMalloc((void **)&h->d_mlp, total * sizeof(float)));
    HIPCHECK(h, hipMemcpy(h->d_mlp, host.data(), total * sizeof(float), hipMemcpyHostToDevice));
    h->mlp.w_in = h->d_mlp + o_win;
    h->mlp.b_in = h->d_mlp + o_bin;
    for (int l = 0; l < 3; ++l) { h->mlp.w_h[l] = h->d_mlp + o_wh[l]; h->mlp.b_h[l] = h->d_mlp + o_bh[l]; }
    h->mlp.w_out = h->d_mlp + o_wo;
    for (int i = 0; i < 3; ++i) h->mlp.b_out[i] = b_out[i];
    {   // the same weights as f16 (hi, lo) planes for k_rollout_mlp_h3
        const size_t n_in16 = (size_t)2 * 16 * 1 * 64 * 8, n_h16 = (size_t)2 * 16 * 32 * 64 * 8, tot16 = n_in16 + 3 * n_h16;
        std::vector<unsigned short> h16(tot16);
        pack_linear_h3(w_in, 5, h16.data());
        for (int l = 0; l < n_hidden; ++l) pack_linear_h3(w_hidden[l], 512, h16.data() + n_in16 + (size_t)l * n_h16);
        if (!h->d_mlp16) HIPCHECK(h, hipMalloc((void **)&h->d_mlp16, tot16 * sizeof(unsigned short)));
        HIPCHECK(h, hipMemcpy(h->d_mlp16, h16.data(), tot16 * sizeof(unsigned short), hipMemcpyHostToDevice));
        h->mlp.h3_w_in = h->d_mlp16;
        for (int l = 0; l < 3; ++l) h->mlp.h3_w_h[l] = h->d_mlp16 + n_in16 + (size_t)l * n_h16;
        h->mlp.use_h3 = getenv("MPPI_MLP_F32") ? 0 : 1;
        // The split kernel carries every WEIGHT as two f16 numbers: one beyond the f16 range (65504; e.g. W_in / in_scale with a
        // StandardScaler scale near 1e-6) would become inf in the high plane.  Such a model takes the f32-input MFMA kernel,
        // which has no range limit, and the handle says so (mppi_last_error; mppi_get_mlp_kernel).  Inputs and first-layer
        // pre-activations of any magnitude are handled inside the split kernel (per-sample power-of-two scales).
        double wmax = 0.0;
        for (int i = 0; i < 512 * 5; ++i) wmax = fmax(wmax, fabs((double)w_in[i]));
        for (int l = 0; l < n_hidden; ++l)
            for (size_t i = 0; i < (size_t)512 * 512; ++i) wmax = fmax(wmax, fabs((double)w_hidden[l][i]));
        if (!(wmax <= 65504.0)) {  // (also NaN)
            h->mlp.use_h3 = 0;
            char b[256];
            snprintf(b, sizeof(b), "mppi_set_mlp: max |weight| = %.4g exceeds the f16 range: the f32-input MFMA kernel serves this model", wmax);
            h->err = b;
        }
    }
    {   // bounds the split kernel scales the first layer's output with: |W_in z + b_in|_inf <= in_gain |z|_inf + in_bias
        double gain = 0.0, bias = 0.0;
        for (int n = 0; n < 512; ++n) {
            double row = 0.0;
            for (int j = 0; j < 5; ++j) row += fabs((double)w_in[n * 5 + j]);
            gain = fmax(gain, row);
            bias = fmax(bias, fabs((double)b_in[n]));
        }
        h->mlp.in_gain = (float)(gain * (1.0 + 1e-6));
        h->mlp.in_bias = (float)(bias * (1.0 + 1e-6));
    }
    h->mlp.n_hidden = n_hidden;
    h->n_part = mlp_blocks(h->cfg.K, mlp_tile(h->mlp));  // records the rollout kernel that serves this model leaves
    h->mlp_set = true;
    return MPPI_OK;
}

extern "C" int mppi_set_mlp_scaled(mppi_handle *h, int32_t hidden, int32_t n_hidden, const float *w_in, const float *b_in,
                                   const float *const *w_hidden, const float *const *b_hidden, const float *w_out,
                                   const float *b_out, const double *in_mean, const double *in_scale, const double *out_mean,
                                   const double *out_scale) {
    if (!h || !w_in || !b_in || !w_out || !b_out) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_set_mlp_scaled: null argument");
    if ((in_mean == nullptr) != (in_scale == nullptr) || (out_mean == nullptr) != (out_scale == nullptr))
        FAIL(h, MPPI_ERR_BAD_ARG, "mppi_set_mlp_scaled: a mean without its scale (or the reverse)");
    if (hidden != 512) FAIL(h, MPPI_ERR_SHAPE, "mppi_set_mlp: the hidden width must be 512 (got %d)", hidden);
    std::vector<float> wi(w_in, w_in + (size_t)hidden * 5), bi(b_in, b_in + hidden), wo(w_out, w_out + (size_t)3 * hidden),
        bo(b_out, b_out + 3);
    if (in_mean) {
        for (int j = 0; j < 5; ++j)
            if (!(in_scale[j] != 0.0)) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_set_mlp_scaled: in_scale[%d] is zero", j);
        for (int n = 0; n < hidden; ++n) {
            double shift = 0.0;
            for (int j = 0; j < 5; ++j) {
                const double w = (double)w_in[(size_t)n * 5 + j] / in_scale[j];
                wi[(size_t)n * 5 + j] = (float)w;
                shift += w * in_mean[j];
            }
            bi[n] = (float)((double)b_in[n] - shift);
        }
    }
    if (out_mean) {
        for (int j = 0; j < 3; ++j) {
            for (int n = 0; n < hidden; ++n) wo[(size_t)j * hidden + n] = (float)((double)w_out[(size_t)j * hidden + n] * out_scale[j]);
            bo[j] = (float)((double)b_out[j] * out_scale[j] + out_mean[j]);
        }
    }
    return mppi_set_mlp(h, hidden, n_hidden, wi.data(), bi.data(), w_hidden, b_hidden, wo.data(), bo.data());
}

extern "C" int mppi_set_u_prev(mppi_handle *h, const double *u) {
    if (!h || !u) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_set_u_prev: null argument");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());
    return upload_real(h, h->d_u, u, (size_t)h->B * 2 * h->cfg.T);  // [n_agents][T][2]
}

extern "C" int mppi_get_u_prev(mppi_handle *h, double *u) {
    if (!h || !u) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_get_u_prev: null argument");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());
    return download_real(h, u, h->d_u, (size_t)h->B * 2 * h->cfg.T);  // [n_agents][T][2]
}

extern "C" int mppi_set_waypoint_idx(mppi_handle *h, int32_t idx) {
    if (!h) return MPPI_ERR_BAD_ARG;
    if (idx < 0 || (h->n_ref > 0 && idx >= h->n_ref)) FAIL(h, MPPI_ERR_BAD_ARG, "waypoint index %d out of range", idx);
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());
    for (int a = 0; a < h->B; ++a)  // (every agent of a batched handle)
        HIPCHECK(h, hipMemcpy(&h->d_st[a].p, &idx, sizeof(int), hipMemcpyHostToDevice));
    h->idx = idx;
    h->dev_loop_primed = false;
    return MPPI_OK;
}

extern "C" int mppi_get_waypoint_idx(mppi_handle *h, int32_t *idx) {
    if (!h || !idx) return MPPI_ERR_BAD_ARG;
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());
    HIPCHECK(h, hipMemcpy(idx, &h->d_st->p, sizeof(int), hipMemcpyDeviceToHost));
    h->idx = *idx;
    return MPPI_OK;
}

extern "C" int mppi_set_iteration(mppi_handle *h, int64_t iteration) {
    if (!h || iteration < 0) return MPPI_ERR_BAD_ARG;
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());
    long long it = iteration;
    for (int a = 0; a < h->B; ++a)
        HIPCHECK(h, hipMemcpy(&h->d_st[a].iter, &it, sizeof(it), hipMemcpyHostToDevice));
    h->iter = it;
    return MPPI_OK;
}

extern "C" int mppi_set_state(mppi_handle *h, const double *x) {  // x: [n_agents][nx]
    if (!h || !x) return MPPI_ERR_BAD_ARG;
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());
    for (int a = 0; a < h->B; ++a) {
        double v[4] = {0, 0, 0, 0};
        for (int i = 0; i < h->nx; ++i) v[i] = x[(size_t)a * h->nx + i];
        HIPCHECK(h, hipMemcpy(h->d_st[a].x0, v, sizeof(v), hipMemcpyHostToDevice));
    }
    h->dev_loop_primed = false;
    return MPPI_OK;
}

extern "C" int mppi_get_state(mppi_handle *h, double *x) {  // x: [n_agents][nx]
    if (!h || !x) return MPPI_ERR_BAD_ARG;
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());
    for (int a = 0; a < h->B; ++a) {
        double v[4];
        HIPCHECK(h, hipMemcpy(v, h->d_st[a].x0, sizeof(v), hipMemcpyDeviceToHost));
        for (int i = 0; i < h->nx; ++i) x[(size_t)a * h->nx + i] = v[i];
    }
    return MPPI_OK;
}

// ------------------------------------------------------------------------------------------
// parameter packing
// ------------------------------------------------------------------------------------------
template <typename R> static KParams<R> make_params(const mppi_handle *h, const float *eps) {
    const mppi_config &c = h->cfg;
    KParams<R> P;
    memset(&P, 0, sizeof(P));
    P.K = c.K;
    P.T = c.T;
    P.k_offset = c.k_offset;
    const double thr = (1.0 - c.param_exploration) * c.K_global;  // k < thr, mppi_differential_drive.py:116
    double ne = ceil(thr);
    if (ne < 0) ne = 0;
    if (ne > c.K_global) ne = c.K_global;
    P.n_exploit = (int)ne;
    P.n_ref = h->n_ref;
    P.n_obs = c.obstacle_model == MPPI_OBSTACLE_NONE ? 0 : h->n_obs;
    P.window = c.search_window;
    P.model = c.model == MPPI_MODEL_RACECAR ? MODEL_RACE : MODEL_DIFF;
    P.accumulate = c.accumulate_stage_cost;
    P.sequential = c.waypoint_mode == MPPI_WAYPOINT_SEQUENTIAL;
    P.per_rollout = c.waypoint_mode == MPPI_WAYPOINT_PER_ROLLOUT;
    P.obstacle_model = P.n_obs > 0 ? c.obstacle_model : OBS_NONE;
    P.clamp_rollout = c.clamp_rollout;
    P.wrap_stage = c.wrap_yaw_stage;
    P.wrap_term = c.wrap_yaw_terminal;
    // this call's tensor, else the noise ring when one is set (slot = iteration mod slots, picked in the kernels), else Philox
    const float *noise = eps ? eps : h->noise_ring;
    P.eps_slots = eps ? 0 : (h->noise_ring ? h->noise_slots : 0);
    P.use_philox = noise == nullptr;
    P.traj_per_block = h->traj_per_block;
    P.seed_lo = (unsigned)(c.seed & 0xffffffffu);
    P.seed_hi = (unsigned)(c.seed >> 32);
    P.dt = (R)c.delta_t;
    P.umax0 = (R)c.u_max[0];
    P.umax1 = (R)c.u_max[1];
    P.wheel_base = (R)c.wheel_base;
    const double beta = c.beta_mode == MPPI_BETA_INV_EXPLORATION ? 1.0 / c.param_exploration
                        : c.beta_mode == MPPI_BETA_INV_LAMBDA    ? 1.0 / c.param_lambda
                                                                 : c.param_lambda;
    P.beta = (R)beta;
    P.gamma = (R)(c.param_lambda * (1.0 - c.param_alpha));  // :74
    P.penalty = (R)c.collision_penalty;
    P.two_pi = (R)(2.0 * M_PI);
    const double det = c.sigma[0] * c.sigma[3] - c.sigma[1] * c.sigma[2];
    const double si[4] = {c.sigma[3] / det, -c.sigma[1] / det, -c.sigma[2] / det, c.sigma[0] / det};
    for (int i = 0; i < 4; ++i) {
        P.sinv[i] = (R)si[i];
        P.ws[i] = (R)c.stage_cost_weight[i];
        P.wt[i] = (R)c.terminal_cost_weight[i];
    }
    // vehicle outline, mppi_race_car_obstacle.py:256-264
    const double vw = c.vehicle_w * c.safety_margin, vl = c.vehicle_l * c.safety_margin;
    const double sx[9] = {-0.5 * vl, -0.5 * vl, 0.0, 0.5 * vl, 0.5 * vl, 0.5 * vl, 0.0, -0.5 * vl, -0.5 * vl};
    const double sy[9] = {0.0, 0.5 * vw, 0.5 * vw, 0.5 * vw, 0.0, -0.5 * vw, -0.5 * vw, -0.5 * vw, 0.0};
    for (int i = 0; i < 9; ++i) {
        P.shape_x[i] = (R)sx[i];
        P.shape_y[i] = (R)sy[i];
    }
    const double l00 = sqrt(c.sigma[0]), l10 = c.sigma[2] / l00, l11 = sqrt(c.sigma[3] - l10 * l10);
    P.chol[0] = (float)l00;
    P.chol[1] = (float)l10;
    P.chol[2] = (float)l11;
    P.ref = (const R *)h->d_ref;
    P.obs = (const R *)h->d_obs;
    P.u = (const R *)h->d_u;
    P.eps = noise;
    P.S = (R *)h->d_S;
    P.pout = h->d_pout;
    P.st = h->d_st;
    P.noise_stream = c.noise_stream;
    P.slots = h->slots;
    P.n_agents = h->B;
    P.layout = h->layout;
    P.heads = (R *)h->d_heads;
    // the kernels that can resolve the sequential index in one launch -- while that index can still move: once it sits
    // on the last waypoint (it only grows) every search window holds one candidate and the lean kernels serve
    P.hyp = h->hyp && !(h->idx_valid && h->n_ref > 0 && h->idx >= h->n_ref - 1);
    P.lb_seq = 0;
    P.hyp_slots = h->d_hyp_slots;
    return P;
}

static FinalizeParams make_finalize(const mppi_handle *h, const void *partials, int n_part, int plant) {
    const mppi_config &c = h->cfg;
    FinalizeParams F;
    memset(&F, 0, sizeof(F));
    F.T = c.T;
    F.K = c.K;
    F.n_part = n_part;
    F.filter_mode = c.filter_mode;
    F.filter_window = c.filter_window;
    F.clamp_u = c.clamp_u_after_update;
    F.raise_at_path_end = c.raise_at_path_end;
    F.model = c.model == MPPI_MODEL_RACECAR ? MODEL_RACE : MODEL_DIFF;
    F.sequential = c.waypoint_mode == MPPI_WAYPOINT_SEQUENTIAL;
    F.plant = plant;
    F.n_ref = h->n_ref;
    F.window = c.search_window;
    F.is_f64 = h->f64;
    F.count_hits = c.obstacle_model != MPPI_OBSTACLE_NONE && h->n_obs > 0;
    F.beta = c.beta_mode == MPPI_BETA_INV_EXPLORATION ? 1.0 / c.param_exploration
             : c.beta_mode == MPPI_BETA_INV_LAMBDA    ? 1.0 / c.param_lambda
                                                      : c.param_lambda;
    if (!h->f64) F.beta = (double)(float)F.beta;  // the block partials were scaled with the f32 rate
    F.dt = c.delta_t;
    F.wheel_base = c.wheel_base;
    F.umax0 = c.u_max[0];
    F.umax1 = c.u_max[1];
    F.partials = partials;
    F.heads = partials == h->d_partials2 ? h->d_heads2 : h->d_heads;
    F.u = h->d_u;
    F.u_out = h->d_u;
    F.u_before = h->d_uhist;
    F.ref = h->d_ref;
    F.pout = h->d_pout;
    F.st = h->d_st;
    F.st_out = h->d_st;
    F.res = h->d_res;
    F.slots = h->slots;
    F.n_agents = h->B;
    F.res_stride = h->res_bytes;
    F.u0_trace = nullptr;
    F.hyp = h->hyp && partials == h->d_partials && !(h->idx_valid && h->n_ref > 0 && h->idx >= h->n_ref - 1);
    F.hyp_blocks = h->n_part;
    F.hyp_slots = h->d_hyp_slots;
    F.lb_seq = 0;
    F.pad_lb = 0;
    return F;
}

// Timing (mppi_enable_timing): every launch group of a slot is bracketed by a pair of events on the
// launch stream -- [rollout] [reduce/merge] [finalize] -- plus one EMPTY pair that measures what a pair
// of hipEventRecord costs on this stream (several microseconds); mppi_last_kernel_ms reports the
// averages with that calibration subtracted.
constexpr int EV_PER_SLOT = 8;
constexpr size_t EV_MAX = 8u * 200000u;

static hipEvent_t next_event(mppi_handle *h) {
    if (h->ev_used == h->ev.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        h->ev.push_back(e);
    }
    return h->ev[h->ev_used++];
}

static bool timing_on(const mppi_handle *h) { return h->timing && h->ev_used + EV_PER_SLOT <= EV_MAX; }

// Softmin partial records of this handle's samples: rollout (+ reduce when not fused), and for large K a
// 64:1 merge so that the finalize block never reads more than MAX_FINAL_PARTS records.
constexpr int MAX_FINAL_PARTS = 256;   // = MERGE_MAX_RECORDS of the kernels (ABI records: one per rank)
constexpr int MAX_DIRECT_PARTS = 512;  // block records k_finalize merges itself (two windows of 256)

static void launch_mlp(mppi_handle *h, const KParams<float> &P, hipStream_t s) {
    launch_rollout_mlp(P, h->mlp, h->d_partials, s);
}
static void launch_mlp(mppi_handle *, const KParams<double> &, hipStream_t) {}  // rejected at create

template <typename R>
static void launch_front(mppi_handle *h, const KParams<R> &P, double beta, hipStream_t s, const void **recs,
                         const void **heads, int *n_recs, bool tm) {
    if (tm) hipEventRecord(next_event(h), s);
    const bool mlp = h->cfg.model == MPPI_MODEL_DIFFDRIVE_MLP;
    h->n_rollout_launches += h->rollout_repeats;
    for (int rep = 0; rep < h->rollout_repeats; ++rep) {
        if (mlp) launch_mlp(h, P, s);
        else if (h->fused) launch_rollout_fused<R>(P, h->d_partials, s);
        else launch_rollout<R>(P, s);
    }
    h->rollout_kernel = mlp ? mlp_kernel_name(h->mlp) : last_rollout_kernel();
    if (tm) hipEventRecord(next_event(h), s);
    if (tm) hipEventRecord(next_event(h), s);
    if (!mlp && !h->fused) launch_reduce<R>(P, h->d_partials, h->n_blocks, s);
    // (records per agent this launch left: the handle's count, or the streaming kernel's when that one served)
    const int n_part = (!mlp && h->fused) ? fused_records<R>(P) : h->n_part;
    *recs = h->d_partials;
    *heads = h->d_heads;
    *n_recs = n_part;
    if (n_part > MAX_DIRECT_PARTS) {
        const int group = n_part > 64 * MAX_FINAL_PARTS ? MAX_FINAL_PARTS : 64;
        launch_merge<R>(h->d_partials, h->d_heads, n_part, group, h->cfg.T, beta, h->d_partials2, h->d_heads2, false, s);
        *recs = h->d_partials2;
        *heads = h->d_heads2;
        *n_recs = (n_part + group - 1) / group;
    }
    if (tm) hipEventRecord(next_event(h), s);
}

template <typename R>
static void launch_back(mppi_handle *h, const FinalizeParams &F, bool abi_recs, hipStream_t s, bool tm) {
    if (tm) hipEventRecord(next_event(h), s);
    ++h->n_finalize_launches;
    launch_finalize<R>(F, abi_recs, s);
    if (tm) hipEventRecord(next_event(h), s);
    if (tm) hipEventRecord(next_event(h), s);  // empty pair: the cost of the bracketing itself
    if (tm) hipEventRecord(next_event(h), s);
}

static void arm_exchange(mppi_handle *h, FinalizeParams &F) {
    if (h->x_nranks <= 1) return;
    F.x_nranks = h->x_nranks;
    F.x_rank = h->x_rank;
    F.x_seq = ++h->xseq;
    F.x_timeout = h->x_timeout;
    F.x_peers = h->d_xpeers;
    F.x_err = h->d_xerr;
}

// rollout (-> reduce) -> finalize
template <typename R>
static void launch_slot(mppi_handle *h, const KParams<R> &P, FinalizeParams F, hipStream_t s) {
    const bool tm = timing_on(h);
    arm_exchange(h, F);
    if (P.hyp) {  // the look-back words of this launch pair carry its own tag (lb_tag)
        if ((++h->lb_seq & 0xffffffu) == 0u) ++h->lb_seq;
        KParams<R> Pl = P;
        Pl.lb_seq = F.lb_seq = h->lb_seq;
        launch_front<R>(h, Pl, F.beta, s, &F.partials, &F.heads, &F.n_part, tm);
    } else {
        launch_front<R>(h, P, F.beta, s, &F.partials, &F.heads, &F.n_part, tm);
    }
    launch_back<R>(h, F, false, s, tm);
}

static void collect_timing(mppi_handle *h) {
    double acc[4] = {0, 0, 0, 0};
    const size_t slots = h->ev_used / EV_PER_SLOT;
    for (size_t i = 0; i < slots; ++i)
        for (int j = 0; j < 4; ++j) {
            float ms = 0;
            hipEventElapsedTime(&ms, h->ev[EV_PER_SLOT * i + 2 * j], h->ev[EV_PER_SLOT * i + 2 * j + 1]);
            acc[j] += ms;
        }
    const double cal = slots ? acc[3] / slots : 0.0;
    for (int j = 0; j < 3; ++j) {
        const double v = slots ? acc[j] / slots - cal : 0.0;
        h->last_ms[j] = (float)(v > 0 ? v : 0);
    }
    h->last_ms[3] = (float)cal;
}

static int check_ready(mppi_handle *h, const char *who) {
    if (!h) return MPPI_ERR_BAD_ARG;
    if (!h->d_ref || h->n_ref < 1) FAIL(h, MPPI_ERR_STATE, "%s: ref_path has not been set", who);
    if (h->cfg.obstacle_model != MPPI_OBSTACLE_NONE && h->n_obs > 0 && !h->d_obs)
        FAIL(h, MPPI_ERR_STATE, "%s: obstacles not uploaded", who);
    if (h->cfg.model == MPPI_MODEL_DIFFDRIVE_MLP && !h->mlp_set)
        FAIL(h, MPPI_ERR_STATE, "%s: mppi_set_mlp has not been called", who);
    return MPPI_OK;
}

// entry points that serve one agent only (the host-in-the-loop and split steps, visualisation, the exchange)
#define SINGLE_AGENT_ONLY(h, who)                                                                                     \
    do {                                                                                                              \
        if ((h)->B > 1) FAIL(h, MPPI_ERR_UNSUPPORTED, "%s serves single-agent handles (n_agents = %d)", who, (h)->B); \
    } while (0)

static void fill_stats(mppi_handle *h, mppi_stats *stats) {
    if (!stats) return;
    if (h->timing && h->ev_used >= EV_PER_SLOT) {  // kernel_us: the window's averages as of this call
        if (hipEventSynchronize(h->ev[h->ev_used - 1]) == hipSuccess) collect_timing(h);
        else (void)hipGetLastError();
    }
    const StepResult *r = h->h_res;
    stats->rho = r->rho;
    stats->eta = r->eta;
    stats->ess = r->ess;
    stats->idx_start = r->idx_start;
    stats->idx_after = r->idx_after;
    stats->path_end = r->path_end;
    stats->rounds = r->rounds;
    stats->iteration = r->iter;
    stats->n_collided = r->n_collided;
    stats->reserved = 0;
    stats->iter_us = h->last_iter_us;
    stats->kernel_us = h->timing ? 1e3 * ((double)h->last_ms[0] + h->last_ms[1] + h->last_ms[2]) : 0.0;
}

// The x0 call (mppi_differential_drive.py:96-99 / mppi_race_car.py:61): nearest waypoint of the observed state
// in the window that starts at prev_way_point_idx, first minimum.  Same f64 arithmetic as k_set_state, on the
// host so that the synchronous step needs no extra launch; the result travels as a kernel argument.
static int host_x0_call(const mppi_handle *h, const double *x0) {
    const int p = h->idx, n = h->n_ref, w = h->cfg.search_window;
    int best_j = 0;
    double best = INFINITY;
    for (int j = 0; j < w && p + j < n; ++j) {
        const double dx = x0[0] - h->ref_host[4 * (size_t)(p + j)], dy = x0[1] - h->ref_host[4 * (size_t)(p + j) + 1];
        const double d = dx * dx + dy * dy;
        if (d < best) { best = d; best_j = j; }
    }
    return p + best_j;
}

// wait for the result of the launches just enqueued: poll the completion word the finalize kernel writes into
// mapped host memory (saves the device-to-host copy launch and the stream synchronisation), or copy + synchronise
static int wait_result(mppi_handle *h, long long seq, hipStream_t s) {
    if (seq) {
        volatile long long *flag = &h->h_res->seq;
        // The poll saves a copy launch and a stream synchronisation (~7 us) on short calls, and on long ones the wake-up of
        // hipStreamSynchronize (up to 0.4 ms per call on some boxes: 200 iterations per call 11.0 us per iteration against
        // 9.2 at 50).  So the host spins for up to 20 ms -- two episodes of the reference driver's run at config 2 -- and only
        // then hands the wait to the runtime (config 5's tens of milliseconds per call; it also surfaces a failed launch).
        // (Sleeping through an ESTIMATE of the call's duration was tried: the pace of the previous call is the wrong
        // estimate across a phase change, and an overslept call costs more than the spin saves.)
        const double t0 = now_s();
        for (long long spins = 0; *flag != seq; ++spins) {
            __builtin_ia32_pause();
            if ((spins & 1023) == 1023 && now_s() - t0 > 0.020) {
                HIPCHECK(h, hipStreamSynchronize(s));
                HIPCHECK(h, hipGetLastError());
                if (*flag != seq) FAIL(h, MPPI_ERR_HIP, "the finalize kernel never published its result");
                break;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return MPPI_OK;
    }
    HIPCHECK(h, hipMemcpyAsync(h->h_res, h->d_res, h->res_bytes, hipMemcpyDeviceToHost, s));
    HIPCHECK(h, hipStreamSynchronize(s));
    HIPCHECK(h, hipGetLastError());
    return MPPI_OK;
}

// (the RCCL carrier of a K-sharded handle, further down)
template <typename R>
static int step_rccl(mppi_handle *h, const double *x0, const double *x0_dev, const float *eps, double *u_out, double *u0_out,
                     mppi_stats *stats, hipStream_t s);
template <typename R>
static int closed_loop_rccl(mppi_handle *h, int n_iters, double *u0_trace, mppi_stats *stats, hipStream_t s);

// x0: host (the observed state travels as kernel arguments) or, when null, x0_dev: device memory (a small kernel moves
// it into the controller state and makes the x0 call)
template <typename R>
static int step_impl(mppi_handle *h, const double *x0, const double *x0_dev, const float *eps, double *u_out, double *u0_out,
                     mppi_stats *stats, hipStream_t s) {
    if (h->rccl_comm && h->x_nranks <= 1) return step_rccl<R>(h, x0, x0_dev, eps, u_out, u0_out, stats, s);
    const double t_call = now_s();
    KParams<R> P = make_params<R>(h, eps);
    FinalizeParams F = make_finalize(h, h->d_partials, h->n_part, 0);
    const bool by_args = x0 && h->idx_valid && h->by_args_ok;
    if (!x0) {
        if (!h->idx_valid) FAIL(h, MPPI_ERR_STATE, "mppi_step_device_x0 after an asynchronous split step: call mppi_sync_result first");
        launch_set_state_dev<R>(P, x0_dev, h->nx, s);
    } else if (by_args) {
        P.use_args = F.use_args = 1;
        P.c_arg = F.c_arg = host_x0_call(h, x0);
        P.hyp = F.hyp = h->hyp && P.c_arg < h->n_ref - 1;  // (a window of one candidate: nothing can move)
        for (int i = 0; i < 4; ++i) P.x0_arg[i] = F.x0_arg[i] = x0[i];
    } else {
        launch_set_state<R>(P, x0, s);
    }
    if (h->poll) F.res = h->res_mapped;
    for (int round = 0;; ++round) {
        F.seq = h->poll ? ++h->seq : 0;
        launch_slot<R>(h, P, F, s);
        HIPCHECK(h, hipGetLastError());  // a refused launch would otherwise show only as the poll's timeout
        int rc = wait_result(h, F.seq, s);
        if (rc) return rc;
        if (h->h_res->status != STATUS_NEED_ROUND) break;
        if (round > h->cfg.K + 1) FAIL(h, MPPI_ERR_STATE, "waypoint speculation did not converge");
        P.use_args = 0;  // repair rounds take the state the finalize kernel left in *st
    }
    h->dev_loop_primed = false;
    h->last_eps = eps;
    h->last_philox = eps == nullptr;
    h->idx = h->h_res->idx_after;
    h->idx_valid = true;
    h->last_iter_us = 1e6 * (now_s() - t_call);
    fill_stats(h, stats);
    if (h->h_res->status == STATUS_EXCHANGE_FAILED)
        FAIL(h, MPPI_ERR_COMM, "peer-to-peer exchange: a rank did not arrive within the timeout");
    if (h->h_res->status == STATUS_PATH_END)
        FAIL(h, MPPI_ERR_PATH_END, "[ERROR] Reached the end of the reference path.");
    h->iter = h->h_res->iter;
    const double *ru = reinterpret_cast<const double *>(h->h_res + 1);
    if (u_out) memcpy(u_out, ru, sizeof(double) * 2 * h->cfg.T);
    if (u0_out) { u0_out[0] = h->h_res->u0[0]; u0_out[1] = h->h_res->u0[1]; }
    return MPPI_OK;
}

extern "C" int mppi_step(mppi_handle *h, const double *x0, const float *eps, double *u_out, double *u0_out,
                         mppi_stats *stats, void *stream) {
    int rc = check_ready(h, "mppi_step");
    if (rc) return rc;
    SINGLE_AGENT_ONLY(h, "mppi_step");
    if (!x0) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_step: x0 is null");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    double x[4] = {0, 0, 0, 0};
    for (int i = 0; i < h->nx; ++i) x[i] = x0[i];
    return h->f64 ? step_impl<double>(h, x, nullptr, eps, u_out, u0_out, stats, (hipStream_t)stream)
                  : step_impl<float>(h, x, nullptr, eps, u_out, u0_out, stats, (hipStream_t)stream);
}

extern "C" int mppi_step_device_x0(mppi_handle *h, const double *x0_device, const float *eps, double *u_out, double *u0_out,
                                   mppi_stats *stats, void *stream) {
    int rc = check_ready(h, "mppi_step_device_x0");
    if (rc) return rc;
    SINGLE_AGENT_ONLY(h, "mppi_step_device_x0");
    if (!x0_device) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_step_device_x0: x0 is null");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    return h->f64 ? step_impl<double>(h, nullptr, x0_device, eps, u_out, u0_out, stats, (hipStream_t)stream)
                  : step_impl<float>(h, nullptr, x0_device, eps, u_out, u0_out, stats, (hipStream_t)stream);
}

extern "C" int mppi_partial_len(const mppi_handle *h, int32_t *n) {
    if (!h || !n) return MPPI_ERR_BAD_ARG;
    *n = partial_len(h->cfg.T);
    return MPPI_OK;
}

// rollout (-> reduce) -> merges down to this rank's ONE record {rho, eta, eta2, W[T][2]} in f64 at `partial` (device)
template <typename R>
static void launch_rank_record(mppi_handle *h, const KParams<R> &P, double beta, double *partial, hipStream_t s, bool tm) {
    const void *recs, *heads;
    int n_recs;
    launch_front<R>(h, P, beta, s, &recs, &heads, &n_recs, tm);
    if (n_recs > MAX_FINAL_PARTS) {  // (k_merge takes 256 records per workgroup)
        launch_merge<R>(recs, heads, n_recs, 64, h->cfg.T, beta, h->d_partials2, h->d_heads2, false, s);
        recs = h->d_partials2;
        heads = h->d_heads2;
        n_recs = (n_recs + 63) / 64;
    }
    launch_merge<R>(recs, heads, n_recs, n_recs, h->cfg.T, beta, partial, nullptr, true, s);
}

template <typename R>
static int begin_impl(mppi_handle *h, const double *x0, const float *eps, double *partial, hipStream_t s) {
    KParams<R> P = make_params<R>(h, eps);
    const FinalizeParams F = make_finalize(h, h->d_partials, h->n_part, 0);
    // closed loop on the device: the previous end_async already made the x0 call for the new state
    if (x0 || !h->dev_loop_primed) launch_set_state<R>(P, x0, s);
    h->dev_loop_primed = x0 == nullptr;
    h->slot_timed = timing_on(h);
    launch_rank_record<R>(h, P, F.beta, partial, s, h->slot_timed);
    HIPCHECK(h, hipGetLastError());
    h->last_eps = eps;
    h->last_philox = eps == nullptr;
    h->begun = true;
    return MPPI_OK;
}

extern "C" int mppi_step_begin(mppi_handle *h, const double *x0, const float *eps, double *partial, void *stream) {
    int rc = check_ready(h, "mppi_step_begin");
    if (rc) return rc;
    SINGLE_AGENT_ONLY(h, "mppi_step_begin");
    if (!partial) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_step_begin: null argument");
    if (h->cfg.waypoint_mode == MPPI_WAYPOINT_SEQUENTIAL)
        FAIL(h, MPPI_ERR_UNSUPPORTED, "the split step needs MPPI_WAYPOINT_FROZEN (no cross-sample waypoint state)");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    double x[4] = {0, 0, 0, 0};
    if (x0)
        for (int i = 0; i < h->nx; ++i) x[i] = x0[i];
    return h->f64 ? begin_impl<double>(h, x0 ? x : nullptr, eps, partial, (hipStream_t)stream)
                  : begin_impl<float>(h, x0 ? x : nullptr, eps, partial, (hipStream_t)stream);
}

extern "C" int mppi_step_end(mppi_handle *h, const double *partials, int32_t nranks, double *u_out, double *u0_out,
                             mppi_stats *stats, void *stream) {
    int rc = check_ready(h, "mppi_step_end");
    if (rc) return rc;
    SINGLE_AGENT_ONLY(h, "mppi_step_end");
    if (!partials || nranks < 1 || nranks > MAX_FINAL_PARTS)
        FAIL(h, MPPI_ERR_BAD_ARG, "mppi_step_end: bad partials/nranks (1..%d)", MAX_FINAL_PARTS);
    if (!h->begun) FAIL(h, MPPI_ERR_STATE, "mppi_step_end without mppi_step_begin");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    FinalizeParams F = make_finalize(h, partials, nranks, 0);
    if (h->f64) launch_back<double>(h, F, true, s, h->slot_timed);
    else launch_back<float>(h, F, true, s, h->slot_timed);
    HIPCHECK(h, hipMemcpyAsync(h->h_res, h->d_res, h->res_bytes, hipMemcpyDeviceToHost, s));
    HIPCHECK(h, hipStreamSynchronize(s));
    HIPCHECK(h, hipGetLastError());
    h->begun = false;
    h->dev_loop_primed = false;  // no plant ran: the next device-state step makes its own x0 call
    h->idx = h->h_res->idx_after;
    fill_stats(h, stats);
    if (h->h_res->status == STATUS_PATH_END)
        FAIL(h, MPPI_ERR_PATH_END, "[ERROR] Reached the end of the reference path.");
    h->iter = h->h_res->iter;
    const double *ru = reinterpret_cast<const double *>(h->h_res + 1);
    if (u_out) memcpy(u_out, ru, sizeof(double) * 2 * h->cfg.T);
    if (u0_out) { u0_out[0] = h->h_res->u0[0]; u0_out[1] = h->h_res->u0[1]; }
    return MPPI_OK;
}

extern "C" int mppi_step_end_async(mppi_handle *h, const double *partials, int32_t nranks, void *stream) {
    int rc = check_ready(h, "mppi_step_end_async");
    if (rc) return rc;
    SINGLE_AGENT_ONLY(h, "mppi_step_end_async");
    if (!partials || nranks < 1 || nranks > MAX_FINAL_PARTS)
        FAIL(h, MPPI_ERR_BAD_ARG, "mppi_step_end_async: bad partials/nranks (1..%d)", MAX_FINAL_PARTS);
    if (!h->begun) FAIL(h, MPPI_ERR_STATE, "mppi_step_end_async without mppi_step_begin");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    FinalizeParams F = make_finalize(h, partials, nranks, 1);  // plant on: the state advances on the device
    if (h->f64) launch_back<double>(h, F, true, (hipStream_t)stream, h->slot_timed);
    else launch_back<float>(h, F, true, (hipStream_t)stream, h->slot_timed);
    HIPCHECK(h, hipGetLastError());
    h->begun = false;
    h->idx_valid = false;  // the waypoint index now advances on the device until mppi_sync_result
    return MPPI_OK;
}

extern "C" int mppi_sync_result(mppi_handle *h, double *u_out, double *u0_out, mppi_stats *stats, void *stream) {
    if (!h) return MPPI_ERR_BAD_ARG;
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    HIPCHECK(h, hipMemcpyAsync(h->h_res, h->d_res, h->res_bytes, hipMemcpyDeviceToHost, s));
    HIPCHECK(h, hipStreamSynchronize(s));
    h->idx = h->h_res->idx_after;
    h->idx_valid = true;
    fill_stats(h, stats);
    if (h->h_res->status == STATUS_PATH_END)
        FAIL(h, MPPI_ERR_PATH_END, "[ERROR] Reached the end of the reference path.");
    h->iter = h->h_res->iter;
    const double *ru = reinterpret_cast<const double *>(h->h_res + 1);
    if (u_out) memcpy(u_out, ru, sizeof(double) * 2 * h->cfg.T);
    if (u0_out) { u0_out[0] = h->h_res->u0[0]; u0_out[1] = h->h_res->u0[1]; }
    return MPPI_OK;
}

extern "C" int mppi_get_costs(mppi_handle *h, double *S) {
    if (!h || !S) return MPPI_ERR_BAD_ARG;
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());
    return download_real(h, S, h->d_S, (size_t)h->B * h->cfg.K);  // [n_agents][K]
}

extern "C" int mppi_get_weights(mppi_handle *h, double *w) {
    if (!h || !w) return MPPI_ERR_BAD_ARG;
    SINGLE_AGENT_ONLY(h, "mppi_get_weights");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());
    if (!h->d_w) HIPCHECK(h, hipMalloc((void **)&h->d_w, sizeof(double) * h->cfg.K));
    if (h->f64) {
        KParams<double> P = make_params<double>(h, nullptr);
        launch_weights<double>(P, h->h_res->rho, h->h_res->eta, h->d_w, nullptr);
    } else {
        KParams<float> P = make_params<float>(h, nullptr);
        launch_weights<float>(P, h->h_res->rho, h->h_res->eta, h->d_w, nullptr);
    }
    HIPCHECK(h, hipDeviceSynchronize());
    HIPCHECK(h, hipMemcpy(w, h->d_w, sizeof(double) * h->cfg.K, hipMemcpyDeviceToHost));
    return MPPI_OK;
}

extern "C" int mppi_sample_epsilon(mppi_handle *h, int64_t iteration, float *eps_out, void *stream) {
    if (!h || !eps_out || iteration < 0) return MPPI_ERR_BAD_ARG;
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    KParams<float> P = make_params<float>(h, nullptr);
    for (int a = 0; a < h->B; ++a)  // [n_agents][K][T][2]: agent a draws with stream word noise_stream + a
        launch_sample(P.seed_lo, P.seed_hi, (unsigned)iteration, h->cfg.K, h->cfg.T, h->cfg.k_offset, P.chol,
                      eps_out + (size_t)a * h->cfg.K * h->cfg.T * 2, (hipStream_t)stream, (unsigned)(h->cfg.noise_stream + a));
    HIPCHECK(h, hipGetLastError());
    return MPPI_OK;
}

extern "C" int mppi_set_noise_ring(mppi_handle *h, const float *eps_ring, int32_t n_slots) {
    if (!h) return MPPI_ERR_BAD_ARG;
    if (!eps_ring || n_slots == 0) {  // back to the in-kernel sampler
        h->noise_ring = nullptr;
        h->noise_slots = 0;
        return MPPI_OK;
    }
    if (n_slots < 1 || (n_slots & (n_slots - 1)) != 0)
        FAIL(h, MPPI_ERR_BAD_ARG, "mppi_set_noise_ring: n_slots must be a power of two (got %d)", n_slots);
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, eps_ring) != hipSuccess || at.type == hipMemoryTypeHost || at.type == hipMemoryTypeUnregistered ||
        at.device != h->cfg.device) {
        (void)hipGetLastError();
        FAIL(h, MPPI_ERR_BAD_ARG, "mppi_set_noise_ring: the ring must be device memory on the handle's GPU (device %d)", h->cfg.device);
    }
    h->noise_ring = eps_ring;
    h->noise_slots = n_slots;
    return MPPI_OK;
}

extern "C" int mppi_rollout_viz(mppi_handle *h, float *optimal_traj, float *sampled_traj, void *stream) {
    int rc = check_ready(h, "mppi_rollout_viz");
    if (rc) return rc;
    SINGLE_AGENT_ONLY(h, "mppi_rollout_viz");
    if (h->iter < 1) FAIL(h, MPPI_ERR_STATE, "mppi_rollout_viz before the first mppi_step");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    const float *eps = h->last_philox ? nullptr : h->last_eps;
    const int T = h->cfg.T;
    if (h->cfg.model == MPPI_MODEL_DIFFDRIVE_MLP) {  // (fp32 only, checked at create)
        KParams<float> P = make_params<float>(h, eps);
        const float *hist = (const float *)h->d_uhist;
        launch_viz_mlp(P, h->mlp, hist, hist + 2 * T, h->iter - 1, optimal_traj, sampled_traj, (hipStream_t)stream);
    } else if (h->f64) {
        KParams<double> P = make_params<double>(h, eps);
        const double *hist = (const double *)h->d_uhist;
        launch_viz<double>(P, hist, hist + 2 * T, h->iter - 1, optimal_traj, sampled_traj, (hipStream_t)stream);
    } else {
        KParams<float> P = make_params<float>(h, eps);
        const float *hist = (const float *)h->d_uhist;
        launch_viz<float>(P, hist, hist + 2 * T, h->iter - 1, optimal_traj, sampled_traj, (hipStream_t)stream);
    }
    HIPCHECK(h, hipGetLastError());
    return MPPI_OK;
}


// ------------------------------------------------------------------------------------------
// RCCL carrier of the same exchange inside the library (include/mppi_hip.h, mppi_comm_init):
// one ncclAllGather of the per-rank record per iteration, enqueued on the caller's stream between the rollout and the
// finalize launches -- no host code per iteration beyond the three enqueues.  librccl.so.1 is resolved at run time
// (dlopen by soname: in a PyTorch-ROCm process that is the copy torch has already loaded), so the library has no
// link-time dependency on RCCL and single-GPU users never load it.
// ------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, RcclUniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};
RcclApi rccl_load() {
    RcclApi a;
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (a.lib) break;
    }
    if (!a.lib) {
        a.why = std::string("librccl.so.1 could not be loaded: ") + dlerror();
        return a;
    }
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.lib, "ncclAllGather"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy || !a.GetErrorString) {
        a.why = "librccl.so.1 lacks an expected symbol";
        a.lib = nullptr;
    }
    return a;
}
RcclApi &rccl_api() {
    static RcclApi a = rccl_load();  // (one load per process; the language makes the initialisation thread-safe)
    return a;
}
}  // namespace

static void rccl_release(mppi_handle *h) {
    if (h->rccl_comm) rccl_api().CommDestroy(h->rccl_comm);
    if (h->d_rccl_part) hipFree(h->d_rccl_part);
    if (h->d_rccl_gath) hipFree(h->d_rccl_gath);
    h->rccl_comm = nullptr;
    h->d_rccl_part = h->d_rccl_gath = nullptr;
    h->rccl_nranks = 0;
}

#define RCCLCHECK(h, call)                                                                                   \
    do {                                                                                                     \
        const int _r = (call);                                                                               \
        if (_r != 0) FAIL(h, MPPI_ERR_COMM, "%s failed: %s", #call, rccl_api().GetErrorString(_r));          \
    } while (0)

extern "C" int mppi_comm_unique_id_bytes(void) { return (int)sizeof(RcclUniqueId); }

extern "C" int mppi_comm_unique_id(void *id_out) {
    if (!id_out) return MPPI_ERR_BAD_ARG;
    RcclApi &a = rccl_api();
    if (!a.lib) {
        g_create_error = a.why;
        return MPPI_ERR_COMM;
    }
    if (a.GetUniqueId(id_out) != 0) {
        g_create_error = "ncclGetUniqueId failed";
        return MPPI_ERR_COMM;
    }
    return MPPI_OK;
}

extern "C" int mppi_comm_init(mppi_handle *h, const void *unique_id, int32_t rank, int32_t nranks) {
    if (!h || !unique_id) return MPPI_ERR_BAD_ARG;
    SINGLE_AGENT_ONLY(h, "mppi_comm_init");
    if (nranks < 1 || nranks > MAX_FINAL_PARTS || rank < 0 || rank >= nranks)
        FAIL(h, MPPI_ERR_BAD_ARG, "mppi_comm_init: rank %d of %d (1..%d ranks)", rank, nranks, MAX_FINAL_PARTS);
    if (h->cfg.waypoint_mode == MPPI_WAYPOINT_SEQUENTIAL)
        FAIL(h, MPPI_ERR_UNSUPPORTED, "the exchange needs MPPI_WAYPOINT_FROZEN (no cross-sample waypoint state)");
    RcclApi &a = rccl_api();
    if (!a.lib) FAIL(h, MPPI_ERR_COMM, "mppi_comm_init: %s", a.why.c_str());
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    rccl_release(h);
    RcclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    RCCLCHECK(h, a.CommInitRank(&h->rccl_comm, nranks, id, rank));
    const size_t len = (size_t)partial_len(h->cfg.T);
    HIPCHECK(h, hipMalloc((void **)&h->d_rccl_part, sizeof(double) * len));
    HIPCHECK(h, hipMalloc((void **)&h->d_rccl_gath, sizeof(double) * len * nranks));
    h->rccl_rank = rank;
    h->rccl_nranks = nranks;
    return MPPI_OK;
}

// One iteration on the stream with the collective in the middle: rollout -> this rank's record -> ncclAllGather ->
// finalize over the gathered records (identical on every rank, so u stays replicated without a broadcast).
template <typename R>
static int rccl_iteration(mppi_handle *h, const KParams<R> &P, int plant, double *u0_trace_dev, hipStream_t s) {
    const bool tm = timing_on(h);
    const FinalizeParams F0 = make_finalize(h, h->d_partials, h->n_part, 0);
    launch_rank_record<R>(h, P, F0.beta, h->d_rccl_part, s, tm);
    RCCLCHECK(h, rccl_api().AllGather(h->d_rccl_part, h->d_rccl_gath, (size_t)partial_len(h->cfg.T), 8 /* ncclFloat64 */,
                                      h->rccl_comm, s));
    FinalizeParams F = make_finalize(h, h->d_rccl_gath, h->rccl_nranks, plant);
    F.u0_trace = u0_trace_dev;
    launch_back<R>(h, F, true, s, tm);
    return MPPI_OK;
}

// outputs of the last finished iteration -> host (copy + synchronise; the collective paces the stream anyway)
static int rccl_fetch(mppi_handle *h, double *u_out, double *u0_out, mppi_stats *stats, hipStream_t s) {
    HIPCHECK(h, hipGetLastError());
    HIPCHECK(h, hipMemcpyAsync(h->h_res, h->d_res, h->res_bytes, hipMemcpyDeviceToHost, s));
    HIPCHECK(h, hipStreamSynchronize(s));
    HIPCHECK(h, hipGetLastError());
    h->idx = h->h_res->idx_after;
    h->idx_valid = true;
    fill_stats(h, stats);
    if (h->h_res->status == STATUS_PATH_END) FAIL(h, MPPI_ERR_PATH_END, "[ERROR] Reached the end of the reference path.");
    h->iter = h->h_res->iter;
    const double *ru = reinterpret_cast<const double *>(h->h_res + 1);
    if (u_out) memcpy(u_out, ru, sizeof(double) * 2 * h->cfg.T);
    if (u0_out) { u0_out[0] = h->h_res->u0[0]; u0_out[1] = h->h_res->u0[1]; }
    return MPPI_OK;
}

template <typename R>
static int step_rccl(mppi_handle *h, const double *x0, const double *x0_dev, const float *eps, double *u_out, double *u0_out,
                     mppi_stats *stats, hipStream_t s) {
    const KParams<R> P = make_params<R>(h, eps);
    if (x0) launch_set_state<R>(P, x0, s);
    else launch_set_state_dev<R>(P, x0_dev, h->nx, s);  // the observed state in device memory (mppi_step_device_x0)
    if (int rc = rccl_iteration<R>(h, P, 0, nullptr, s)) return rc;
    h->dev_loop_primed = false;
    h->last_eps = eps;
    h->last_philox = eps == nullptr;
    return rccl_fetch(h, u_out, u0_out, stats, s);
}

template <typename R>
static int closed_loop_rccl(mppi_handle *h, int n_iters, double *u0_trace, mppi_stats *stats, hipStream_t s) {
    const KParams<R> P = make_params<R>(h, nullptr);
    double *trace = nullptr;
    if (u0_trace) {
        if (h->trace_cap < n_iters) {
            if (h->d_trace) HIPCHECK(h, hipFree(h->d_trace));
            h->d_trace = nullptr;
            HIPCHECK(h, hipMalloc((void **)&h->d_trace, sizeof(double) * 2 * n_iters));
            h->trace_cap = n_iters;
        }
        trace = h->d_trace - 2 * h->iter;  // the kernel indexes by the absolute iteration
    }
    if (!h->dev_loop_primed) launch_set_state<R>(P, nullptr, s);  // (see closed_loop_impl)
    h->dev_loop_primed = false;
    for (int i = 0; i < n_iters; ++i)
        if (int rc = rccl_iteration<R>(h, P, 1, trace, s)) return rc;
    h->last_eps = nullptr;
    h->last_philox = true;
    if (int rc = rccl_fetch(h, nullptr, nullptr, stats, s)) return rc;
    if (u0_trace) HIPCHECK(h, hipMemcpy(u0_trace, h->d_trace, sizeof(double) * 2 * n_iters, hipMemcpyDeviceToHost));
    h->dev_loop_primed = true;  // the last finalize made the next iteration's x0 call
    return MPPI_OK;
}

// ------------------------------------------------------------------------------------------
// Peer-to-peer exchange of the per-rank softmin record (include/mppi_hip.h, mppi_comm_*)
// ------------------------------------------------------------------------------------------
extern "C" int mppi_comm_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int mppi_comm_close(mppi_handle *h) {
    if (!h) return MPPI_ERR_BAD_ARG;
    hipSetDevice(h->cfg.device);
    hipDeviceSynchronize();
    for (void *p : h->x_opened) hipIpcCloseMemHandle(p);
    h->x_opened.clear();
    if (h->d_xpeers) hipFree(h->d_xpeers);
    if (h->d_xerr) hipFree(h->d_xerr);
    if (h->d_xok) hipFree(h->d_xok);
    if (h->xbuf) hipFree(h->xbuf);
    h->d_xpeers = nullptr;
    h->d_xerr = h->d_xok = nullptr;
    h->xbuf = nullptr;
    h->x_nranks = h->x_export_nranks = 0;
    rccl_release(h);
    return MPPI_OK;
}

extern "C" int mppi_comm_export(mppi_handle *h, int32_t nranks, void *handle_out) {
    if (!h || !handle_out) return MPPI_ERR_BAD_ARG;
    SINGLE_AGENT_ONLY(h, "mppi_comm_export");
    if (nranks < 2 || nranks > XCHG_MAX_RANKS)
        FAIL(h, MPPI_ERR_BAD_ARG, "mppi_comm_export: nranks must be 2..%d (got %d)", XCHG_MAX_RANKS, nranks);
    if (h->cfg.waypoint_mode == MPPI_WAYPOINT_SEQUENTIAL)
        FAIL(h, MPPI_ERR_UNSUPPORTED, "the exchange needs MPPI_WAYPOINT_FROZEN (no cross-sample waypoint state)");
    // the finalize kernel stages up to XCHG_LDS_RANKS rank records in LDS next to its merge workspace
    if (merge_lds_elems(h->cfg.T, h->cfg.filter_window, rsz(h)) * rsz(h) + sizeof(double) * XCHG_LDS_RANKS * xchg_rec_len(h->cfg.T) > 64 * 1024)
        FAIL(h, MPPI_ERR_UNSUPPORTED, "the peer-to-peer exchange stages the rank records in LDS: horizon T=%d is too long for it "
                                      "(use the split step with a collective)", h->cfg.T);
    if (h->xbuf) mppi_comm_close(h);
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    h->xbuf_bytes = 2 * xchg_slot_bytes(h->cfg.T, nranks);
    // fine-grained: stores of a peer GPU become visible to this GPU's loads without a cache writeback here
    HIPCHECK(h, hipExtMallocWithFlags((void **)&h->xbuf, h->xbuf_bytes, hipDeviceMallocFinegrained));
    HIPCHECK(h, hipMemset(h->xbuf, 0, h->xbuf_bytes));
    HIPCHECK(h, hipDeviceSynchronize());
    hipIpcMemHandle_t ipc;
    HIPCHECK(h, hipIpcGetMemHandle(&ipc, h->xbuf));
    memcpy(handle_out, &ipc, sizeof(ipc));
    h->x_export_nranks = nranks;
    return MPPI_OK;
}

extern "C" int mppi_comm_buffer(mppi_handle *h, void **device_ptr) {
    if (!h || !device_ptr) return MPPI_ERR_BAD_ARG;
    if (!h->xbuf) FAIL(h, MPPI_ERR_STATE, "mppi_comm_buffer before mppi_comm_export");
    *device_ptr = h->xbuf;
    return MPPI_OK;
}

extern "C" int mppi_comm_connect(mppi_handle *h, int32_t rank, int32_t nranks, const void *handles,
                                 void *const *local_ptrs) {
    if (!h || (!handles && !local_ptrs)) return MPPI_ERR_BAD_ARG;
    if (!h->xbuf || nranks != h->x_export_nranks)
        FAIL(h, MPPI_ERR_STATE, "mppi_comm_connect: call mppi_comm_export with the same nranks first");
    if (rank < 0 || rank >= nranks) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_comm_connect: rank %d of %d", rank, nranks);
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    std::vector<char *> peers(nranks, nullptr);
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) {
            peers[r] = h->xbuf;
        } else if (local_ptrs && local_ptrs[r]) {
            // a peer living in this process: its buffer must be device memory, and when it sits on ANOTHER GPU this
            // handle's device needs peer access to it (the IPC route below enables that lazily; a raw pointer does not)
            hipPointerAttribute_t at;
            memset(&at, 0, sizeof(at));
            if (hipPointerGetAttributes(&at, local_ptrs[r]) != hipSuccess || at.type == hipMemoryTypeHost ||
                at.type == hipMemoryTypeUnregistered) {
                (void)hipGetLastError();
                FAIL(h, MPPI_ERR_UNSUPPORTED, "mppi_comm_connect: local_ptrs[%d] is not device memory (pass the peer's mppi_comm_buffer)", r);
            }
            if (at.device != h->cfg.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, h->cfg.device, at.device) != hipSuccess || !can)
                    FAIL(h, MPPI_ERR_UNSUPPORTED, "mppi_comm_connect: device %d cannot access rank %d's buffer on device %d "
                                                  "(no peer access: use the collective carrier)", h->cfg.device, r, at.device);
                const hipError_t pe = hipDeviceEnablePeerAccess(at.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                    FAIL(h, MPPI_ERR_UNSUPPORTED, "mppi_comm_connect: hipDeviceEnablePeerAccess(%d) failed: %s", at.device,
                         hipGetErrorString(pe));
                (void)hipGetLastError();
            }
            peers[r] = (char *)local_ptrs[r];
        } else {
            if (!handles) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_comm_connect: no handle for rank %d", r);
            hipIpcMemHandle_t ipc;
            memcpy(&ipc, (const char *)handles + sizeof(ipc) * (size_t)r, sizeof(ipc));
            void *p = nullptr;
            HIPCHECK(h, hipIpcOpenMemHandle(&p, ipc, hipIpcMemLazyEnablePeerAccess));
            h->x_opened.push_back(p);
            peers[r] = (char *)p;
        }
    }
    if (h->d_xpeers) { hipFree(h->d_xpeers); h->d_xpeers = nullptr; }  // (connected before: the new wiring replaces it)
    if (h->d_xerr) { hipFree(h->d_xerr); h->d_xerr = nullptr; }
    if (h->d_xok) { hipFree(h->d_xok); h->d_xok = nullptr; }
    HIPCHECK(h, hipMalloc((void **)&h->d_xpeers, sizeof(char *) * nranks));
    HIPCHECK(h, hipMemcpy(h->d_xpeers, peers.data(), sizeof(char *) * nranks, hipMemcpyHostToDevice));
    HIPCHECK(h, hipMalloc((void **)&h->d_xerr, sizeof(int)));
    HIPCHECK(h, hipMalloc((void **)&h->d_xok, sizeof(int)));
    HIPCHECK(h, hipMemset(h->d_xerr, 0, sizeof(int)));
    if (const char *e = getenv("MPPI_EXCHANGE_TIMEOUT_MS")) {
        const long long ms = atoll(e);
        if (ms > 0) h->x_timeout = ms * 100000LL;  // the wall clock counts at 100 MHz
    }
    h->x_rank = rank;
    h->x_nranks = nranks;
    h->xseq = 0;
    return MPPI_OK;
}

extern "C" int mppi_comm_probe(mppi_handle *h, void *stream) {
    if (!h) return MPPI_ERR_BAD_ARG;
    if (h->x_nranks <= 1) FAIL(h, MPPI_ERR_STATE, "mppi_comm_probe before mppi_comm_connect");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    FinalizeParams F = make_finalize(h, nullptr, 0, 0);
    arm_exchange(h, F);
    launch_exchange_probe(F, h->d_xok, (hipStream_t)stream);
    int ok = 0;
    HIPCHECK(h, hipMemcpyAsync(&ok, h->d_xok, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHECK(h, hipStreamSynchronize((hipStream_t)stream));
    if (!ok) {
        HIPCHECK(h, hipMemset(h->d_xerr, 0, sizeof(int)));  // the caller may fall back and retry later
        FAIL(h, MPPI_ERR_COMM, "peer-to-peer exchange probe: a rank did not arrive within the timeout");
    }
    return MPPI_OK;
}

// GRAPH_SLOTS closed-loop iterations (rollout -> finalize, the device-resident plant, next x0 call) as ONE instantiated HIP
// graph, cached per handle and re-captured when a kernel argument changes.  Replayed, an iteration costs the host a
// 64th of a graph launch instead of two kernel launches (2.8 us each against the GPU's 4.4 us per kernel: a slower or
// shared host core paces the eager loop, 10-12 us per iteration on such boxes of the pool), and the GPU side runs the
// chain at 8.6-8.7 us per iteration on every box (mppi_time_rollout_launch measures exactly this).  Only iterations that
// cannot ask for another round: frozen waypoint index, or the sequential one resting at the end of the path.
// GRAPH_SLOTS closed-loop iterations (rollout -> finalize, the device-resident plant, next x0 call) as an instantiated HIP
// graph, cached per handle and re-captured when a kernel argument changes.  Replayed, an iteration costs the host a
// 64th of a graph launch instead of two kernel launches (2.8 us each against the GPU's 4.4 us per kernel: a slower or
// shared host core paces the eager loop, 10-12 us per iteration on such boxes of the pool), and the GPU side runs the
// chain at 8.7 us per iteration on every box (mppi_time_rollout_launch measures exactly this).  Two instances of the
// graph are launched in turn and an instance is launched again only when its previous run has finished: several queued
// launches of ONE instance take a slow path in the runtime (11.2 us per iteration with 19 of them queued, measured).
// Only iterations that cannot ask for another round: frozen waypoint index, or the sequential one resting at the end of
// the path.  OPT-IN (MPPI_GRAPH=1): on the pool's boxes the replayed loop holds 9.6-9.7 us per iteration where the eager
// one wanders between 9.7 and 11.2, but over whole episodes (restarts, the eager traversal, re-captures when an argument
// changes) it came out 0.1-0.4 us per iteration behind, so eager launches stay the default.  MPPI_GRAPH_SLOTS: iterations
// per graph (experiments).
static int graph_slots() {
    static const int n = getenv("MPPI_GRAPH_SLOTS") ? atoi(getenv("MPPI_GRAPH_SLOTS")) : 64;
    return n < 1 ? 1 : n;
}
#define GRAPH_SLOTS graph_slots()
template <typename R>
static bool ensure_graph(mppi_handle *h, const KParams<R> &P, const FinalizeParams &F) {
    std::vector<char> key(sizeof(P) + sizeof(F) + 2 * sizeof(int));
    memcpy(key.data(), &P, sizeof(P));
    memcpy(key.data() + sizeof(P), &F, sizeof(F));
    const int tail[2] = {h->rollout_repeats, (int)sizeof(R)};
    memcpy(key.data() + sizeof(P) + sizeof(F), tail, sizeof(tail));
    if (h->graph_exec[0] && key == h->graph_key) return true;
    static const bool verbose = getenv("MPPI_GRAPH_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "[mppi] capturing a graph of %d iterations\n", GRAPH_SLOTS);
    for (int i = 0; i < 2; ++i) {
        if (h->graph_exec[i]) hipGraphExecDestroy(h->graph_exec[i]);
        h->graph_exec[i] = nullptr;
    }
    hipError_t e = hipSuccess;
    if (!h->graph_stream) e = hipStreamCreateWithFlags(&h->graph_stream, hipStreamNonBlocking);
    if (e == hipSuccess && !h->graph_ev_in) e = hipEventCreateWithFlags(&h->graph_ev_in, hipEventDisableTiming);
    if (e == hipSuccess && !h->graph_ev_out) e = hipEventCreateWithFlags(&h->graph_ev_out, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i)
        if (!h->graph_done[i]) e = hipEventCreateWithFlags(&h->graph_done[i], hipEventDisableTiming);
    hipGraph_t graph = nullptr;
    const long long l0 = h->n_rollout_launches, f0 = h->n_finalize_launches;
    if (e == hipSuccess) e = hipStreamBeginCapture(h->graph_stream, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
        for (int j = 0; j < GRAPH_SLOTS; ++j) launch_slot<R>(h, P, F, h->graph_stream);
        e = hipStreamEndCapture(h->graph_stream, &graph);
    }
    h->n_rollout_launches = l0;  // (counted per replay)
    h->n_finalize_launches = f0;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipGraphInstantiate(&h->graph_exec[i], graph, nullptr, nullptr, 0);
    if (graph) hipGraphDestroy(graph);
    if (e != hipSuccess) {  // no graphs on this runtime: the eager loop serves
        (void)hipGetLastError();
        for (int i = 0; i < 2; ++i) {
            if (h->graph_exec[i]) hipGraphExecDestroy(h->graph_exec[i]);
            h->graph_exec[i] = nullptr;
        }
        h->graph_on = false;
        return false;
    }
    h->graph_key.swap(key);
    return true;
}

template <typename R>
static int closed_loop_impl(mppi_handle *h, int n_iters, double *u0_trace, mppi_stats *stats, hipStream_t s) {
    if (h->rccl_comm && h->x_nranks <= 1) return closed_loop_rccl<R>(h, n_iters, u0_trace, stats, s);
    KParams<R> P = make_params<R>(h, nullptr);
    FinalizeParams F = make_finalize(h, h->d_partials, h->n_part, 1);
    if (u0_trace) {
        if (h->trace_cap < n_iters) {
            if (h->d_trace) HIPCHECK(h, hipFree(h->d_trace));
            h->d_trace = nullptr;
            HIPCHECK(h, hipMalloc((void **)&h->d_trace, sizeof(double) * 2 * n_iters));
            h->trace_cap = n_iters;
        }
        F.u0_trace = h->d_trace - 2 * h->iter;  // the kernel indexes by the absolute iteration
    }
    const long long target = h->iter + n_iters;
    // x0 call for the state already on the device -- unless the finalize kernel of the previous closed-loop iteration
    // has made it (a second call would search from the index the first one left: with the frozen index that is c, not
    // prev_waypoints_idx, and the result would depend on how a run is cut into calls)
    if (!h->dev_loop_primed) launch_set_state<R>(P, nullptr, s);
    h->dev_loop_primed = false;
    long long done = h->iter;
    int guard = 0;
    const double t_call = now_s();
    // While the HYPK kernels are in use the host looks in on the waypoint index after 32, 64, 128 ... slots: at the end
    // of the path (the reference driver's run reaches it after some 23 of its 1000 iterations) the lean kernels take over
    long long batch = P.hyp ? 32 : (1LL << 62);
    int idx_now = h->idx_valid ? h->idx : -1;
    while (done < target) {
        const long long todo = target - done < batch ? target - done : batch;
        // The last slot of the batch writes its result straight into mapped host memory and publishes a sequence word
        // the host polls (as mppi_step does): no copy launch and no stream synchronisation at the end of the call
        // (14 -> 7 us of fixed cost per call).  Several agents per handle: one result per agent, copied as before.
        const bool poll = h->poll && h->B == 1;
        const double t_enq = now_s();
        // the bulk of a long batch from the cached graph (see ensure_graph), the rest -- and the slot that publishes the
        // result -- eagerly behind it, all on the graph's stream, which waits for and is waited for by the caller's
        const bool rests = h->cfg.waypoint_mode != MPPI_WAYPOINT_SEQUENTIAL || (idx_now >= 0 && idx_now >= h->n_ref - 1);
        hipStream_t ls = s;
        long long i = 0;
        if (h->graph_on && rests && !P.hyp && !u0_trace && h->x_nranks <= 1 && !timing_on(h) && todo > GRAPH_SLOTS &&
            ensure_graph<R>(h, P, F)) {
            ls = h->graph_stream;
            HIPCHECK(h, hipEventRecord(h->graph_ev_in, s));
            HIPCHECK(h, hipStreamWaitEvent(ls, h->graph_ev_in, 0));
            const long long reps = (todo - 1) / GRAPH_SLOTS;
            for (long long r = 0; r < reps; ++r) {
                const int inst = (int)(r & 1);
                if (r >= 2) HIPCHECK(h, hipEventSynchronize(h->graph_done[inst]));  // that instance's previous run is over
                HIPCHECK(h, hipGraphLaunch(h->graph_exec[inst], ls));
                HIPCHECK(h, hipEventRecord(h->graph_done[inst], ls));
            }
            i = reps * GRAPH_SLOTS;
            h->n_rollout_launches += i * h->rollout_repeats;
            h->n_finalize_launches += i;
        }
        for (; i < todo; ++i) {
            if (poll && i == todo - 1) {
                FinalizeParams Fl = F;
                Fl.res = h->res_mapped;
                Fl.seq = ++h->seq;
                launch_slot<R>(h, P, Fl, ls);
            } else {
                launch_slot<R>(h, P, F, ls);
            }
        }
        if (ls != s) {
            HIPCHECK(h, hipEventRecord(h->graph_ev_out, ls));
            HIPCHECK(h, hipStreamWaitEvent(s, h->graph_ev_out, 0));
        }
        HIPCHECK(h, hipGetLastError());
        h->t_enqueue_s += now_s() - t_enq;
        if (poll) {
            if (int rc = wait_result(h, h->seq, s)) return rc;
        } else {
            HIPCHECK(h, hipMemcpyAsync(h->h_res, h->d_res, (size_t)h->B * h->res_bytes, hipMemcpyDeviceToHost, s));
            HIPCHECK(h, hipStreamSynchronize(s));
            HIPCHECK(h, hipGetLastError());
        }
        for (int a = 1; a < h->B; ++a) {  // an agent at the end of its path stops the batch like the single agent does
            const StepResult *ra =
                reinterpret_cast<const StepResult *>(reinterpret_cast<const char *>(h->h_res) + (size_t)a * h->res_bytes);
            if (ra->status == STATUS_PATH_END) h->h_res->status = STATUS_PATH_END;
        }
        if (h->h_res->status == STATUS_PATH_END || h->h_res->status == STATUS_EXCHANGE_FAILED) break;
        done = h->h_res->iter;  // slots spent on speculation rounds did not complete an iteration
        idx_now = h->h_res->idx_after;
        if (P.hyp) {
            if (h->B == 1 && h->h_res->idx_after >= h->n_ref - 1) {
                P.hyp = F.hyp = 0;
                batch = 1LL << 62;
            } else {
                batch *= 2;
            }
        }
        if (++guard > h->cfg.K + 64) FAIL(h, MPPI_ERR_STATE, "closed loop did not make progress");
    }
    h->t_loop_s += now_s() - t_call;
    h->last_iter_us = 1e6 * (now_s() - t_call) / (n_iters > 0 ? n_iters : 1);
    h->last_eps = nullptr;
    h->last_philox = true;
    h->idx = h->h_res->idx_after;
    fill_stats(h, stats);
    if (h->h_res->status == STATUS_EXCHANGE_FAILED)
        FAIL(h, MPPI_ERR_COMM, "peer-to-peer exchange: a rank did not arrive within the timeout");
    if (h->h_res->status == STATUS_PATH_END)
        FAIL(h, MPPI_ERR_PATH_END, "[ERROR] Reached the end of the reference path.");
    if (u0_trace)
        HIPCHECK(h, hipMemcpy(u0_trace, h->d_trace, sizeof(double) * 2 * n_iters, hipMemcpyDeviceToHost));
    h->iter = h->h_res->iter;
    h->dev_loop_primed = true;  // the last finalize made the next iteration's x0 call
    return MPPI_OK;
}

extern "C" int mppi_run_closed_loop(mppi_handle *h, int32_t n_iters, double *u0_trace, mppi_stats *stats,
                                    void *stream) {
    int rc = check_ready(h, "mppi_run_closed_loop");
    if (rc) return rc;
    if (n_iters < 1) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_run_closed_loop: n_iters < 1");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    return h->f64 ? closed_loop_impl<double>(h, n_iters, u0_trace, stats, (hipStream_t)stream)
                  : closed_loop_impl<float>(h, n_iters, u0_trace, stats, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// Launch-to-launch duration of the rollout kernel on the GPU, with the host out of the loop (include/mppi_hip.h,
// mppi_time_rollout_launch): n_slots closed-loop iterations captured into a HIP graph twice -- as they are, and with the
// (idempotent) rollout kernel launched 1 + extra times per iteration -- and replayed between two events each; the
// difference per extra launch is what one more launch of that kernel costs the stream.  Eager launches measure the same
// on a host that keeps the queue full; on a slow or shared host core they measure the host.
// ------------------------------------------------------------------------------------------
template <typename R>
static int time_rollout_impl(mppi_handle *h, int n_slots, int extra, hipStream_t s, double *us_out) {
    KParams<R> P = make_params<R>(h, nullptr);
    FinalizeParams F = make_finalize(h, h->d_partials, h->n_part, 1);
    if (P.hyp || (h->cfg.waypoint_mode == MPPI_WAYPOINT_SEQUENTIAL && h->idx < h->n_ref - 1))
        FAIL(h, MPPI_ERR_STATE, "mppi_time_rollout_launch: the sequential waypoint index can still move (speculation rounds "
                                "cannot be replayed from a graph); call it once the index rests at the end of the path");
    if (timing_on(h))
        FAIL(h, MPPI_ERR_STATE, "mppi_time_rollout_launch: switch mppi_enable_timing off first (its events cannot be recorded "
                                "inside a stream capture)");
    if (!h->dev_loop_primed) launch_set_state<R>(P, nullptr, s);
    HIPCHECK(h, hipStreamSynchronize(s));
    struct Scope {  // the stream and the event pair are released on every exit
        hipStream_t gs = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Scope() {
            if (e0) hipEventDestroy(e0);
            if (e1) hipEventDestroy(e1);
            if (gs) hipStreamDestroy(gs);
        }
    } sc;
    HIPCHECK(h, hipStreamCreateWithFlags(&sc.gs, hipStreamNonBlocking));
    HIPCHECK(h, hipEventCreate(&sc.e0));
    HIPCHECK(h, hipEventCreate(&sc.e1));
    const hipStream_t gs = sc.gs;
    const hipEvent_t e0 = sc.e0, e1 = sc.e1;
    const int saved = h->rollout_repeats, reps = 4;
    const long long l0 = h->n_rollout_launches, f0 = h->n_finalize_launches;
    double ms[2] = {0.0, 0.0};
    int rc = MPPI_OK;
    for (int v = 0; v < 2 && rc == MPPI_OK; ++v) {
        h->rollout_repeats = v == 0 ? 1 : 1 + extra;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipError_t e = hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            for (int j = 0; j < n_slots; ++j) launch_slot<R>(h, P, F, gs);
            e = hipStreamEndCapture(gs, &graph);
        }
        if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e == hipSuccess) e = hipGraphLaunch(exec, gs);  // warm
        if (e == hipSuccess) e = hipStreamSynchronize(gs);
        const double w0 = now_s();
        if (e == hipSuccess) e = hipEventRecord(e0, gs);
        for (int r = 0; r < reps && e == hipSuccess; ++r) e = hipGraphLaunch(exec, gs);
        if (e == hipSuccess) e = hipEventRecord(e1, gs);
        if (e == hipSuccess) e = hipStreamSynchronize(gs);
        if (getenv("MPPI_GRAPH_VERBOSE")) fprintf(stderr, "[mppi] variant %d: wall %.2f us per iteration\n", v, 1e6 * (now_s() - w0) / (reps * n_slots));
        float t = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
        ms[v] = t;
        if (exec) hipGraphExecDestroy(exec);
        if (graph) hipGraphDestroy(graph);
        if (e != hipSuccess) {
            h->err = std::string("mppi_time_rollout_launch: ") + hipGetErrorString(e);
            rc = MPPI_ERR_HIP;
        }
    }
    h->rollout_repeats = saved;
    h->n_rollout_launches = l0;  // (a diagnostic: the bookkeeping of the caller's runs stays as it was)
    h->n_finalize_launches = f0;
    if (rc != MPPI_OK) return rc;
    // the replays advanced the closed loop like any other run: pick the state up where they left it
    HIPCHECK(h, hipMemcpy(h->h_res, h->d_res, h->res_bytes, hipMemcpyDeviceToHost));
    h->iter = h->h_res->iter;
    h->idx = h->h_res->idx_after;
    h->dev_loop_primed = true;
    h->last_eps = nullptr;
    h->last_philox = true;
    const double per = (double)reps * n_slots;
    us_out[0] = 1e3 * (ms[1] - ms[0]) / (per * extra);
    us_out[1] = 1e3 * ms[0] / per;
    return MPPI_OK;
}

extern "C" int mppi_time_rollout_launch(mppi_handle *h, int32_t n_slots, int32_t extra, void *stream, double *us_out2) {
    int rc = check_ready(h, "mppi_time_rollout_launch");
    if (rc) return rc;
    if (!us_out2 || n_slots < 1 || n_slots > 4096 || extra < 1 || extra > 16)
        FAIL(h, MPPI_ERR_BAD_ARG, "mppi_time_rollout_launch: n_slots 1..4096, extra 1..16");
    if (h->x_nranks > 1 || h->rccl_comm)
        FAIL(h, MPPI_ERR_UNSUPPORTED, "mppi_time_rollout_launch: not with an exchange between ranks (its sequence numbers cannot be replayed)");
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    return h->f64 ? time_rollout_impl<double>(h, n_slots, extra, (hipStream_t)stream, us_out2)
                  : time_rollout_impl<float>(h, n_slots, extra, (hipStream_t)stream, us_out2);
}

// ------------------------------------------------------------------------------------------
// Batched stage methods (include/mppi_hip.h, mppi_eval_*): host arrays in, host arrays out; the arithmetic is the
// device code of the rollout kernels in the handle's precision.
// ------------------------------------------------------------------------------------------
struct DevBuf {  // scratch device buffer released at scope exit
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};

static void eval_transition_mlp(mppi_handle *h, const KParams<float> &P, const void *x, const void *v, int n, void *out) {
    launch_eval_mlp(P, h->mlp, (const float *)x, (const float *)v, n, (float *)out, nullptr);
}
static void eval_transition_mlp(mppi_handle *, const KParams<double> &, const void *, const void *, int, void *) {}  // rejected at create

template <typename R>
static int eval_impl(mppi_handle *h, int what, const double *x, const double *v, int n, int32_t *prev_idx, int update,
                     double *out, int32_t *idx_out) {
    const KParams<R> P = make_params<R>(h, nullptr);
    const int nx = h->nx;
    const bool need_x = what != EVAL_CLAMP, need_v = what == EVAL_TRANSITION || what == EVAL_CLAMP;
    const bool need_idx = what == EVAL_COST_STAGE || what == EVAL_COST_TERMINAL || what < 0;  // what < 0: index only
    const int n_out = what == EVAL_TRANSITION ? nx : what == EVAL_CLAMP ? 2 : 1;
    DevBuf dx, dv, di, dp, dout;
    if (need_x) {
        HIPCHECK(h, dx.alloc(sizeof(R) * (size_t)n * nx));
        if (int rc = upload_real(h, dx.p, x, (size_t)n * nx)) return rc;
    }
    if (need_v) {
        HIPCHECK(h, dv.alloc(sizeof(R) * (size_t)n * 2));
        if (int rc = upload_real(h, dv.p, v, (size_t)n * 2)) return rc;
    }
    if (need_idx) {
        if (!prev_idx) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_eval: prev_idx is null");
        if (*prev_idx < 0 || *prev_idx >= h->n_ref) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_eval: waypoint index %d out of range", *prev_idx);
        HIPCHECK(h, di.alloc(sizeof(int) * (size_t)n));
        HIPCHECK(h, dp.alloc(sizeof(int)));
        launch_eval_index<R>(P, (const R *)dx.p, nx, n, *prev_idx, update ? 1 : 0, (int *)di.p, (int *)dp.p, nullptr);
    }
    if (what >= 0) {
        HIPCHECK(h, dout.alloc(sizeof(R) * (size_t)n * n_out));
        if (what == EVAL_TRANSITION && h->cfg.model == MPPI_MODEL_DIFFDRIVE_MLP)  // x + dt (f + MLP([x, v])), fp32 handles only
            eval_transition_mlp(h, P, dx.p, dv.p, n, dout.p);
        else
            launch_eval<R>(P, what, (const R *)dx.p, (const R *)dv.p, (const int *)di.p, n, (R *)dout.p, nullptr);
    }
    HIPCHECK(h, hipGetLastError());
    HIPCHECK(h, hipDeviceSynchronize());
    if (what >= 0)
        if (int rc = download_real(h, out, dout.p, (size_t)n * n_out)) return rc;
    if (need_idx) {
        if (idx_out) HIPCHECK(h, hipMemcpy(idx_out, di.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
        int pn = *prev_idx;
        HIPCHECK(h, hipMemcpy(&pn, dp.p, sizeof(int), hipMemcpyDeviceToHost));
        if (update) *prev_idx = pn;
    }
    return MPPI_OK;
}

static int eval_entry(mppi_handle *h, const char *who, int what, const double *x, const double *v, int n, int32_t *prev_idx,
                      int update, double *out, int32_t *idx_out) {
    int rc = check_ready(h, who);
    if (rc) return rc;
    if (n < 1 || (what >= 0 && !out)) FAIL(h, MPPI_ERR_BAD_ARG, "%s: bad argument", who);
    if ((what != EVAL_CLAMP && !x) || ((what == EVAL_TRANSITION || what == EVAL_CLAMP) && !v))
        FAIL(h, MPPI_ERR_BAD_ARG, "%s: null input", who);
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    return h->f64 ? eval_impl<double>(h, what, x, v, n, prev_idx, update, out, idx_out)
                  : eval_impl<float>(h, what, x, v, n, prev_idx, update, out, idx_out);
}

extern "C" int mppi_eval_state_transition(mppi_handle *h, const double *x, const double *v, int32_t n, double *x_next) {
    return eval_entry(h, "mppi_eval_state_transition", EVAL_TRANSITION, x, v, n, nullptr, 0, x_next, nullptr);
}
extern "C" int mppi_eval_clamp(mppi_handle *h, const double *v, int32_t n, double *out) {
    return eval_entry(h, "mppi_eval_clamp", EVAL_CLAMP, nullptr, v, n, nullptr, 0, out, nullptr);
}
extern "C" int mppi_eval_is_collided(mppi_handle *h, const double *x, int32_t n, double *out) {
    return eval_entry(h, "mppi_eval_is_collided", EVAL_COLLIDED, x, nullptr, n, nullptr, 0, out, nullptr);
}
extern "C" int mppi_eval_nearest_waypoint(mppi_handle *h, const double *x, int32_t n, int32_t *prev_idx, int32_t update_prev_idx,
                                          int32_t *idx_out) {
    if (h && !idx_out) FAIL(h, MPPI_ERR_BAD_ARG, "mppi_eval_nearest_waypoint: idx_out is null");
    return eval_entry(h, "mppi_eval_nearest_waypoint", -1, x, nullptr, n, prev_idx, update_prev_idx, nullptr, idx_out);
}
extern "C" int mppi_eval_cost(mppi_handle *h, int32_t terminal, const double *x, int32_t n, int32_t *prev_idx,
                              int32_t update_prev_idx, double *cost, int32_t *idx_out) {
    return eval_entry(h, "mppi_eval_cost", terminal ? EVAL_COST_TERMINAL : EVAL_COST_STAGE, x, nullptr, n, prev_idx,
                      update_prev_idx, cost, idx_out);
}

extern "C" int mppi_eval_moving_average(mppi_handle *h, const double *xx, double *out) {
    if (!h || !xx || !out) return MPPI_ERR_BAD_ARG;
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    const int T = h->cfg.T;
    DevBuf in, o;
    HIPCHECK(h, in.alloc(rsz(h) * 2 * T));
    HIPCHECK(h, o.alloc(rsz(h) * 2 * T));
    if (int rc = upload_real(h, in.p, xx, (size_t)2 * T)) return rc;
    if (h->f64) launch_eval_filter<double>((const double *)in.p, (double *)o.p, T, h->cfg.filter_window, h->cfg.filter_mode, nullptr);
    else launch_eval_filter<float>((const float *)in.p, (float *)o.p, T, h->cfg.filter_window, h->cfg.filter_mode, nullptr);
    HIPCHECK(h, hipGetLastError());
    HIPCHECK(h, hipDeviceSynchronize());
    return download_real(h, out, o.p, (size_t)2 * T);
}

extern "C" int mppi_eval_weights(mppi_handle *h, const double *S, int32_t n, double *w) {
    if (!h || !S || !w || n < 1) return MPPI_ERR_BAD_ARG;
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    DevBuf in, o;
    HIPCHECK(h, in.alloc(sizeof(double) * (size_t)n));
    HIPCHECK(h, o.alloc(sizeof(double) * (size_t)n));
    HIPCHECK(h, hipMemcpy(in.p, S, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    const mppi_config &c = h->cfg;
    const double beta = c.beta_mode == MPPI_BETA_INV_EXPLORATION ? 1.0 / c.param_exploration
                        : c.beta_mode == MPPI_BETA_INV_LAMBDA    ? 1.0 / c.param_lambda
                                                                 : c.param_lambda;
    launch_eval_weights((const double *)in.p, n, beta, (double *)o.p, nullptr);
    HIPCHECK(h, hipGetLastError());
    HIPCHECK(h, hipDeviceSynchronize());
    HIPCHECK(h, hipMemcpy(w, o.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return MPPI_OK;
}

extern "C" int mppi_get_counters(mppi_handle *h, int64_t *out3) {
    if (!h || !out3) return MPPI_ERR_BAD_ARG;
    out3[0] = h->iter;
    out3[1] = h->n_rollout_launches;
    out3[2] = h->n_finalize_launches;
    return MPPI_OK;
}

extern "C" int mppi_get_host_timing(const mppi_handle *h, double *out2) {
    if (!h || !out2) return MPPI_ERR_BAD_ARG;
    out2[0] = h->t_enqueue_s;
    out2[1] = h->t_loop_s;
    return MPPI_OK;
}

extern "C" int mppi_get_rollout_layout(const mppi_handle *h, int32_t *layout) {
    if (!h || !layout) return MPPI_ERR_BAD_ARG;
    *layout = (h->fused && h->cfg.model != MPPI_MODEL_DIFFDRIVE_MLP) ? h->layout : -1;
    return MPPI_OK;
}

extern "C" int mppi_get_rollout_kernel(const mppi_handle *h, char *buf, int32_t n) {
    if (!h || !buf || n < 1) return MPPI_ERR_BAD_ARG;
    const char *nm = h->cfg.model == MPPI_MODEL_DIFFDRIVE_MLP && h->mlp_set ? mlp_kernel_name(h->mlp) : h->rollout_kernel;
    snprintf(buf, (size_t)n, "%s", nm ? nm : "");
    return MPPI_OK;
}

extern "C" int mppi_enable_timing(mppi_handle *h, int32_t on) {
    if (!h) return MPPI_ERR_BAD_ARG;
    if (on && !h->timing) h->ev_used = 0;  // a new measurement window
    h->timing = on != 0;
    return MPPI_OK;
}

extern "C" int mppi_set_rollout_repeats(mppi_handle *h, int32_t n) {
    if (!h || n < 1 || n > 64) return MPPI_ERR_BAD_ARG;
    h->rollout_repeats = n;
    return MPPI_OK;
}

extern "C" int mppi_last_kernel_ms(mppi_handle *h, float *out4) {
    if (!h || !out4) return MPPI_ERR_BAD_ARG;
    HIPCHECK(h, hipSetDevice(h->cfg.device));
    HIPCHECK(h, hipDeviceSynchronize());  // every recorded event has completed
    collect_timing(h);
    memcpy(out4, h->last_ms, sizeof(h->last_ms));
    return MPPI_OK;
}
