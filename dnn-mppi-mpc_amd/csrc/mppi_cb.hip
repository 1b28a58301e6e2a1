// `pytorch_mppi`-style MPPI with BUILT-IN dynamics / running-cost models: what the reference's
// test/test_mppi.py, test/test_mppi_diff.py, test/test_mppi_diff_dyna.py and train/bullet_mppi_differential_drive.py
// hand to `pytorch_mppi.MPPI` as Python callbacks, as selectable device functions, so that those callers need neither
// the library nor a callback per step.
//
//   dynamics      UNICYCLE     x' = x + v cos(th) dt, y' = y + v sin(th) dt, th' = th + w dt   (test/test_mppi.py:12-26)
//                 SKID_STEER   5 states [x, y, th, v, w], 4 wheel forces                       (test/test_mppi_diff_dyna.py:13-40)
//   running cost  (s - goal)^T diag(Q) (s - goal) + u^T diag(R) u + obstacle_weight * sum_m o_m(d_m)
//                 INVERSE      o = 1 / (d + 1e-6) inside the safety distance, 0 outside         (test/test_mppi.py:40-50)
//                 EXPONENTIAL  o = exp(-(d - safety)), obstacles moving: p_m(t) = p_m + vel_m t  (test/test_mppi_diff.py:25-52)
//
// The loop around them is the published information-theoretic MPPI (Williams et al. 2017) the way `pytorch_mppi`
// arranges it: shift U, sample noise, clamp the perturbed action and recompute the noise from it, roll out, add
// lambda * U^T Sigma^-1 noise, softmin weights with rate 1/lambda, U += sum_k w_k noise_k, return U[0].
// `pytorch_mppi` itself is absent from the build container and un-pinned by the reference (no version anywhere): the
// loop is PARITY UNPINNED; tests check it against oracle/mppi_cb_oracle.py, a NumPy restatement of the same algorithm.
//
// Small problems (K ~ 1000, T ~ 25): one THREAD per sample, serial in t -- nothing here is tuned; the tuned path is
// mppi_kernels.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/mppi_hip.h"
#include "philox.h"

namespace {

constexpr int CB_NX = 5, CB_NU = 4, CB_OBS = 16;

struct CbParams {
    int K, T, nx, nu, dynamics, obstacle_kind, n_obs, sample_null_action;
    float dt, lambda_, safety, obstacle_weight;
    float goal[CB_NX], q[CB_NX], r[CB_NU], u_min[CB_NU], u_max[CB_NU];
    float sinv[CB_NU * CB_NU], chol[CB_NU * CB_NU];  // Sigma^-1 and the lower Cholesky factor of Sigma (row-major)
    float obs[CB_OBS][4];                            // x, y, vx, vy
    float skid[5];                                   // m, I, r, L, damping (test/test_mppi_diff_dyna.py:15-19, 29-30)
    unsigned seed_lo, seed_hi;
};

__device__ __forceinline__ void cb_dynamics(const CbParams &P, float *s, const float *u) {
    if (P.dynamics == MPPI_CB_DYN_UNICYCLE) {  // test/test_mppi.py:12-26
        const float th = s[2];
        s[0] += u[0] * cosf(th) * P.dt;
        s[1] += u[0] * sinf(th) * P.dt;
        s[2] += u[1] * P.dt;
    } else {  // test/test_mppi_diff_dyna.py:13-40: F_fr, F_fl, F_rr, F_rl
        const float m = P.skid[0], I = P.skid[1], r = P.skid[2], L = P.skid[3], damp = P.skid[4];
        const float th = s[2], v = s[3], om = s[4];
        const float dv = (r / (4.f * m)) * (u[0] + u[1] + u[2] + u[3]) - damp * v;
        const float dom = (r / (L * I)) * ((u[0] + u[2]) - (u[1] + u[3])) / 2.f - damp * om;
        s[0] += v * cosf(th) * P.dt;
        s[1] += v * sinf(th) * P.dt;
        s[2] += om * P.dt;
        s[3] += dv * P.dt;
        s[4] += dom * P.dt;
    }
}

__device__ __forceinline__ float cb_running_cost(const CbParams &P, const float *s, const float *u, int t) {
    float c = 0.f;
    for (int i = 0; i < P.nx; ++i) {
        const float e = s[i] - P.goal[i];
        c += P.q[i] * e * e;
    }
    for (int i = 0; i < P.nu; ++i) c += P.r[i] * u[i] * u[i];
    float oc = 0.f;
    for (int m = 0; m < P.n_obs; ++m) {
        const float ox = P.obs[m][0] + P.obs[m][2] * (float)t, oy = P.obs[m][1] + P.obs[m][3] * (float)t;
        const float d = sqrtf((s[0] - ox) * (s[0] - ox) + (s[1] - oy) * (s[1] - oy));
        if (P.obstacle_kind == MPPI_CB_OBS_INVERSE) oc += d < P.safety ? 1.f / (d + 1e-6f) : 0.f;  // test_mppi.py:46-48
        else oc += expf(-(d - P.safety));                                                          // test_mppi_diff.py:42-44
    }
    return c + P.obstacle_weight * oc;
}

// noise[k, t, :] ~ N(0, Sigma): injected tensor, or Philox4x32-10 keyed by (k, t, iteration) -> two Box-Muller pairs
__device__ __forceinline__ void cb_noise(const CbParams &P, const float *eps, unsigned iter, int k, int t, float *n) {
    if (eps) {
        for (int i = 0; i < P.nu; ++i) n[i] = eps[((size_t)k * P.T + t) * P.nu + i];
        return;
    }
    unsigned r[4];
    px::philox4x32_10((unsigned)k, (unsigned)t, iter, 0x6362u, P.seed_lo, P.seed_hi, r);
    const float id[3] = {1.f, 0.f, 1.f};
    float z[4];
    px::box_muller(r[0], r[1], id, z[0], z[1]);
    px::box_muller(r[2], r[3], id, z[2], z[3]);
    for (int i = 0; i < P.nu; ++i) {
        float a = 0.f;
        for (int j = 0; j <= i; ++j) a += P.chol[i * CB_NU + j] * z[j];
        n[i] = a;
    }
}

// the action sample k applies at step t and the noise that stands for it after clamping
__device__ __forceinline__ void cb_action(const CbParams &P, const float *U, const float *eps, unsigned iter, int k, int t,
                                          float *act, float *noise) {
    float n[CB_NU];
    cb_noise(P, eps, iter, k, t, n);
    for (int i = 0; i < P.nu; ++i) {
        float a = U[t * P.nu + i] + n[i];
        if (P.sample_null_action && k == P.K - 1) a = 0.f;
        a = fminf(fmaxf(a, P.u_min[i]), P.u_max[i]);
        act[i] = a;
        noise[i] = a - U[t * P.nu + i];
    }
}

__global__ void k_cb_rollout(const CbParams P, const float *__restrict__ U, const float *__restrict__ state,
                             const float *__restrict__ eps, unsigned iter, float *__restrict__ cost_total) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P.K) return;
    float s[CB_NX];
    for (int i = 0; i < P.nx; ++i) s[i] = state[i];
    float cost = 0.f, pert = 0.f;
    for (int t = 0; t < P.T; ++t) {
        float act[CB_NU], noise[CB_NU];
        cb_action(P, U, eps, iter, k, t, act, noise);
        cb_dynamics(P, s, act);
        cost += cb_running_cost(P, s, act, t);
        for (int i = 0; i < P.nu; ++i) {  // lambda * U^T Sigma^-1 noise
            float a = 0.f;
            for (int j = 0; j < P.nu; ++j) a += noise[j] * P.sinv[j * CB_NU + i];
            pert += U[t * P.nu + i] * P.lambda_ * a;
        }
    }
    cost_total[k] = cost + pert;
}

// softmin weights with rate 1/lambda, U += sum_k w_k noise_k; one workgroup
__global__ __launch_bounds__(256) void k_cb_update(const CbParams P, float *__restrict__ U, const float *__restrict__ eps,
                                                   unsigned iter, const float *__restrict__ cost_total,
                                                   float *__restrict__ omega, float *__restrict__ action_out) {
    __shared__ float sh[256];
    __shared__ float sh_unew[256];
    const int tid = threadIdx.x;
    float m = INFINITY;
    for (int k = tid; k < P.K; k += 256) m = fminf(m, cost_total[k]);
    sh[tid] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = fminf(sh[tid], sh[tid + s]);
        __syncthreads();
    }
    const float beta = sh[0];
    __syncthreads();
    float a = 0.f;
    for (int k = tid; k < P.K; k += 256) a += expf(-(cost_total[k] - beta) / P.lambda_);
    sh[tid] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const float eta = sh[0];
    for (int k = tid; k < P.K; k += 256) omega[k] = expf(-(cost_total[k] - beta) / P.lambda_) / eta;
    __syncthreads();
    for (int e = tid; e < P.T * P.nu; e += 256) {  // (T * nu <= 256, checked at create)
        const int t = e / P.nu, i = e % P.nu;
        float acc = 0.f;
        for (int k = 0; k < P.K; ++k) {
            float act[CB_NU], noise[CB_NU];
            cb_action(P, U, eps, iter, k, t, act, noise);
            acc += omega[k] * noise[i];
        }
        sh_unew[e] = U[e] + acc;
    }
    __syncthreads();
    for (int e = tid; e < P.T * P.nu; e += 256) U[e] = sh_unew[e];
    if (tid < P.nu) action_out[tid] = sh_unew[tid];
}

__global__ void k_cb_shift(const CbParams P, float *__restrict__ U, const float *__restrict__ u_init) {  // roll(-1), last = u_init
    __shared__ float sh[256];
    const int n = P.T * P.nu, e = threadIdx.x;
    if (e < n) sh[e] = e + P.nu < n ? U[e + P.nu] : u_init[e - (n - P.nu)];
    __syncthreads();
    if (e < n) U[e] = sh[e];
}

__global__ void k_cb_eval(const CbParams P, int what, const float *__restrict__ s_in, const float *__restrict__ u_in, int t,
                          int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s[CB_NX], u[CB_NU];
    for (int q = 0; q < P.nx; ++q) s[q] = s_in[(size_t)i * P.nx + q];
    for (int q = 0; q < P.nu; ++q) u[q] = u_in[(size_t)i * P.nu + q];
    if (what == 0) {
        cb_dynamics(P, s, u);
        for (int q = 0; q < P.nx; ++q) out[(size_t)i * P.nx + q] = s[q];
    } else {
        out[i] = cb_running_cost(P, s, u, t);
    }
}

// the trajectory the nominal sequence U drives from `state` (what the reference's get_trajectories rolls out)
__global__ void k_cb_nominal(const CbParams P, const float *__restrict__ U, const float *__restrict__ state, float *__restrict__ traj) {
    if (threadIdx.x != 0) return;
    float s[CB_NX];
    for (int i = 0; i < P.nx; ++i) s[i] = state[i];
    for (int t = 0; t < P.T; ++t) {
        cb_dynamics(P, s, U + t * P.nu);
        for (int i = 0; i < P.nx; ++i) traj[t * P.nx + i] = s[i];
    }
}

}  // namespace

struct mppi_cb_handle {
    mppi_cb_config cfg;
    CbParams P;
    float *d_U = nullptr, *d_state = nullptr, *d_cost = nullptr, *d_omega = nullptr, *d_action = nullptr, *d_uinit = nullptr;
    long long iter = 0;
    std::string err;
};

static thread_local std::string g_cb_error;
#define CB_FAIL(h, code, ...)                                   \
    do {                                                        \
        char _b[400];                                           \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                  \
        if (h) (h)->err = _b; else g_cb_error = _b;             \
        return (code);                                          \
    } while (0)
#define CB_HIP(h, call)                                                                              \
    do {                                                                                             \
        hipError_t _e = (call);                                                                      \
        if (_e != hipSuccess) CB_FAIL(h, MPPI_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); \
    } while (0)

extern "C" const char *mppi_cb_last_error(const mppi_cb_handle *h) { return h ? h->err.c_str() : g_cb_error.c_str(); }

extern "C" int mppi_cb_create(const mppi_cb_config *c, mppi_cb_handle **out) {
    mppi_cb_handle *none = nullptr;
    if (!c || !out) CB_FAIL(none, MPPI_ERR_BAD_ARG, "mppi_cb_create: null argument");
    if (c->struct_size != (int32_t)sizeof(mppi_cb_config)) CB_FAIL(none, MPPI_ERR_BAD_ARG, "mppi_cb_create: struct_size mismatch");
    const int nx = c->dynamics == MPPI_CB_DYN_UNICYCLE ? 3 : 5, nu = c->dynamics == MPPI_CB_DYN_UNICYCLE ? 2 : 4;
    if (c->dynamics != MPPI_CB_DYN_UNICYCLE && c->dynamics != MPPI_CB_DYN_SKID_STEER)
        CB_FAIL(none, MPPI_ERR_BAD_ARG, "mppi_cb_create: unknown dynamics %d", c->dynamics);
    if (c->obstacle_kind != MPPI_CB_OBS_INVERSE && c->obstacle_kind != MPPI_CB_OBS_EXPONENTIAL)
        CB_FAIL(none, MPPI_ERR_BAD_ARG, "mppi_cb_create: unknown obstacle cost %d", c->obstacle_kind);
    if (c->K < 1 || c->T < 1 || c->T * nu > 256 || c->n_obs < 0 || c->n_obs > CB_OBS || !(c->lambda_ > 0) || !(c->dt > 0))
        CB_FAIL(none, MPPI_ERR_SHAPE, "mppi_cb_create: K=%d T=%d (T*nu <= 256) n_obs=%d (<= %d) lambda=%g dt=%g", c->K, c->T, c->n_obs,
                CB_OBS, c->lambda_, c->dt);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || c->device < 0 || c->device >= ndev)
        CB_FAIL(none, MPPI_ERR_NO_DEVICE, "mppi_cb_create: no such HIP device %d", c->device);
    // Cholesky factor and inverse of the nu x nu noise covariance (f64 on the host)
    double L[16] = {0}, Li[16] = {0}, Sinv[16] = {0};
    for (int i = 0; i < nu; ++i)
        for (int j = 0; j <= i; ++j) {
            double a = c->noise_sigma[i * 4 + j];
            for (int k = 0; k < j; ++k) a -= L[i * 4 + k] * L[j * 4 + k];
            if (i == j) {
                if (!(a > 0)) CB_FAIL(none, MPPI_ERR_BAD_ARG, "noise_sigma must be symmetric positive definite");
                L[i * 4 + i] = sqrt(a);
            } else {
                L[i * 4 + j] = a / L[j * 4 + j];
            }
        }
    for (int i = 0; i < nu; ++i) {  // L^-1 by forward substitution
        Li[i * 4 + i] = 1.0 / L[i * 4 + i];
        for (int j = 0; j < i; ++j) {
            double a = 0;
            for (int k = j; k < i; ++k) a -= L[i * 4 + k] * Li[k * 4 + j];
            Li[i * 4 + j] = a / L[i * 4 + i];
        }
    }
    for (int i = 0; i < nu; ++i)
        for (int j = 0; j < nu; ++j) {
            double a = 0;
            for (int k = 0; k < nu; ++k) a += Li[k * 4 + i] * Li[k * 4 + j];
            Sinv[i * 4 + j] = a;
        }
    mppi_cb_handle *h = new mppi_cb_handle();
    h->cfg = *c;
    CbParams &P = h->P;
    memset(&P, 0, sizeof(P));
    P.K = c->K; P.T = c->T; P.nx = nx; P.nu = nu; P.dynamics = c->dynamics; P.obstacle_kind = c->obstacle_kind;
    P.n_obs = c->n_obs; P.sample_null_action = c->sample_null_action;
    P.dt = (float)c->dt; P.lambda_ = (float)c->lambda_; P.safety = (float)c->safety_distance; P.obstacle_weight = (float)c->obstacle_weight;
    for (int i = 0; i < nx; ++i) { P.goal[i] = (float)c->goal[i]; P.q[i] = (float)c->q_diag[i]; }
    for (int i = 0; i < nu; ++i) { P.r[i] = (float)c->r_diag[i]; P.u_min[i] = (float)c->u_min[i]; P.u_max[i] = (float)c->u_max[i]; }
    for (int i = 0; i < 16; ++i) { P.sinv[i] = (float)Sinv[i]; P.chol[i] = (float)L[i]; }
    for (int m = 0; m < c->n_obs; ++m)
        for (int q = 0; q < 4; ++q) P.obs[m][q] = (float)c->obstacles[m * 4 + q];
    for (int i = 0; i < 5; ++i) P.skid[i] = (float)c->skid_params[i];
    P.seed_lo = (unsigned)(c->seed & 0xffffffffu);
    P.seed_hi = (unsigned)(c->seed >> 32);
    auto fail = [&](const char *what, hipError_t e) {
        g_cb_error = std::string(what) + " failed: " + hipGetErrorString(e);
        mppi_cb_destroy(h);
        return (int)MPPI_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return fail("hipSetDevice", e);
    if ((e = hipMalloc((void **)&h->d_U, sizeof(float) * c->T * nu)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc((void **)&h->d_state, sizeof(float) * CB_NX)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc((void **)&h->d_cost, sizeof(float) * c->K)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc((void **)&h->d_omega, sizeof(float) * c->K)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc((void **)&h->d_action, sizeof(float) * CB_NU)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc((void **)&h->d_uinit, sizeof(float) * CB_NU)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMemset(h->d_U, 0, sizeof(float) * c->T * nu)) != hipSuccess) return fail("hipMemset", e);
    float ui[CB_NU];
    for (int i = 0; i < CB_NU; ++i) ui[i] = (float)c->u_init[i];
    if ((e = hipMemcpy(h->d_uinit, ui, sizeof(ui), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
    *out = h;
    return MPPI_OK;
}

extern "C" int mppi_cb_destroy(mppi_cb_handle *h) {
    if (!h) return MPPI_OK;
    hipSetDevice(h->cfg.device);
    for (float *p : {h->d_U, h->d_state, h->d_cost, h->d_omega, h->d_action, h->d_uinit})
        if (p) hipFree(p);
    delete h;
    return MPPI_OK;
}

static int cb_io(mppi_cb_handle *h, float *dev, double *host_out, const double *host_in, size_t n) {
    std::vector<float> tmp(n);
    if (host_in) {
        for (size_t i = 0; i < n; ++i) tmp[i] = (float)host_in[i];
        CB_HIP(h, hipMemcpy(dev, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice));
    } else {
        CB_HIP(h, hipMemcpy(tmp.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) host_out[i] = (double)tmp[i];
    }
    return MPPI_OK;
}

extern "C" int mppi_cb_set_nominal(mppi_cb_handle *h, const double *U) {
    if (!h || !U) return MPPI_ERR_BAD_ARG;
    CB_HIP(h, hipSetDevice(h->cfg.device));
    CB_HIP(h, hipDeviceSynchronize());
    return cb_io(h, h->d_U, nullptr, U, (size_t)h->P.T * h->P.nu);
}
extern "C" int mppi_cb_get_nominal(mppi_cb_handle *h, double *U) {
    if (!h || !U) return MPPI_ERR_BAD_ARG;
    CB_HIP(h, hipSetDevice(h->cfg.device));
    CB_HIP(h, hipDeviceSynchronize());
    return cb_io(h, h->d_U, U, nullptr, (size_t)h->P.T * h->P.nu);
}
extern "C" int mppi_cb_get_costs(mppi_cb_handle *h, double *cost_total, double *omega) {
    if (!h) return MPPI_ERR_BAD_ARG;
    CB_HIP(h, hipSetDevice(h->cfg.device));
    CB_HIP(h, hipDeviceSynchronize());
    if (cost_total)
        if (int rc = cb_io(h, h->d_cost, cost_total, nullptr, (size_t)h->P.K)) return rc;
    if (omega)
        if (int rc = cb_io(h, h->d_omega, omega, nullptr, (size_t)h->P.K)) return rc;
    return MPPI_OK;
}

extern "C" int mppi_cb_command(mppi_cb_handle *h, const double *state, const float *eps, int32_t shift_nominal_trajectory,
                               double *action_out, void *stream) {
    if (!h || !state || !action_out) return MPPI_ERR_BAD_ARG;
    CB_HIP(h, hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    float st[CB_NX] = {0, 0, 0, 0, 0};
    for (int i = 0; i < h->P.nx; ++i) st[i] = (float)state[i];
    CB_HIP(h, hipMemcpyAsync(h->d_state, st, sizeof(st), hipMemcpyHostToDevice, s));
    if (shift_nominal_trajectory) hipLaunchKernelGGL(k_cb_shift, dim3(1), dim3(256), 0, s, h->P, h->d_U, h->d_uinit);
    hipLaunchKernelGGL(k_cb_rollout, dim3((h->P.K + 255) / 256), dim3(256), 0, s, h->P, h->d_U, h->d_state, eps,
                       (unsigned)h->iter, h->d_cost);
    hipLaunchKernelGGL(k_cb_update, dim3(1), dim3(256), 0, s, h->P, h->d_U, eps, (unsigned)h->iter, h->d_cost, h->d_omega,
                       h->d_action);
    CB_HIP(h, hipGetLastError());
    float act[CB_NU];
    CB_HIP(h, hipMemcpyAsync(act, h->d_action, sizeof(act), hipMemcpyDeviceToHost, s));
    CB_HIP(h, hipStreamSynchronize(s));
    for (int i = 0; i < h->P.nu; ++i) action_out[i] = (double)act[i];
    ++h->iter;
    return MPPI_OK;
}

extern "C" int mppi_cb_eval(mppi_cb_handle *h, int32_t what, const double *states, const double *actions, int32_t t, int32_t n,
                            double *out) {
    if (!h || !states || !actions || !out || n < 1 || (what != 0 && what != 1)) return MPPI_ERR_BAD_ARG;
    CB_HIP(h, hipSetDevice(h->cfg.device));
    const int nx = h->P.nx, nu = h->P.nu, n_out = what == 0 ? nx : 1;
    struct Scratch {  // released on every exit, the early ones of CB_HIP included
        float *p = nullptr;
        ~Scratch() { if (p) hipFree(p); }
    } s_ds, s_du, s_dout;
    CB_HIP(h, hipMalloc((void **)&s_ds.p, sizeof(float) * (size_t)n * nx));
    CB_HIP(h, hipMalloc((void **)&s_du.p, sizeof(float) * (size_t)n * nu));
    CB_HIP(h, hipMalloc((void **)&s_dout.p, sizeof(float) * (size_t)n * n_out));
    float *ds = s_ds.p, *du = s_du.p, *dout = s_dout.p;
    int rc = cb_io(h, ds, nullptr, states, (size_t)n * nx);
    if (!rc) rc = cb_io(h, du, nullptr, actions, (size_t)n * nu);
    if (!rc) {
        hipLaunchKernelGGL(k_cb_eval, dim3((n + 255) / 256), dim3(256), 0, nullptr, h->P, what, ds, du, t, n, dout);
        if (hipDeviceSynchronize() != hipSuccess) rc = MPPI_ERR_HIP;
    }
    if (!rc) rc = cb_io(h, dout, out, nullptr, (size_t)n * n_out);
    return rc;
}

extern "C" int mppi_cb_nominal_trajectory(mppi_cb_handle *h, const double *state, double *traj) {
    if (!h || !state || !traj) return MPPI_ERR_BAD_ARG;
    CB_HIP(h, hipSetDevice(h->cfg.device));
    float st[CB_NX] = {0, 0, 0, 0, 0};
    for (int i = 0; i < h->P.nx; ++i) st[i] = (float)state[i];
    struct Scratch {
        float *p = nullptr;
        ~Scratch() { if (p) hipFree(p); }
    } s_dt;
    CB_HIP(h, hipMalloc((void **)&s_dt.p, sizeof(float) * (size_t)h->P.T * h->P.nx));
    float *dt = s_dt.p;
    CB_HIP(h, hipMemcpy(h->d_state, st, sizeof(st), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cb_nominal, dim3(1), dim3(64), 0, nullptr, h->P, h->d_U, h->d_state, dt);
    int rc = hipDeviceSynchronize() == hipSuccess ? MPPI_OK : MPPI_ERR_HIP;
    if (!rc) rc = cb_io(h, dt, traj, nullptr, (size_t)h->P.T * h->P.nx);
    return rc;
}
