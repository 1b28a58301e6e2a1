// Device helpers shared by the analytic (mppi_kernels.hip) and learned-dynamics (mppi_mlp.hip) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>

#include "mathfn.h"
#include "mppi_kernels.h"
#include "philox.h"
#include "wave_ops.h"

namespace mppi {

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------

template <typename R> __device__ __forceinline__ R dist2(const R *__restrict__ ref, int i, R x, R y) {
    const R dx = x - ref[4 * i], dy = y - ref[4 * i + 1];
    return dx * dx + dy * dy;
}

// first-minimum argmin over ref[c .. c+wlen) for this lane's (x, y); c, wlen wave-uniform
// (`get_nearest_waypoint` mppi_race_car.py:157-174, `_get_nearest_waypoint` mppi_differential_drive.py:201-220)
template <typename R>
__device__ __forceinline__ int nearest_in_window(const R *__restrict__ ref, int c, int wlen, R x, R y) {
    R best = dist2(ref, c, x, y);
    int bj = 0;
#pragma unroll 4
    for (int j = 1; j < wlen; ++j) {
        const R d = dist2(ref, c + j, x, y);
        if (d < best) { best = d; bj = j; }
    }
    return c + bj;
}

// ---- the same search with the window staged in LDS ------------------------------------------------------
// The frozen-index modes search the SAME window ref[c .. c+wlen) for every step of every sample of a
// workgroup (mppi_race_car.py:157-174: up to 200 candidates per call), so the window is copied to LDS once per
// workgroup as pairs {x_2q, x_2q+1, y_2q, y_2q+1}: one broadcast 16/32-byte LDS read serves two candidates and the
// distance arithmetic packs (v_pk_add / v_pk_mul / v_pk_fma), instead of a scalar load + address arithmetic
// per candidate.  An odd window is padded with a far-away point that can never win.
constexpr int WINDOW_LDS_MAX = 256;  // candidates

template <typename R> struct alignas(16) RefPair { R x0, x1, y0, y1; };

template <typename R>
__device__ __forceinline__ void stage_window(RefPair<R> *sh, const R *__restrict__ ref, int c, int wlen, int tid,
                                             int nthreads) {
    const int npairs = (wlen + 1) >> 1;
    for (int q = tid; q < npairs; q += nthreads) {
        const int j0 = c + 2 * q, j1 = j0 + 1;
        RefPair<R> r;
        r.x0 = ref[4 * j0];
        r.y0 = ref[4 * j0 + 1];
        const bool has1 = 2 * q + 1 < wlen;
        r.x1 = has1 ? ref[4 * j1] : R(1e30);
        r.y1 = has1 ? ref[4 * j1 + 1] : R(1e30);
        sh[q] = r;
    }
}

template <typename R>
__device__ __forceinline__ int nearest_in_window_lds(const RefPair<R> *sh, int c, int wlen, R x, R y) {
    const int npairs = (wlen + 1) >> 1;
    R best = R(INFINITY);
    int bj = 0;
#pragma unroll 4
    for (int q = 0; q < npairs; ++q) {
        const RefPair<R> r = sh[q];
        const R dx0 = x - r.x0, dx1 = x - r.x1, dy0 = y - r.y0, dy1 = y - r.y1;
        const R d0 = dx0 * dx0 + dy0 * dy0, d1 = dx1 * dx1 + dy1 * dy1;
        if (d0 < best) { best = d0; bj = 2 * q; }
        if (d1 < best) { best = d1; bj = 2 * q + 1; }
    }
    return c + bj;
}

// Few active steps in this chunk (the tail of a horizon that is not a multiple of 64): `split` lanes share
// one step's candidates (interleaved pairs), then combine with first-minimum semantics.  `split` is a power of
// two <= 16 and n_act * split <= 64.  Returns, on lane t (< n_act), the index for step t.
template <typename R>
__device__ __forceinline__ int nearest_in_window_split(const RefPair<R> *sh, int c, int wlen, R x, R y, int split,
                                                       int lane) {
    const int lg = 31 - __clz(split);
    const int src = lane >> lg, part = lane & (split - 1);
    const R xs = __shfl(x, src), ys = __shfl(y, src);
    const int npairs = (wlen + 1) >> 1;
    R best = R(INFINITY);
    int bj = 0x7fffffff;
    for (int q = part; q < npairs; q += split) {
        const RefPair<R> r = sh[q];
        const R dx0 = xs - r.x0, dx1 = xs - r.x1, dy0 = ys - r.y0, dy1 = ys - r.y1;
        const R d0 = dx0 * dx0 + dy0 * dy0, d1 = dx1 * dx1 + dy1 * dy1;
        if (d0 < best) { best = d0; bj = 2 * q; }
        if (d1 < best) { best = d1; bj = 2 * q + 1; }
    }
    for (int m = 1; m < split; m <<= 1) {  // (smallest d, then smallest j) over the `split` lanes of a step
        const R od = __shfl_xor(best, m);
        const int oj = __shfl_xor(bj, m);
        if (od < best || (od == best && oj < bj)) { best = od; bj = oj; }
    }
    return c + __shfl(bj, lane << lg);  // lane t reads the result of its group's first lane
}

// Same search for ONE wave-uniform position with the candidates spread over the lanes.
template <typename R>
__device__ __forceinline__ int nearest_uniform(const R *__restrict__ ref, int c, int wlen, R x, R y, int lane) {
    R best = R(INFINITY);
    int bj = INT_MAX;
    for (int j = lane; j < wlen; j += 64) {
        const R d = dist2(ref, c + j, x, y);
        if (d < best) { best = d; bj = j; }
    }
    wv::argmin_first(best, bj);
    return c + bj;
}

// the noise tensor [K][T][2] of (iteration, agent): the caller's tensor of this call, or the slot of a noise ring
// (mppi_set_noise_ring: `_calc_epsilon` materialised for a closed loop; eps_slots is a power of two)
template <typename R> __device__ __forceinline__ const float *eps_tensor(const KParams<R> &P, unsigned iter, int agent) {
    size_t tensor = (size_t)agent;
    if (P.eps_slots > 0) tensor += (size_t)(iter & (unsigned)(P.eps_slots - 1)) * (size_t)(P.n_agents > 1 ? P.n_agents : 1);
    return P.eps + tensor * ((size_t)P.K * (size_t)P.T * 2);
}

template <typename R> __device__ __forceinline__ int window_len(int window, int n_ref, int c) {
    const int rem = n_ref - c;
    return rem < window ? rem : window;
}

// collision indicator of one state (mppi_differential_drive_obs.py:301-313,
// mppi_race_car_obstacle.py:241-274)
// The obstacle table lives in registers: lane m holds circle m {centre, squared radius} (one vector load issued at
// the top of the kernel, with all lanes active).  The loops below read one circle per iteration with v_readlane, so
// there is no memory wait inside them; a table walked through a pointer costs a load round trip per circle -- and
// the race car repeats the walk for each of its 8 outline points (config 4: 32 us per launch, 3/4 of it such waits).
// Circles beyond the 64th come from P.obs directly.
template <typename R> struct ObsLanes { R x, y, r2; };

template <typename R> __device__ __forceinline__ ObsLanes<R> load_obstacles(const KParams<R> &P, int lane) {
    ObsLanes<R> o{R(0), R(0), R(0)};
    if (P.obstacle_model != OBS_NONE && lane < P.n_obs) {
        const R *q = P.obs + 4 * lane;
        o.x = q[0]; o.y = q[1]; o.r2 = q[2];
    }
    return o;
}

// WIDE: the 8 outline points stay in registers and each circle is read once (the race-car kernels); otherwise the
// points are visited one after the other, which keeps the diff-drive kernels at their register count
// (`have_sc`: the caller already holds sin/cos of this yaw)
template <bool WIDE, typename R>
__device__ __forceinline__ bool collided(const KParams<R> &P, R x, R y, R yaw, const ObsLanes<R> &tab, bool have_sc = false,
                                         R sn_in = R(0), R cs_in = R(1)) {
    if (P.obstacle_model == OBS_NONE) return false;
    bool hit = false;
    const int n_reg = P.n_obs < 64 ? P.n_obs : 64;
    if (P.obstacle_model == OBS_CIRCLE) {
        for (int m = 0; m < n_reg; ++m) {
            const R dx = x - wv::read_lane(tab.x, m), dy = y - wv::read_lane(tab.y, m);
            hit |= dx * dx + dy * dy < wv::read_lane(tab.r2, m);
        }
        for (int m = 64; m < P.n_obs; ++m) {
            const R dx = x - P.obs[4 * m], dy = y - P.obs[4 * m + 1];
            hit |= dx * dx + dy * dy < P.obs[4 * m + 2];
        }
        return hit;
    }
    // OBS_OUTLINE; the reference's 9th point repeats the 1st (mppi_race_car_obstacle.py:263-264)
    R sn = sn_in, cs = cs_in;
    if (!have_sc) mf::sincos_(yaw, sn, cs);
    if (WIDE) {
        // The outline is the reference's fixed pattern -- (+-a, 0), (+-a, +-b), (0, +-b) in the body frame, a and b half the
        // scaled length and width (:260-261) -- and closed under mirroring in both axes.  So instead of rotating eight
        // points into the world, each circle's centre is rotated into the body frame and folded into the first quadrant,
        // where the nearest outline point is one of (a, 0), (a, b), (0, b): 16 instructions per circle instead of 32 per
        // pose + 43 per circle (14 % of the race-car launch with its two circles).  Same squared distances up to f32
        // rounding of the rotation.
        const R a = P.shape_x[3], b = P.shape_y[3];
        auto circle = [&](R ox, R oy, R r2) {
            const R dx = ox - x, dy = oy - y;
            const R bx = fabs(dx * cs + dy * sn), by = fabs(dy * cs - dx * sn);
            const R ex = bx - a, ey = by - b;
            const R ex2 = ex * ex, ey2 = ey * ey;
            hit |= fmin(fmin(ex2 + by * by, ex2 + ey2), bx * bx + ey2) < r2;
        };
        for (int m = 0; m < n_reg; ++m) circle(wv::read_lane(tab.x, m), wv::read_lane(tab.y, m), wv::read_lane(tab.r2, m));
        for (int m = 64; m < P.n_obs; ++m) circle(P.obs[4 * m], P.obs[4 * m + 1], P.obs[4 * m + 2]);
    } else {
        for (int q = 0; q < 8; ++q) {
            const R px = P.shape_x[q] * cs - P.shape_y[q] * sn + x;
            const R py = P.shape_x[q] * sn + P.shape_y[q] * cs + y;
            for (int m = 0; m < n_reg; ++m) {
                const R dx = px - wv::read_lane(tab.x, m), dy = py - wv::read_lane(tab.y, m);
                hit |= dx * dx + dy * dy < wv::read_lane(tab.r2, m);
            }
            for (int m = 64; m < P.n_obs; ++m) {
                const R dx = px - P.obs[4 * m], dy = py - P.obs[4 * m + 1];
                hit |= dx * dx + dy * dy < P.obs[4 * m + 2];
            }
        }
    }
    return hit;
}

// weighted squared tracking error against waypoint i (`_compute_cost` :222-236, `_c` mppi_race_car.py:137-146)
template <typename R, int MODEL>
__device__ __forceinline__ R tracking_cost_row(const KParams<R> &P, const R (&w)[4], bool wrap, const R *r, R x, R y, R yaw,
                                               R vel) {  // r: the waypoint's row {x, y, yaw, v}
    if (wrap) yaw = mf::pymod(yaw + P.two_pi, P.two_pi);
    const R ex = x - r[0], ey = y - r[1], eyaw = yaw - r[2];
    R c = w[0] * (ex * ex) + w[1] * (ey * ey) + w[2] * (eyaw * eyaw);
    if (MODEL == MODEL_RACE) {
        const R ev = vel - r[3];
        c += w[3] * (ev * ev);
    }
    return c;
}
template <typename R, int MODEL>
__device__ __forceinline__ R tracking_cost(const KParams<R> &P, const R (&w)[4], bool wrap, int i, R x, R y, R yaw,
                                           R vel) {
    return tracking_cost_row<R, MODEL>(P, w, wrap, P.ref + 4 * i, x, y, yaw, vel);
}


// controller state of this launch: *st, with the observed state / x0 index overridden by kernel arguments
// The controller state as ONE vector load (lane i <- dword i) + readlanes: a struct copy through scalar loads is
// split by the compiler into dependent pieces (pointer, then the field the first branch needs, then the rest),
// one memory round trip each.
__device__ __forceinline__ DevState load_state_words(const DevState *st) {
    static_assert(sizeof(DevState) == 72, "DevState layout");
    const int lane = wv::lane_id();
    const int w = reinterpret_cast<const int *>(st)[lane < 18 ? lane : 0];
    auto word = [&](int i) { return __builtin_amdgcn_readlane(w, i); };
    auto dbl = [&](int i) { return __builtin_bit_cast(double, ((long long)word(i + 1) << 32) | (unsigned int)word(i)); };
    DevState sv;
    sv.x0[0] = dbl(0); sv.x0[1] = dbl(2); sv.x0[2] = dbl(4); sv.x0[3] = dbl(6);
    sv.p = word(8); sv.c = word(9); sv.k_start = word(10); sv.first_k = word(11);
    sv.round = word(12); sv.path_end = word(13); sv.idx_start = word(14); sv.pad = 0;
    sv.iter = ((long long)word(17) << 32) | (unsigned int)word(16);
    return sv;
}

// `st` = P.st; the hot kernels take it as their FIRST kernel argument as well, where the dispatcher preloads it
// into SGPRs (-amdgpu-kernarg-preload-count), so that this load does not wait for the kernel-argument fetch
template <typename R> __device__ __forceinline__ DevState load_state(const KParams<R> &P, const DevState *st) {
    DevState sv = load_state_words(st);
    if (P.use_args) {
        sv.x0[0] = P.x0_arg[0]; sv.x0[1] = P.x0_arg[1]; sv.x0[2] = P.x0_arg[2]; sv.x0[3] = P.x0_arg[3];
        sv.c = P.c_arg;
    }
    return sv;
}

__device__ __forceinline__ bool round_unresolved(const DevState *st, int K) {
    const int fk = st->first_k;
    return fk != NO_TRIGGER && fk + 1 < K;
}

}  // namespace mppi
