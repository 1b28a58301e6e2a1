// gfx950 (MI355X) kernels of the MPPI iteration.
//
// Mapping (north star: one wavefront per trajectory): a wave owns sample k and spreads the
// HORIZON over its 64 lanes.  Both analytic models are triangular in time -- the unicycle's
// yaw is a prefix sum of the clamped turn rates and x/y are prefix sums of v*cos/sin(yaw)*dt
// (controllers/mppi_differential_drive.py:182-198); the bicycle cascades speed -> yaw ->
// position the same way (controllers/mppi_race_car.py:183-197) -- so a T-step rollout is
// 3-4 DPP wave scans plus ONE sincos per lane instead of a T-deep dependent chain.  Lane t
// then holds the state after step t and evaluates that step's waypoint search, cost and
// obstacle test; eps[k, :, :] is read as one coalesced 8-byte-per-lane row.
//
// Stages: S1 philox.h, S2-S4 k_rollout, S5-S6 k_reduce,
// S7 k_finalize.  file:line citations are relative to the reference repository.
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

template <typename R> __device__ __forceinline__ int window_len(int window, int n_ref, int c) {
    const int rem = n_ref - c;
    return rem < window ? rem : window;
}

// collision indicator of one state (mppi_differential_drive_obs.py:301-313,
// mppi_race_car_obstacle.py:241-274)
template <typename R> __device__ __forceinline__ bool collided(const KParams<R> &P, R x, R y, R yaw) {
    bool hit = false;
    if (P.obstacle_model == OBS_CIRCLE) {
        for (int m = 0; m < P.n_obs; ++m) {
            const R dx = x - P.obs[4 * m], dy = y - P.obs[4 * m + 1];
            hit |= dx * dx + dy * dy < P.obs[4 * m + 2];
        }
    } else if (P.obstacle_model == OBS_OUTLINE) {
        R sn, cs;
        mf::sincos_(yaw, sn, cs);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const R px = P.shape_x[q] * cs - P.shape_y[q] * sn + x;
            const R py = P.shape_x[q] * sn + P.shape_y[q] * cs + y;
            for (int m = 0; m < P.n_obs; ++m) {
                const R dx = px - P.obs[4 * m], dy = py - P.obs[4 * m + 1];
                hit |= dx * dx + dy * dy < P.obs[4 * m + 2];
            }
        }
    }
    return hit;
}

// weighted squared tracking error against waypoint i (`_compute_cost` :222-236, `_c` mppi_race_car.py:137-146)
template <typename R, int MODEL>
__device__ __forceinline__ R tracking_cost(const KParams<R> &P, const R (&w)[4], bool wrap, int i, R x, R y, R yaw,
                                           R vel) {
    const R *r = P.ref + 4 * i;
    if (wrap) yaw = mf::pymod(yaw + P.two_pi, P.two_pi);
    const R ex = x - r[0], ey = y - r[1], eyaw = yaw - r[2];
    R c = w[0] * (ex * ex) + w[1] * (ey * ey) + w[2] * (eyaw * eyaw);
    if (MODEL == MODEL_RACE) {
        const R ev = vel - r[3];
        c += w[3] * (ev * ev);
    }
    return c;
}

// ------------------------------------------------------------------------------------------
// x0 call: c <- nearest waypoint of the observed state, searched from prev_way_point_idx
// (mppi_differential_drive.py:96-99, mppi_race_car.py:61-65).  One wave, f64.
// ------------------------------------------------------------------------------------------
template <typename R>
__device__ __forceinline__ void x0_call(DevState *st, const R *ref, int n_ref, int window, int sequential, int lane,
                                        double x, double y, int p) {
    double best = INFINITY;
    int bj = INT_MAX;
    const int wlen = window_len<R>(window, n_ref, p);
    for (int j = lane; j < wlen; j += 64) {
        const double dx = x - (double)ref[4 * (p + j)], dy = y - (double)ref[4 * (p + j) + 1];
        const double d = dx * dx + dy * dy;
        if (d < best) { best = d; bj = j; }
    }
    wv::argmin_first(best, bj);
    if (lane == 0) {
        const int c = p + bj;
        st->c = c;  // the reference clamps to n_ref-1 here, which c already satisfies (:97-99)
        st->idx_start = c;
        st->path_end = c >= n_ref - 1;
        if (!sequential) st->p = c;  // update_prev_idx=True at x0 only (mppi_race_car.py:61)
        st->k_start = 0;
        st->first_k = NO_TRIGGER;
        st->round = 0;
    }
}

template <typename R>
__global__ __launch_bounds__(64) void k_set_state(const R *ref, int n_ref, int window, int sequential, DevState *st,
                                                  double x0, double x1, double x2, double x3, int have_x) {
    const int lane = threadIdx.x;
    if (have_x) {
        if (lane == 0) { st->x0[0] = x0; st->x0[1] = x1; st->x0[2] = x2; st->x0[3] = x3; }
    } else {
        x0 = st->x0[0];
        x1 = st->x0[1];
    }
    x0_call<R>(st, ref, n_ref, window, sequential, lane, x0, x1, st->p);
}

// ------------------------------------------------------------------------------------------
// S2-S4: perturb + clamp, rollout, cost.  One wave per sample, lanes over the horizon.
// ------------------------------------------------------------------------------------------
template <typename R, int MODEL> struct Rollout {
    const KParams<R> &P;
    const int k, lane, c;
    const unsigned iter;
    const bool exploit;  // k < (1-expl)*K, :116
    R cx, cy, cyaw, cvel;  // state carried from chunk to chunk of 64 steps (wave-uniform)
    int p;                 // sequential mode: the threaded waypoint index
    bool slow;             // sequential mode: some call moved the index, evolve it call by call
    R s_acc, s_last;
    const int n_chunk, lane_last;

    __device__ __forceinline__ Rollout(const KParams<R> &P_, const DevState *st, int k_, int lane_)
        : P(P_), k(k_), lane(lane_), c(st->c), iter((unsigned)st->iter),
          exploit((k_ + P_.k_offset) < P_.n_exploit), cx((R)st->x0[0]), cy((R)st->x0[1]), cyaw((R)st->x0[2]),
          cvel(MODEL == MODEL_RACE ? (R)st->x0[3] : R(0)), p(st->c), slow(false), s_acc(0), s_last(0),
          n_chunk((P_.T + 63) >> 6), lane_last((P_.T - 1) & 63) {}

    // this lane's noise for step t of sample k (S1, or the caller's tensor)
    __device__ __forceinline__ void load_eps(int ch, float &e0, float &e1) const {
        const int t = ch * 64 + lane;
        e0 = 0.f;
        e1 = 0.f;
        if (t < P.T) {
            if (P.use_philox) {
                px::sample(P.seed_lo, P.seed_hi, iter, (unsigned)(k + P.k_offset), t, P.chol, e0, e1);
            } else {
                const float2 e = *reinterpret_cast<const float2 *>(P.eps + ((size_t)k * P.T + t) * 2);
                e0 = e.x;
                e1 = e.y;
            }
        }
    }

    // steps [64 ch, 64 ch + 64) of the horizon; returns the noise it used
    __device__ __forceinline__ void chunk(int ch, float &e0, float &e1) {
        const R *__restrict__ ref = P.ref;
        const int t = ch * 64 + lane;
        const bool act = t < P.T;
        load_eps(ch, e0, e1);
        R u0 = 0, u1 = 0;
        if (act) {
            u0 = P.u[2 * t];
            u1 = P.u[2 * t + 1];
        }
        R v0 = exploit ? u0 + (R)e0 : (R)e0, v1 = exploit ? u1 + (R)e1 : (R)e1;  // :116-119
        if (P.clamp_rollout) {                                                   // `_g` :285-289
            v0 = mf::clamp(v0, P.umax0);
            v1 = mf::clamp(v1, P.umax1);
        }
        if (!act) { v0 = 0; v1 = 0; }

        // ---- dynamics as wave scans -----------------------------------------------------
        R x, y, yaw, vel = 0;
        if (MODEL == MODEL_DIFF) {  // :194-196
            const R dyaw = v1 * P.dt;
            yaw = cyaw + wv::scan_incl<wv::OpAdd>(dyaw);
            const R yaw_b = wv::shift_up1(yaw, cyaw);  // yaw before the step
            R sn, cs;
            mf::sincos_(yaw_b, sn, cs);
            x = cx + wv::scan_incl<wv::OpAdd>(v0 * cs * P.dt);
            y = cy + wv::scan_incl<wv::OpAdd>(v0 * sn * P.dt);
        } else {  // mppi_race_car.py:190-193, controls = [steer, accel]
            const R dvel = act ? v1 * P.dt : R(0);
            vel = cvel + wv::scan_incl<wv::OpAdd>(dvel);
            const R vel_b = wv::shift_up1(vel, cvel);
            const R dyaw = act ? vel_b / P.wheel_base * mf::tan_(v0) * P.dt : R(0);
            yaw = cyaw + wv::scan_incl<wv::OpAdd>(dyaw);
            const R yaw_b = wv::shift_up1(yaw, cyaw);
            R sn, cs;
            mf::sincos_(yaw_b, sn, cs);
            x = cx + wv::scan_incl<wv::OpAdd>(act ? vel_b * cs * P.dt : R(0));
            y = cy + wv::scan_incl<wv::OpAdd>(act ? vel_b * sn * P.dt : R(0));
        }
        cx = wv::read_lane(x, 63);
        cy = wv::read_lane(y, 63);
        cyaw = wv::read_lane(yaw, 63);
        if (MODEL == MODEL_RACE) cvel = wv::read_lane(vel, 63);

        // ---- waypoint index of every call in this chunk ----------------------------------
        int my_idx;
        if (!P.sequential) {
            my_idx = nearest_in_window(ref, c, window_len<R>(P.window, P.n_ref, c), x, y);
        } else {
            if (!slow) {  // does any call move the index away from p?
                const int wlen = window_len<R>(P.window, P.n_ref, p);
                const R d0 = dist2(ref, p, x, y);
                bool trig = false;
#pragma unroll 4
                for (int j = 1; j < wlen; ++j) trig |= dist2(ref, p + j, x, y) < d0;
                slow = __ballot(trig && act) != 0ull;
            }
            my_idx = p;
            if (slow) {  // rare: thread the index through this chunk's calls in order
                const int n_act = min(64, P.T - ch * 64);
                for (int tt = 0; tt < n_act; ++tt) {
                    const R xt = wv::read_lane(x, tt), yt = wv::read_lane(y, tt);
                    p = nearest_uniform(ref, p, window_len<R>(P.window, P.n_ref, p), xt, yt, lane);
                    if (lane == tt) my_idx = p;
                }
            }
        }

        // ---- stage cost of every call (only the last one survives when !accumulate) ------
        const bool last_chunk = ch == n_chunk - 1;
        if (P.accumulate || last_chunk) {
            const bool hit = collided(P, x, y, yaw);
            R st_c = tracking_cost<R, MODEL>(P, P.ws, P.wrap_stage, my_idx, x, y, yaw, vel);
            if (hit) st_c += P.penalty;
            R ctrl;
            if (MODEL == MODEL_DIFF)  // u^T Sigma^-1 v, :124
                ctrl = (u0 * P.sinv[0] + u1 * P.sinv[2]) * v0 + (u0 * P.sinv[1] + u1 * P.sinv[3]) * v1;
            else  // u (Sigma^-1 v), mppi_race_car.py:84
                ctrl = u0 * (P.sinv[0] * v0 + P.sinv[1] * v1) + u1 * (P.sinv[2] * v0 + P.sinv[3] * v1);
            const R stage = st_c + P.gamma * ctrl;
            if (P.accumulate) {
                if (sizeof(R) == 4) {
                    // `S[k] += ...` one step at a time (mppi_race_car.py:84): with 1e10 collision penalties
                    // in f32 (ulp 1024) the order of the additions decides which tracking terms survive, so
                    // the f32 kernels add in the reference's order (s_acc is wave-uniform here).
                    const int n_act = min(64, P.T - ch * 64);
                    for (int tt = 0; tt < n_act; ++tt) s_acc += wv::read_lane(stage, tt);
                } else {
                    s_acc += act ? stage : R(0);
                }
            }
            if (last_chunk) {
                // terminal call: same state; the sequential index takes one more step (:244)
                int idx_term = my_idx;
                if (P.sequential && slow) {
                    const R xt = wv::read_lane(x, lane_last), yt = wv::read_lane(y, lane_last);
                    p = nearest_uniform(ref, p, window_len<R>(P.window, P.n_ref, p), xt, yt, lane);
                    idx_term = p;
                }
                R term = tracking_cost<R, MODEL>(P, P.wt, P.wrap_term, idx_term, x, y, yaw, vel);
                if (hit) term += P.penalty;
                s_last = P.accumulate ? term : stage + term;
            }
        }
    }

    // S[k] once every chunk has run; also publishes the index this sample leaves behind
    __device__ __forceinline__ R finish() {
        R total = wv::read_lane(s_last, lane_last);
        if (P.accumulate) total = (sizeof(R) == 4 ? s_acc : wv::reduce<wv::OpAdd>(s_acc)) + total;
        if (lane == 0) {
            P.S[k] = total;
            P.pout[k] = p;
            if (P.sequential && p != c) atomicMin(&P.st->first_k, k);
        }
        return total;
    }
};

// Any horizon: S[k] only (the softmin partials come from k_reduce).
template <typename R, int MODEL>
__global__ __launch_bounds__(256) void k_rollout(const KParams<R> P) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // wave-uniform
    const DevState *st = P.st;
    if (k >= P.K || k < st->k_start) return;
    Rollout<R, MODEL> r(P, st, k, lane);
    for (int ch = 0; ch < r.n_chunk; ++ch) {
        float e0, e1;
        r.chunk(ch, e0, e1);
    }
    r.finish();
}

// T <= 64 NCH: rollout + cost + the block's softmin partial {rho_b, eta_b, eta2_b, W_b[T][2]} in one launch.
// The noise stays in registers between the rollout and the weighted sum (no second pass over eps, S5-S6
// fused into S2-S4); the block's waves meet once in LDS.  Samples below k_start (already final in an
// earlier speculation round) re-enter with their stored cost.
constexpr int FUSED_WAVES = 16;

template <typename R, int MODEL, int NCH>
__global__ __launch_bounds__(64 * FUSED_WAVES) void k_rollout_fused(const KParams<R> P,
                                                                      double *__restrict__ partials) {
    __shared__ R sh_S[FUSED_WAVES];
    __shared__ R sh_acc[FUSED_WAVES][128 * NCH];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int k = blockIdx.x * FUSED_WAVES + wid;  // wave-uniform
    const DevState *st = P.st;
    const int k_start = st->k_start;
    if ((blockIdx.x + 1) * FUSED_WAVES <= k_start) return;  // every sample final: the old partial stands
    const bool valid = k < P.K;
    float e0[NCH], e1[NCH];
    R S_k = R(INFINITY);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) { e0[ch] = 0.f; e1[ch] = 0.f; }
    if (valid) {
        Rollout<R, MODEL> r(P, st, k, lane);
        if (k >= k_start) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                if (ch < r.n_chunk) r.chunk(ch, e0[ch], e1[ch]);
            S_k = r.finish();
        } else {
            S_k = P.S[k];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) r.load_eps(ch, e0[ch], e1[ch]);
        }
    }
    if (lane == 0) sh_S[wid] = S_k;
    __syncthreads();
    R rho = sh_S[0];
#pragma unroll
    for (int w = 1; w < FUSED_WAVES; ++w) rho = fmin(rho, sh_S[w]);
    const R e = valid ? mf::exp_(-P.beta * (S_k - rho)) : R(0);  // :175
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        sh_acc[wid][ch * 128 + 2 * lane] = e * (R)e0[ch];
        sh_acc[wid][ch * 128 + 2 * lane + 1] = e * (R)e1[ch];
    }
    __syncthreads();
    double *out = partials + (size_t)blockIdx.x * partial_len(P.T);
    for (int i = threadIdx.x; i < 2 * P.T; i += blockDim.x) {  // W_b[t] = sum_k e_k eps[k, t], :132-135
        R s = 0;
#pragma unroll
        for (int w = 0; w < FUSED_WAVES; ++w) s += sh_acc[w][i];
        out[3 + i] = (double)s;
    }
    if (threadIdx.x == 0) {
        R eta = 0, eta2 = 0;
        for (int w = 0; w < FUSED_WAVES; ++w) {
            const R ew = blockIdx.x * FUSED_WAVES + w < P.K ? mf::exp_(-P.beta * (sh_S[w] - rho)) : R(0);
            eta += ew;
            eta2 += ew * ew;
        }
        out[0] = (double)rho;
        out[1] = (double)eta;
        out[2] = (double)eta2;
    }
}

// ------------------------------------------------------------------------------------------
// S5-S6: block-local softmin partials {rho_b, eta_b, eta2_b, W_b[T][2]}.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool round_unresolved(const DevState *st, int K) {
    const int fk = st->first_k;
    return fk != NO_TRIGGER && fk + 1 < K;
}

template <typename R>
__global__ __launch_bounds__(256) void k_reduce(const KParams<R> P, double *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevState *st = P.st;
    if (round_unresolved(st, P.K)) return;  // a repair round will recompute S first
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    const int k0 = blockIdx.x * P.traj_per_block, k1 = min(P.K, k0 + P.traj_per_block), nk = k1 - k0;
    R *sh_e = reinterpret_cast<R *>(smem);                 // [traj_per_block]
    R *sh_red = sh_e + P.traj_per_block;                   // [3 * 4]
    R *sh_acc = sh_red + 16;                               // [nw][128] (one chunk of 64 steps x 2)
    const unsigned iter = (unsigned)st->iter;

    R m = R(INFINITY);
    for (int i = tid; i < nk; i += blockDim.x) m = fmin(m, P.S[k0 + i]);
    m = wv::reduce<wv::OpMin>(m);
    if (lane == 0) sh_red[wid] = m;
    __syncthreads();
    R rho = sh_red[0];
    for (int w = 1; w < nw; ++w) rho = fmin(rho, sh_red[w]);
    __syncthreads();

    R eta = 0, eta2 = 0;
    for (int i = tid; i < nk; i += blockDim.x) {
        const R e = mf::exp_(-P.beta * (P.S[k0 + i] - rho));  // :175
        sh_e[i] = e;
        eta += e;
        eta2 += e * e;
    }
    eta = wv::reduce<wv::OpAdd>(eta);
    eta2 = wv::reduce<wv::OpAdd>(eta2);
    if (lane == 0) { sh_red[4 + wid] = eta; sh_red[8 + wid] = eta2; }
    __syncthreads();
    double *out = partials + (size_t)blockIdx.x * partial_len(P.T);
    if (tid == 0) {
        R a = 0, b = 0;
        for (int w = 0; w < nw; ++w) { a += sh_red[4 + w]; b += sh_red[8 + w]; }
        out[0] = (double)rho;
        out[1] = (double)a;
        out[2] = (double)b;
    }

    // W_b[t] = sum_k e_k eps[k, t]  (:132-135 with the 1/eta factored out), lanes over t
    const int n_chunk = (P.T + 63) >> 6;
    for (int ch = 0; ch < n_chunk; ++ch) {
        const int t = ch * 64 + lane;
        R a0 = 0, a1 = 0;
        if (t < P.T) {
            for (int i = wid; i < nk; i += nw) {
                float e0, e1;
                if (P.use_philox) {
                    px::sample(P.seed_lo, P.seed_hi, iter, (unsigned)(k0 + i + P.k_offset), t, P.chol, e0, e1);
                } else {
                    const float2 e = *reinterpret_cast<const float2 *>(P.eps + ((size_t)(k0 + i) * P.T + t) * 2);
                    e0 = e.x;
                    e1 = e.y;
                }
                const R w = sh_e[i];
                a0 += w * (R)e0;
                a1 += w * (R)e1;
            }
        }
        sh_acc[wid * 128 + 2 * lane] = a0;
        sh_acc[wid * 128 + 2 * lane + 1] = a1;
        __syncthreads();
        if (tid < 128 && ch * 64 + (tid >> 1) < P.T) {
            R s = 0;
            for (int w = 0; w < nw; ++w) s += sh_acc[w * 128 + tid];
            out[3 + ch * 128 + tid] = (double)s;
        }
        __syncthreads();
    }
}

// normalised weights of the last iteration (`_compute_weight` :167-180), for inspection
template <typename R>
__global__ void k_weights(const R *__restrict__ S, int K, double beta, double rho, double eta, double *w) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) w[k] = exp(-beta * ((double)S[k] - rho)) / eta;
}

// ------------------------------------------------------------------------------------------
// S7: merge partials, moving-average filter, update, clamp, shift, plant.  One block, f64.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double load_real(const void *p, int i, int is_f64) {
    return is_f64 ? ((const double *)p)[i] : (double)((const float *)p)[i];
}
__device__ __forceinline__ void store_real(void *p, int i, int is_f64, double v) {
    if (is_f64) ((double *)p)[i] = v;
    else ((float *)p)[i] = (float)v;
}

// Block-wide merge of n softmin partial records with the rescale trick:
// rho = min rho_b, s_b = exp(-beta (rho_b - rho)), eta = sum s_b eta_b, W = sum s_b W_b.  The loads of
// W are independent of rho, laid out so that a 1024-thread block keeps ~n*2T/1024 of them in flight per
// thread (the records come from other CUs' plain stores of a PREVIOUS launch, so ordinary loads are fine).
// Result: W in sh_w[0, 2T); rho/eta/eta2 returned to every thread.
constexpr int MERGE_THREADS = 1024, MERGE_GROUPS = MERGE_THREADS / 128;

__device__ __forceinline__ double block_reduce_min(double v, double *sh_red, int tid) {
    v = wv::reduce<wv::OpMin>(v);
    __syncthreads();
    if ((tid & 63) == 0) sh_red[tid >> 6] = v;
    __syncthreads();
    double r = sh_red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmin(r, sh_red[w]);
    return r;
}
__device__ __forceinline__ double block_reduce_add(double v, double *sh_red, int tid) {
    v = wv::reduce<wv::OpAdd>(v);
    __syncthreads();
    if ((tid & 63) == 0) sh_red[tid >> 6] = v;
    __syncthreads();
    double r = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh_red[w];
    return r;
}

__device__ __forceinline__ void merge_records(const double *__restrict__ recs, int n, int T, double beta, double *sh_w,
                                              double *sh_s, double *sh_red, double *sh_part, double &rho,
                                              double &eta, double &eta2) {
    const int tid = threadIdx.x, plen = partial_len(T);
    double m = INFINITY;
    for (int b = tid; b < n; b += blockDim.x) m = fmin(m, recs[(size_t)b * plen]);
    rho = block_reduce_min(m, sh_red, tid);
    double a = 0, a2 = 0;
    for (int b = tid; b < n; b += blockDim.x) {
        const double *pb = recs + (size_t)b * plen;
        const double sc = exp(-beta * (pb[0] - rho));
        sh_s[b] = sc;
        a += sc * pb[1];
        a2 += sc * sc * pb[2];
    }
    eta = block_reduce_add(a, sh_red, tid);
    eta2 = block_reduce_add(a2, sh_red, tid);  // (the barriers inside also publish sh_s)
    const int col = tid & 127, grp = tid >> 7, ngrp = blockDim.x >> 7;
    for (int i0 = 0; i0 < 2 * T; i0 += 128) {
        const int i = i0 + col;
        double acc = 0;
        if (i < 2 * T) {
#pragma unroll 8
            for (int b = grp; b < n; b += ngrp) acc += sh_s[b] * recs[(size_t)b * plen + 3 + i];
        }
        sh_part[grp * 128 + col] = acc;
        __syncthreads();
        if (grp == 0 && i < 2 * T) {
            double t = 0;
            for (int g = 0; g < ngrp; ++g) t += sh_part[g * 128 + col];
            sh_w[i] = t;
        }
        __syncthreads();
    }
}

// groups of `group` records -> one record each (large K, and the per-rank record of the split step)
__global__ __launch_bounds__(MERGE_THREADS) void k_merge(const double *__restrict__ recs, int n, int group, int T,
                                                         double beta, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *sh_w = reinterpret_cast<double *>(smem);
    double *sh_s = sh_w + 2 * T, *sh_red = sh_s + group, *sh_part = sh_red + 64;
    const int b0 = blockIdx.x * group, nb = min(group, n - b0);
    double rho, eta, eta2;
    merge_records(recs + (size_t)b0 * partial_len(T), nb, T, beta, sh_w, sh_s, sh_red, sh_part, rho, eta, eta2);
    double *o = out + (size_t)blockIdx.x * partial_len(T);
    for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) o[3 + i] = sh_w[i];
    if (threadIdx.x == 0) { o[0] = rho; o[1] = eta; o[2] = eta2; }
}

__global__ __launch_bounds__(MERGE_THREADS) void k_finalize(const FinalizeParams F) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *sh_w = reinterpret_cast<double *>(smem);  // [2T] weighted noise, then filtered
    double *sh_u = sh_w + 2 * F.T;                    // [2T] updated u
    double *sh_s = sh_u + 2 * F.T;                    // [n_part] scale factors
    double *sh_red = sh_s + F.n_part;                 // [64]
    double *sh_part = sh_red + 64;                    // [MERGE_GROUPS][128]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    DevState *st = F.st;
    StepResult *res = F.res;
    double *res_u = res ? reinterpret_cast<double *>(res + 1) : nullptr;

    // --- sequential-waypoint speculation: did a sample move the index? ---------------------
    int c_final = st->c;
    if (F.sequential) {
        const int fk = st->first_k;
        if (fk != NO_TRIGGER) {
            const int c_new = F.pout[fk];
            if (fk + 1 < F.K) {  // samples after fk were evaluated from a stale index: another round
                if (tid == 0) {
                    st->k_start = fk + 1;
                    st->c = c_new;
                    st->first_k = NO_TRIGGER;
                    st->round = st->round + 1;
                    res->status = STATUS_NEED_ROUND;
                    res->k_next = fk + 1;
                    res->c_next = c_new;
                    res->rounds = st->round;
                    res->iter = st->iter;
                }
                return;
            }
            c_final = c_new;
        }
    }

    double rho, eta, eta2;
    merge_records(F.partials, F.n_part, F.T, F.beta, sh_w, sh_s, sh_red, sh_part, rho, eta, eta2);
    const bool path_end_abort = F.raise_at_path_end && st->path_end;  // mppi_race_car.py:63-65

    // --- w_eps, moving average (window W) -----------------------------------------------
    const int T = F.T, W = F.filter_window, H = W / 2;
    for (int i = tid; i < 2 * T; i += blockDim.x) sh_w[i] = sh_w[i] / eta;
    __syncthreads();
    for (int i = tid; i < 2 * T; i += blockDim.x) {
        const int t = i >> 1, d = i & 1;
        double f;
        if (F.filter_mode == FILTER_DIFF) {  // np.convolve(x, ones(W)/W, 'same'): taps t-H .. t+W-1-H
            double s = 0;
            for (int j = t + W - 1 - H; j >= t - H; --j)
                if (j >= 0 && j < T) s += sh_w[2 * j + d] * (1.0 / W);
            const int n_conv = (W + 1) / 2;  // mppi_differential_drive.py:265-269
            if (t == 0) s *= (double)W / n_conv;
            else if (t < n_conv) s *= (double)W / (t + n_conv);
            if (t == T - 1)
                for (int q = 1; q < n_conv; ++q) s *= (double)W / (q + n_conv - (W % 2));
            f = s;
        } else if (F.filter_mode == FILTER_RACE) {  // mppi_race_car.py:211-222
            double s = 0;
            for (int j = t + W - 1; j >= t; --j) {  // index into the padded signal
                const int src = j < H ? j : (j < T + H ? j - H : j - 2 * H);
                s += sh_w[2 * src + d] * (1.0 / W);
            }
            f = s;
        } else {
            f = sh_w[i];
        }
        double un = load_real(F.u, i, F.is_f64) + f;  // u += w_epsilon, :141
        if (F.clamp_u) un = mf::clamp(un, d == 0 ? F.umax0 : F.umax1);  // :145-149
        sh_u[i] = un;
    }
    __syncthreads();
    if (path_end_abort) {
        if (tid == 0) {
            res->status = STATUS_PATH_END;
            res->idx_start = st->idx_start; res->idx_after = st->p; res->path_end = 1;
            res->rounds = st->round + 1; res->iter = st->iter;
            st->first_k = NO_TRIGGER; st->k_start = 0;
        }
        return;
    }
    // --- shift (:162-163); the returned sequence aliases u_prev (:165) ---------------------
    for (int i = tid; i < 2 * T; i += blockDim.x) {
        const int t = i >> 1, d = i & 1;
        const double old = load_real(F.u, i, F.is_f64);
        const double shifted = sh_u[2 * (t < T - 1 ? t + 1 : T - 1) + d];
        store_real(F.u_before, i, F.is_f64, old);
        store_real(F.u_before, 2 * T + i, F.is_f64, sh_u[i]);
        res_u[i] = shifted;
        store_real(F.u, i, F.is_f64, shifted);  // element i is read and written by this thread only
    }

    if (wid == 0) {
        const double u0a = sh_u[2 * (T > 1 ? 1 : 0)], u0b = sh_u[2 * (T > 1 ? 1 : 0) + 1];
        double xn[4] = {st->x0[0], st->x0[1], st->x0[2], st->x0[3]};
        if (F.plant) {  // the driver's plant with the returned control
            if (F.model == MODEL_DIFF) {  // DifferentialDrive.update_state :33-40
                const double yaw = xn[2];
                xn[0] += u0a * cos(yaw) * F.dt;
                xn[1] += u0a * sin(yaw) * F.dt;
                xn[2] += u0b * F.dt;
            } else {  // Vehicle.update models/vehicle.py:85-114
                const double steer = mf::clamp(u0a, F.umax0), accel = mf::clamp(u0b, F.umax1);
                const double yaw = xn[2], v = xn[3];
                xn[0] += v * cos(yaw) * F.dt;
                xn[1] += v * sin(yaw) * F.dt;
                xn[2] += v / F.wheel_base * tan(steer) * F.dt;
                xn[3] += accel * F.dt;
            }
        }
        if (lane == 0) {
            res->status = STATUS_DONE;
            res->k_next = 0; res->c_next = c_final;
            res->idx_start = st->idx_start;
            res->idx_after = F.sequential ? c_final : st->p;
            res->path_end = st->path_end;
            res->rounds = st->round + 1;
            res->rho = rho; res->eta = eta; res->ess = eta * eta / eta2;
            res->u0[0] = u0a; res->u0[1] = u0b;
            for (int q = 0; q < 4; ++q) res->x_next[q] = xn[q];
            if (F.u0_trace) { F.u0_trace[2 * st->iter] = u0a; F.u0_trace[2 * st->iter + 1] = u0b; }
            if (F.sequential) st->p = c_final;
            st->iter = st->iter + 1;
            res->iter = st->iter;
            if (F.plant) for (int q = 0; q < 4; ++q) st->x0[q] = xn[q];
        }
        if (F.plant) {  // next iteration's x0 call, so the next slot needs no host input
            const int p_now = F.sequential ? c_final : st->p;
            if (F.is_f64)
                x0_call<double>(st, (const double *)F.ref, F.n_ref, F.window, F.sequential, lane, xn[0], xn[1], p_now);
            else
                x0_call<float>(st, (const float *)F.ref, F.n_ref, F.window, F.sequential, lane, xn[0], xn[1], p_now);
        } else if (lane == 0) {
            st->first_k = NO_TRIGGER;
            st->k_start = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// S1 materialised (`_calc_epsilon`), and the visualisation rollouts (:144-159)
// ------------------------------------------------------------------------------------------
__global__ void k_sample(unsigned seed_lo, unsigned seed_hi, unsigned iter, int K, int T, int k_offset, float l00,
                         float l10, float l11, float *__restrict__ eps) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)K * T) return;
    const int k = (int)(i / T), t = (int)(i % T);
    const float chol[3] = {l00, l10, l11};
    float e0, e1;
    px::sample(seed_lo, seed_hi, iter, (unsigned)(k + k_offset), t, chol, e0, e1);
    reinterpret_cast<float2 *>(eps)[i] = make_float2(e0, e1);
}

// Row `row` of the output: row < 0 is the optimal trajectory driven by the updated u,
// row >= 0 sample k's trajectory driven by its clamped v.  Step t uses control (t-1) mod T.
template <typename R, int MODEL>
__global__ __launch_bounds__(256) void k_viz(const KParams<R> P, const R *__restrict__ u_before,
                                             const R *__restrict__ u_upd, unsigned iter, float *__restrict__ opt,
                                             float *__restrict__ smp) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) - 1;
    if (row >= P.K) return;
    if (row < 0 && !opt) return;
    if (row >= 0 && !smp) return;
    const DevState *st = P.st;
    constexpr int NX = MODEL == MODEL_RACE ? 4 : 3;
    R cx = (R)st->x0[0], cy = (R)st->x0[1], cyaw = (R)st->x0[2], cvel = MODEL == MODEL_RACE ? (R)st->x0[3] : R(0);
    const bool exploit = (row + P.k_offset) < P.n_exploit;
    float *dst = row < 0 ? opt : smp + (size_t)row * P.T * NX;
    const int n_chunk = (P.T + 63) >> 6;
    for (int ch = 0; ch < n_chunk; ++ch) {
        const int t = ch * 64 + lane;
        const bool act = t < P.T;
        const int tc = (t + P.T - 1) % P.T;
        R v0 = 0, v1 = 0;
        if (act) {
            if (row < 0) {
                v0 = mf::clamp(u_upd[2 * tc], P.umax0);
                v1 = mf::clamp(u_upd[2 * tc + 1], P.umax1);
            } else {
                float e0, e1;
                if (P.use_philox) {
                    px::sample(P.seed_lo, P.seed_hi, iter, (unsigned)(row + P.k_offset), tc, P.chol, e0, e1);
                } else {
                    const float2 e = *reinterpret_cast<const float2 *>(P.eps + ((size_t)row * P.T + tc) * 2);
                    e0 = e.x;
                    e1 = e.y;
                }
                v0 = exploit ? u_before[2 * tc] + (R)e0 : (R)e0;
                v1 = exploit ? u_before[2 * tc + 1] + (R)e1 : (R)e1;
                v0 = mf::clamp(v0, P.umax0);  // the viz loop clamps even where the rollout did not (:158)
                v1 = mf::clamp(v1, P.umax1);
            }
        }
        R x, y, yaw, vel = 0;
        if (MODEL == MODEL_DIFF) {
            yaw = cyaw + wv::scan_incl<wv::OpAdd>(v1 * P.dt);
            const R yaw_b = wv::shift_up1(yaw, cyaw);
            R sn, cs;
            mf::sincos_(yaw_b, sn, cs);
            x = cx + wv::scan_incl<wv::OpAdd>(v0 * cs * P.dt);
            y = cy + wv::scan_incl<wv::OpAdd>(v0 * sn * P.dt);
        } else {
            vel = cvel + wv::scan_incl<wv::OpAdd>(act ? v1 * P.dt : R(0));
            const R vel_b = wv::shift_up1(vel, cvel);
            yaw = cyaw + wv::scan_incl<wv::OpAdd>(act ? vel_b / P.wheel_base * mf::tan_(v0) * P.dt : R(0));
            const R yaw_b = wv::shift_up1(yaw, cyaw);
            R sn, cs;
            mf::sincos_(yaw_b, sn, cs);
            x = cx + wv::scan_incl<wv::OpAdd>(act ? vel_b * cs * P.dt : R(0));
            y = cy + wv::scan_incl<wv::OpAdd>(act ? vel_b * sn * P.dt : R(0));
        }
        cx = wv::read_lane(x, 63); cy = wv::read_lane(y, 63); cyaw = wv::read_lane(yaw, 63);
        if (MODEL == MODEL_RACE) cvel = wv::read_lane(vel, 63);
        if (act) {
            dst[(size_t)t * NX] = (float)x;
            dst[(size_t)t * NX + 1] = (float)y;
            dst[(size_t)t * NX + 2] = (float)yaw;
            if (MODEL == MODEL_RACE) dst[(size_t)t * NX + 3] = (float)vel;
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int reduce_blocks(int K, int traj_per_block) { return (K + traj_per_block - 1) / traj_per_block; }

template <typename R> void launch_set_state(const KParams<R> &P, const double *x, hipStream_t s) {
    const double z[4] = {0, 0, 0, 0};
    const double *v = x ? x : z;
    hipLaunchKernelGGL(k_set_state<R>, dim3(1), dim3(64), 0, s, P.ref, P.n_ref, P.window, P.sequential, P.st, v[0],
                       v[1], v[2], v[3], x ? 1 : 0);
}

template <typename R> void launch_rollout(const KParams<R> &P, hipStream_t s) {
    const int waves_per_block = 4, blocks = (P.K + waves_per_block - 1) / waves_per_block;
    if (P.model == MODEL_DIFF)
        hipLaunchKernelGGL((k_rollout<R, MODEL_DIFF>), dim3(blocks), dim3(64 * waves_per_block), 0, s, P);
    else
        hipLaunchKernelGGL((k_rollout<R, MODEL_RACE>), dim3(blocks), dim3(64 * waves_per_block), 0, s, P);
}

bool fused_supported(int T) { return T <= 128; }
int fused_blocks(int K) { return (K + FUSED_WAVES - 1) / FUSED_WAVES; }

template <typename R, int MODEL> static void launch_fused_m(const KParams<R> &P, double *partials, hipStream_t s) {
    const dim3 grid(fused_blocks(P.K)), block(64 * FUSED_WAVES);
    if (P.T <= 64)
        hipLaunchKernelGGL((k_rollout_fused<R, MODEL, 1>), grid, block, 0, s, P, partials);
    else
        hipLaunchKernelGGL((k_rollout_fused<R, MODEL, 2>), grid, block, 0, s, P, partials);
}

template <typename R> void launch_rollout_fused(const KParams<R> &P, double *partials, hipStream_t s) {
    if (P.model == MODEL_DIFF) launch_fused_m<R, MODEL_DIFF>(P, partials, s);
    else launch_fused_m<R, MODEL_RACE>(P, partials, s);
}

template <typename R> void launch_reduce(const KParams<R> &P, double *partials, int n_blocks, hipStream_t s) {
    const size_t shmem = sizeof(R) * ((size_t)P.traj_per_block + 16 + 4 * 128);
    hipLaunchKernelGGL(k_reduce<R>, dim3(n_blocks), dim3(256), shmem, s, P, partials);
}

void launch_merge(const double *recs, int n, int group, int T, double beta, double *out, hipStream_t s) {
    const int blocks = (n + group - 1) / group;
    const size_t shmem = sizeof(double) * ((size_t)2 * T + group + 64 + MERGE_GROUPS * 128);
    hipLaunchKernelGGL(k_merge, dim3(blocks), dim3(MERGE_THREADS), shmem, s, recs, n, group, T, beta, out);
}

void launch_finalize(const FinalizeParams &F, hipStream_t s) {
    const size_t shmem = sizeof(double) * ((size_t)4 * F.T + F.n_part + 64 + MERGE_GROUPS * 128);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(MERGE_THREADS), shmem, s, F);
}

template <typename R> void launch_weights(const KParams<R> &P, double rho, double eta, double *w, hipStream_t s) {
    hipLaunchKernelGGL(k_weights<R>, dim3((P.K + 255) / 256), dim3(256), 0, s, P.S, P.K, (double)P.beta, rho, eta, w);
}

void launch_sample(unsigned seed_lo, unsigned seed_hi, unsigned iter, int K, int T, int k_offset, const float *chol,
                   float *eps_out, hipStream_t s) {
    const size_t n = (size_t)K * T;
    hipLaunchKernelGGL(k_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed_lo, seed_hi, iter, K, T,
                       k_offset, chol[0], chol[1], chol[2], eps_out);
}

template <typename R>
void launch_viz(const KParams<R> &P, const R *u_before, const R *u_upd, long long iter, float *opt, float *smp,
                hipStream_t s) {
    const int rows = P.K + 1, blocks = (rows + 3) / 4;
    if (P.model == MODEL_DIFF)
        hipLaunchKernelGGL((k_viz<R, MODEL_DIFF>), dim3(blocks), dim3(256), 0, s, P, u_before, u_upd, (unsigned)iter,
                           opt, smp);
    else
        hipLaunchKernelGGL((k_viz<R, MODEL_RACE>), dim3(blocks), dim3(256), 0, s, P, u_before, u_upd, (unsigned)iter,
                           opt, smp);
}

#define INSTANTIATE(R)                                                                                    \
    template void launch_set_state<R>(const KParams<R> &, const double *, hipStream_t);                   \
    template void launch_rollout<R>(const KParams<R> &, hipStream_t);                                     \
    template void launch_reduce<R>(const KParams<R> &, double *, int, hipStream_t);                       \
    template void launch_rollout_fused<R>(const KParams<R> &, double *, hipStream_t);                     \
    template void launch_weights<R>(const KParams<R> &, double, double, double *, hipStream_t);           \
    template void launch_viz<R>(const KParams<R> &, const R *, const R *, long long, float *, float *, hipStream_t);
INSTANTIATE(float)
INSTANTIATE(double)

}  // namespace mppi
