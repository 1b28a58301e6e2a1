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
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>

#include "mppi_device.h"

namespace mppi {

template <typename R> struct alignas(4 * sizeof(R)) VecT4 { R x, y, z, w; };

// Diagnostic build only (make stamps): wall-clock stamps (s_memrealtime, 10 ns ticks) of block 0 / wave 0
// at phase boundaries, written to a buffer nothing else reads.  The shipped library has no stamps.
#ifdef MPPI_STAMPS
__device__ unsigned long long g_stamps[128];  // [64 + n]: the same stamp of the LAST workgroup of the launch
#define STAMP(n)                                                                 \
    do {                                                                         \
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) g_stamps[64 + (n)] = wall_clock64(); \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                               \
            g_stamps[n] = wall_clock64();                                        \
            if ((n) < 8) g_stamps[40 + (n)] = clock64(); /* shader clock ticks */ \
            if ((n) == 16 || (n) == 21) g_stamps[32 + (n)] = clock64();           \
        }                                                                        \
    } while (0)
#else
#define STAMP(n) \
    do {         \
    } while (0)
#endif

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
    st += blockIdx.x;  // one workgroup per agent (have_x: single agent only)
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
// OBS = false: a handle without obstacles compiles to the kernel without the obstacle table and the collision tests.
// PLAIN: in addition the noise is drawn in the kernel, the rollout clamps its controls and no cost wraps the yaw -- the
// reference's diff-drive NumPy controller as bench.py runs it; the four run-time switches become constants (0.19 us per
// iteration at config 2, A/B on one box).  Anything else takes the general instantiation.
template <typename R, int MODEL, bool OBS = true, bool PLAIN = false> struct Rollout {
    __device__ __forceinline__ bool use_philox() const { return PLAIN || P.use_philox; }
    __device__ __forceinline__ bool clamp_rollout() const { return PLAIN || P.clamp_rollout; }
    __device__ __forceinline__ bool wrap_stage() const { return !PLAIN && P.wrap_stage; }
    __device__ __forceinline__ bool wrap_term() const { return !PLAIN && P.wrap_term; }
    const KParams<R> &P;
    const int k, lane, c;
    const unsigned iter;
    const bool exploit;  // k < (1-expl)*K, :116
    R cx, cy, cyaw, cvel;  // state carried from chunk to chunk of 64 steps (wave-uniform)
    int p;                 // sequential mode: the threaded waypoint index
    bool slow;             // sequential mode: some call moved the index, evolve it call by call
    bool hit_seen = false; // some step so far collided: the f32 cost sum must follow the reference's order
    bool doomed = false;   // sequential mode: a smaller sample of the workgroup moved the index too (see chunk)
    int *sh_first = nullptr;  // LDS: smallest sample of the workgroup that moved the index so far (fused kernels)
    R s_acc, s_last;
    const int n_chunk, lane_last;
    const RefPair<R> *win;  // the search window at c staged in LDS by the workgroup (or null)
    const ObsLanes<R> obs;  // obstacle table, one circle per lane
    const int agent;        // several agents per launch (blockIdx.y): offsets into u / S / pout / state, noise stream

    __device__ __forceinline__ Rollout(const KParams<R> &P_, const DevState &sv, int k_, int lane_,
                                       const RefPair<R> *win_, const ObsLanes<R> &obs_, int agent_ = 0)
        : P(P_), k(k_), lane(lane_), c(sv.c), iter((unsigned)sv.iter),
          exploit((k_ + P_.k_offset) < P_.n_exploit), cx((R)sv.x0[0]), cy((R)sv.x0[1]), cyaw((R)sv.x0[2]),
          cvel(MODEL == MODEL_RACE ? (R)sv.x0[3] : R(0)), p(sv.c), slow(false), s_acc(0), s_last(0),
          n_chunk((P_.T + 63) >> 6), lane_last((P_.T - 1) & 63), win(win_), obs(obs_), agent(agent_) {}

    // this lane's noise for step t of sample k (S1, or the caller's tensor)
    __device__ __forceinline__ void load_eps(int ch, float &e0, float &e1) const {
        const int t = ch * 64 + lane;
        e0 = 0.f;
        e1 = 0.f;
        if (t < P.T) {
            if (use_philox()) {
                px::sample(P.seed_lo, P.seed_hi, iter, (unsigned)(k + P.k_offset), t, P.chol, e0, e1,
                           (unsigned)(P.noise_stream + agent));
            } else {
                const float2 e = *reinterpret_cast<const float2 *>(eps_tensor(P, iter, agent) + ((size_t)k * P.T + t) * 2);
                e0 = e.x;
                e1 = e.y;
            }
        }
    }

    // S2-S3 of steps [64 ch, 64 ch + 64): this lane's controls and the state after its step (returns the noise it used)
    struct Step { R x, y, yaw, vel, u0, u1, v0, v1; bool act; };
    __device__ __forceinline__ Step dynamics(int ch, float &e0, float &e1) {
        const int t = ch * 64 + lane;
        const bool act = t < P.T;
        STAMP(8);
        load_eps(ch, e0, e1);
        STAMP(9);
        R u0 = 0, u1 = 0;
        if (act) {
            const R *u = P.u + (size_t)agent * 2 * P.T;
            u0 = u[2 * t];
            u1 = u[2 * t + 1];
        }
        R v0 = exploit ? u0 + (R)e0 : (R)e0, v1 = exploit ? u1 + (R)e1 : (R)e1;  // :116-119
        if (clamp_rollout()) {                                                   // `_g` :285-289
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
        STAMP(10);
        return Step{x, y, yaw, vel, u0, u1, v0, v1, act};
    }

    // steps [64 ch, 64 ch + 64) of the horizon; returns the noise it used
    __device__ __forceinline__ void chunk(int ch, float &e0, float &e1) {
        const R *__restrict__ ref = P.ref;
        const Step sp = dynamics(ch, e0, e1);
        const R x = sp.x, y = sp.y, yaw = sp.yaw, vel = sp.vel, u0 = sp.u0, u1 = sp.u1, v0 = sp.v0, v1 = sp.v1;
        const bool act = sp.act;
        // ---- waypoint index of every call in this chunk ----------------------------------
        int my_idx;
        if (P.per_rollout) {  // the threading below, for every sample on its own (no first-mover bookkeeping); a window
            slow = window_len<R>(P.window, P.n_ref, p) > 1;  // of one candidate leaves nothing to thread
            doomed = false;
        }
        if (!P.sequential && !P.per_rollout) {
            const int wlen = window_len<R>(P.window, P.n_ref, c);
            if (win) {
                const int n_act = min(64, P.T - ch * 64);
                int split = 1;  // a short tail chunk: several lanes share one step's candidates
                while (split < 16 && 2 * split * n_act <= 64) split <<= 1;
                my_idx = split > 1 ? nearest_in_window_split(win, c, wlen, x, y, split, lane)
                                   : nearest_in_window_lds(win, c, wlen, x, y);
                if (!act) my_idx = c;
            } else {
                my_idx = nearest_in_window(ref, c, wlen, x, y);
            }
        } else {
            // lane t's first minimum over the window at the incoming index: from the LDS copy of the window at c in
            // one pass while the index still is c (one broadcast read per two candidates; walking the path with
            // scalar loads costs a memory round trip per four candidates), by the scalar walk below otherwise
            int first_min = p;
            bool have_first_min = false;
            if (!slow) {  // does any call move the index away from p?
                const int wlen = window_len<R>(P.window, P.n_ref, p);
                bool trig = false;
                if (win != nullptr && p == c) {
                    first_min = nearest_in_window_lds(win, c, wlen, x, y);
                    trig = first_min != c;
                    have_first_min = true;
                } else {
                    const R d0 = dist2(ref, p, x, y);
#pragma unroll 4
                    for (int j = 1; j < wlen; ++j) trig |= dist2(ref, p + j, x, y) < d0;
                }
                slow = __ballot(trig && act) != 0ull;
                if (slow && sh_first) {
                    // Only the FIRST sample that moves the index keeps its result this round (the later ones are
                    // rolled out again from the index it leaves).  A wave that finds a smaller sample of its own
                    // workgroup registered skips the threading below: while the robot travels every sample
                    // moves the index, and with every wave threading its own a launch took 18 us instead of 7.
                    int old = 0;
                    if (lane == 0) old = atomicMin(sh_first, k);
                    doomed = wv::read_lane(old, 0) < k;
                }
            }
            my_idx = p;
            if (slow && !doomed) {  // thread the index through this chunk's calls in order
                // All calls at once first: lane t's first minimum over the window at the incoming index p0.  While
                // that window reaches the path's end (or the index has not moved yet) the window of a later call
                // [p, ..) is a suffix of it, so a first minimum at or beyond p IS that call's answer; only a
                // minimum behind p needs the search proper.  (One dependent wave-wide search per call cost the
                // wave that moves the index 14 us at config 2 -- the whole launch waited for it.)
                const int p0 = p, w0 = window_len<R>(P.window, P.n_ref, p0);
                const bool to_end = p0 + w0 >= P.n_ref;
                if (!have_first_min) {
                    R best = dist2(ref, p0, x, y);
#pragma unroll 4
                    for (int j = 1; j < w0; ++j) {
                        const R d = dist2(ref, p0 + j, x, y);
                        if (d < best) { best = d; first_min = p0 + j; }
                    }
                }
                const int n_act = min(64, P.T - ch * 64);
                for (int tt = 0; tt < n_act; ++tt) {
                    const int a = wv::read_lane(first_min, tt);
                    if (a >= p && (to_end || p == p0)) {
                        p = a;
                    } else {
                        const R xt = wv::read_lane(x, tt), yt = wv::read_lane(y, tt);
                        p = nearest_uniform(ref, p, window_len<R>(P.window, P.n_ref, p), xt, yt, lane);
                    }
                    if (lane == tt) my_idx = p;
                }
            }
        }

        STAMP(11);
        costs(ch, sp, my_idx, false, 0);
    }

    // stage cost of every call of chunk `ch` (only the last one survives when !accumulate) and, in the last chunk, the
    // terminal cost.  `my_idx`: this lane's call's waypoint index; `have_term`: the terminal call's index is `idx_term_in`
    // (the caller threaded it), else it is taken here (one more search in the threading modes, :244)
    __device__ __forceinline__ void costs(int ch, const Step &sp, int my_idx, bool have_term, int idx_term_in) {
        const R *__restrict__ ref = P.ref;
        const R x = sp.x, y = sp.y, yaw = sp.yaw, vel = sp.vel, u0 = sp.u0, u1 = sp.u1, v0 = sp.v0, v1 = sp.v1;
        const bool act = sp.act;
        // ---- stage cost of every call (only the last one survives when !accumulate) ------
        const bool last_chunk = ch == n_chunk - 1;
        if (P.accumulate || last_chunk) {
            const bool hit = OBS ? collided<MODEL == MODEL_RACE>(P, x, y, yaw, obs) : false;
            R st_c = tracking_cost<R, MODEL>(P, P.ws, wrap_stage(), my_idx, x, y, yaw, vel);
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
                    // a sample that has collided is added up in the reference's order (s_acc is wave-uniform
                    // here).  Without a penalty in the sum the order moves the last ulps only (1e-7 relative,
                    // like the rest of the f32 arithmetic) and a wave reduction does: ~15 instructions
                    // instead of 2 per step, 15 % of the race-car launch at T = 75.
                    hit_seen |= __ballot(act && hit) != 0ull;
                    if (hit_seen) s_acc = wv::ordered_sum(s_acc, stage, 0, min(64, P.T - ch * 64));
                    else s_acc += wv::reduce<wv::OpAdd>(act ? stage : R(0));
                } else {
                    s_acc += act ? stage : R(0);
                }
            }
            if (last_chunk) {
                // terminal call: same state; the sequential index takes one more step (:244)
                int idx_term = my_idx;
                if (have_term) {
                    idx_term = idx_term_in;
                } else if ((P.sequential || P.per_rollout) && slow && !doomed) {
                    const R xt = wv::read_lane(x, lane_last), yt = wv::read_lane(y, lane_last);
                    p = nearest_uniform(ref, p, window_len<R>(P.window, P.n_ref, p), xt, yt, lane);
                    idx_term = p;
                }
                R term = tracking_cost<R, MODEL>(P, P.wt, wrap_term(), idx_term, x, y, yaw, vel);
                if (hit) term += P.penalty;
                s_last = P.accumulate ? term : stage + term;
            }
        }
    }

    // S[k] once every chunk has run; also publishes the index this sample leaves behind
    // `publish`: register this sample in first_k if it moved the index.  The fused kernels pass false and register
    // the smallest such sample of the WORKGROUP instead (publish_first_mover): while the robot travels every sample
    // moves the index, and 4096 atomics on one address kept a launch busy for ~15 us after its last wave.
    __device__ __forceinline__ R finish(bool publish = true) {
        R total = wv::read_lane(s_last, lane_last);
        if (P.accumulate) total = (sizeof(R) == 4 ? s_acc : wv::reduce<wv::OpAdd>(s_acc)) + total;
        if (lane == 0) {
            P.S[(size_t)agent * P.K + k] = total;
            P.pout[(size_t)agent * P.K + k] = p;
            if (publish && moved()) atomicMin(&(P.st + agent)->first_k, k);
        }
        return total;
    }
    __device__ __forceinline__ bool moved() const { return P.sequential && p != c; }
};

// Any horizon: S[k] only (the softmin partials come from k_reduce).
template <typename R, int MODEL>
__global__ __launch_bounds__(256) void k_rollout(const DevState *st_pre, const KParams<R> P) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // wave-uniform
    const DevState sv = load_state(P, st_pre);
    __shared__ RefPair<R> sh_win[WINDOW_LDS_MAX / 2];
    const ObsLanes<R> obs = load_obstacles(P, lane);
    const int wlen0 = window_len<R>(P.window, P.n_ref, sv.c);
    const bool use_win = wlen0 <= WINDOW_LDS_MAX;  // (both waypoint modes search the window at c first)
    if (use_win) stage_window(sh_win, P.ref, sv.c, wlen0, (int)threadIdx.x, (int)blockDim.x);
    __syncthreads();
    if (k >= P.K || k < sv.k_start) return;
    Rollout<R, MODEL> r(P, sv, k, lane, use_win ? sh_win : nullptr, obs);
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

// one atomic per workgroup: the smallest of its samples that moved the waypoint index, if any (one thread calls)
template <int N> __device__ __forceinline__ void publish_first_mover(const int *sh_mover, int *first_k) {
    int m = NO_TRIGGER;
#pragma unroll
    for (int q = 0; q < N; ++q) m = min(m, sh_mover[q]);
    if (m != NO_TRIGGER) atomicMin(first_k, m);
}

// ------------------------------------------------------------------------------------------
// The sequential waypoint index in ONE launch (LB_CAND in mppi_kernels.h; mppi_differential_drive.py:201-249).
// Pass A: every lane counts the strict descents of its call's distances over the LB_CAND candidates behind c and checks
// that none follows a non-descent (then the count IS the first nearest candidate m, and a search entered at any p
// returns max(p, m)); the wave's and the workgroup's maxima follow, the workgroup publishes its maximum in its slot.
// Look-back: wave 0 waits for the slots of all workgroups before its own (they were dispatched earlier and wait for
// nothing behind them) and takes their maximum E -- the offset this workgroup is entered at.  Meanwhile every wave
// prices its sample under every offset (lanes over the offset).  Pass B: sample w's index is max(E, the samples before it,
// its own) -- `S[k] =` semantics only (:124): the cost needs the index of the last stage call and of the terminal call,
// which searches from the same state and so stays where the last stage call left it.
// A call that is not unimodal, an index that leaves the candidates' reach or a wait that times out raises the slot's
// `bad` bit: k_finalize hands the iteration to the speculation rounds.  Returns S_k; lane 0 stores S and pout.
// ------------------------------------------------------------------------------------------
// Look-back (one wave): waits until the words of workgroups [0, b) carry this iteration's tag; E = the largest offset among
// them, bad = one of them is marked.  Four words per lane and 16-byte load, copy b mod LB_COPIES of the words; volatile =
// loads that bypass this XCD's L2 (the words come from the other XCDs' workgroups).  Returns false when the wait timed out
// (never seen: the iteration is then redone by the speculation rounds).
__device__ __forceinline__ bool lb_wait(const unsigned *slots, int b, unsigned tag, int lane, int &E, bool &bad) {
    static_assert(HYP_MAX_BLOCKS <= 512, "two 16-byte loads per lane cover the words");
    typedef unsigned lb_u4 __attribute__((ext_vector_type(4)));
    const volatile __attribute__((address_space(1))) lb_u4 *src =
        (const volatile __attribute__((address_space(1))) lb_u4 *)(slots + (b & (LB_COPIES - 1)) * LB_COPY_STRIDE);
    E = 0;
    bad = false;
    if (b <= 0) return true;
    unsigned long long t0 = 0ull;
    for (int n = 0;; ++n) {
        bool ready = true, bd = false;
        int mx = 0;
#pragma unroll
        for (int i = 0; i < HYP_MAX_BLOCKS / 256; ++i) {
            const int first = 4 * (lane + 64 * i);
            if (first < b) {
                const lb_u4 w4 = src[lane + 64 * i];
                const unsigned w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (first + j < b) {
                        ready &= (w[j] & LB_TAG_MASK) == tag;
                        bd |= (w[j] & LB_BAD) != 0u;
                        mx = max(mx, (int)(w[j] & (LB_BAD - 1)));
                    }
                }
            }
        }
        if (__ballot(!ready) == 0ull) {
            E = wv::reduce<wv::OpMaxInt>(mx);
            bad = __ballot(bd) != 0ull;
            return true;
        }
        if ((n & 15) == 15) {  // (the clock is a memory round trip of its own: looked at every 16th poll only)
            const unsigned long long now = wall_clock64();
            if (t0 == 0ull) t0 = now;
            if (now - t0 > LB_TIMEOUT_TICKS) return false;
        }
    }
}
// the word of workgroup b into every copy (lanes < LB_COPIES of one wave)
__device__ __forceinline__ void lb_publish(unsigned *slots, int b, unsigned word, int lane) {
    if (lane < LB_COPIES) __hip_atomic_store(&slots[b + lane * LB_COPY_STRIDE], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the offsets a workgroup's calls were entered at stay within the candidates' reach (see LB_CAND)
__device__ __forceinline__ bool lb_reach(int leave, int window, int n_ref, int c) {
    return leave < window && (leave + window <= LB_CAND || n_ref - c <= LB_CAND);
}

// pass A of one pair of candidates {x_2q, x_2q+1, y_2q, y_2q+1}: does this lane's distance fall at the first / the second
// of them (f32: two packed subtractions, a packed product and a packed fma)
typedef float lb_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lb_pair(const RefPair<float> &rp, float x, float y, float &prev, bool &g0, bool &g1) {
    const lb_f2 dx = lb_f2{x, x} - lb_f2{rp.x0, rp.x1}, dy = lb_f2{y, y} - lb_f2{rp.y0, rp.y1};
    const lb_f2 d = dx * dx + dy * dy;
    g0 = d.x < prev;
    g1 = d.y < d.x;
    prev = d.y;
}
__device__ __forceinline__ void lb_pair(const RefPair<double> &rp, double x, double y, double &prev, bool &g0, bool &g1) {
    const double dx0 = x - rp.x0, dx1 = x - rp.x1, dy0 = y - rp.y0, dy1 = y - rp.y1;
    const double d0 = dx0 * dx0 + dy0 * dy0, d1 = dx1 * dx1 + dy1 * dy1;
    g0 = d0 < prev;
    g1 = d1 < d0;
    prev = d1;
}

// Pass A for the NP positions a lane owns (an idle one: x = NaN, it never descends): m[i] = this lane's count of descents
// over the LB_CAND candidates of sh_c, `bad` = a call of this lane whose descents do not all come first.  Per candidate a
// comparison and ONE add-with-carry per lane: the comparisons are shifted into a word, first candidate in the top bit
// (w <- w + w + g), so that the count is a population count and "the descents come first" reads w == 1..10..0.  (Nothing
// on the scalar unit: lane masks per candidate had the compiler spill SGPRs into VGPR lanes in the two-samples-per-wave
// kernel, 330 lane moves per wave.)
template <int NP, typename R>
__device__ __forceinline__ void lb_scan(const RefPair<R> *sh_c, const R (&x)[NP], const R (&y)[NP], int (&m)[NP], bool &bad) {
    static_assert(LB_CAND == 32, "one bit per candidate");
    unsigned w[NP];
    R prev[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        w[i] = 0u;
        prev[i] = R(INFINITY);  // (candidate 0 is compared against +inf: the top bit, counted off below)
    }
#pragma unroll
    for (int q0 = 0; q0 < LB_CAND / 2; q0 += 4) {
        RefPair<R> rp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rp[j] = sh_c[q0 + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                bool g0, g1;
                lb_pair(rp[j], x[i], y[i], prev[i], g0, g1);
                w[i] = w[i] + w[i] + (g0 ? 1u : 0u);
                w[i] = w[i] + w[i] + (g1 ? 1u : 0u);
            }
        }
    }
    bad = false;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int pc = __popc(w[i]);
        m[i] = pc - 1;
        // (an idle position: w = 0, nothing to check)
        bad |= pc > 0 && w[i] != (0xffffffffu << (32 - pc));
    }
}

// Returns, on lane l, S of the workgroup's sample l & 15 (every wave computes all sixteen); lane 0 stores S_k and pout.
template <typename R, bool OBS, bool PLAIN>
__device__ __forceinline__ R fused_lookback(const KParams<R> &P, const DevState &sv, int k, bool valid_in, float &e0, float &e1) {
    __shared__ RefPair<R> sh_c[LB_CAND / 2];
    __shared__ VecT4<R> sh_row[LB_CAND];          // the candidates' rows {x, y, yaw, v}: what the costs are taken against
    __shared__ R sh_cost[FUSED_WAVES][LB_CAND];   // sample w priced under offset j
    __shared__ int sh_M[FUSED_WAVES];
    __shared__ int sh_E;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool valid = __builtin_amdgcn_readfirstlane((int)valid_in) != 0;  // (wave-uniform: said so that the counting below stays scalar)
    const int c = sv.c, b = blockIdx.x;
    const int nc = min(LB_CAND, P.n_ref - c);  // (> 1: the caller's condition)
    const unsigned tag = lb_tag(P.lb_seq);
    // thread j < LB_CAND fetches candidate j's row (an absent one: far away, it never descends); the loads are in flight
    // during the draw
    VecT4<R> mine{R(1e30), R(1e30), R(0), R(0)};
    if ((int)threadIdx.x < nc) mine = *reinterpret_cast<const VecT4<R> *>(P.ref + 4 * (c + (int)threadIdx.x));
    const ObsLanes<R> obs = OBS ? load_obstacles(P, lane) : ObsLanes<R>{R(0), R(0), R(0)};
    Rollout<R, MODEL_DIFF, OBS, PLAIN> r(P, sv, k, lane, nullptr, obs, 0);
    typename Rollout<R, MODEL_DIFF, OBS, PLAIN>::Step sp{R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0), false};
    if (valid) sp = r.dynamics(0, e0, e1);
    if (threadIdx.x < LB_CAND) {
        const int j = threadIdx.x;
        sh_row[j] = mine;
        R *pair = reinterpret_cast<R *>(&sh_c[j >> 1]);  // {x_2q, x_2q+1, y_2q, y_2q+1}
        pair[j & 1] = mine.x;
        pair[2 + (j & 1)] = mine.y;
    }
    __syncthreads();
    int Mw = 0;
    bool bad_w = false;
    if (valid) {
        // ---- pass A: this lane's call (the state after step `lane`) against the candidates (lb_scan) ---------------
        const R xs[1] = {sp.act ? sp.x : R(NAN)}, ys[1] = {sp.y};
        int m[1];
        bool bad;
        lb_scan<1, R>(sh_c, xs, ys, m, bad);
        if (!sp.act) m[0] = 0;
        Mw = wv::reduce<wv::OpMaxInt>(m[0]);
        bad_w = __ballot(bad && sp.act) != 0ull;
    }
    STAMP(11);
    if (lane == 0) sh_M[wid] = Mw | (bad_w ? LB_BAD : 0);
    __syncthreads();
    STAMP(12);
    // the workgroup's maximum and, for wave w, the maximum of the samples before it (one DPP row of 16 values)
    static_assert(FUSED_WAVES == 16, "one DPP row of maxima");
    const int mv = sh_M[lane & 15];
    const int incl = wv::scan_incl_row<wv::OpMaxInt>(mv & (LB_BAD - 1));
    const int Mb = wv::read_lane(incl, 15);
    const bool bad_b = (__ballot((mv & LB_BAD) != 0) & 0xffffull) != 0ull;
    unsigned *slots = P.hyp_slots;
    // (LB_COPIES copies of the words, LB_COPY_STRIDE apart: 256 workgroups polling one kilobyte queue up at one memory
    // channel -- workgroup b polls copy b mod LB_COPIES, every workgroup writes all of them, k_finalize reads copy 0)
    if (wid == 0) lb_publish(slots, b, tag | (bad_b ? (unsigned)LB_BAD : 0u) | (unsigned)Mb, lane);
    // ---- this sample priced under every offset, lanes over the offset (wave 0: while its word travels) -----------
    {
        R cost_j = R(INFINITY);
        if (valid) {  // (every lane takes part in the ballot and the lane reads; lanes >= LB_CAND price the last candidate again)
            const int ll = r.lane_last;
            const R xT = wv::read_lane(sp.x, ll), yT = wv::read_lane(sp.y, ll), yawT = wv::read_lane(sp.yaw, ll);
            const R ctrl = (sp.u0 * P.sinv[0] + sp.u1 * P.sinv[2]) * sp.v0 + (sp.u0 * P.sinv[1] + sp.u1 * P.sinv[3]) * sp.v1;  // :124
            const R ctrlT = wv::read_lane(ctrl, ll);
            const bool hit_lane = OBS ? collided<false>(P, sp.x, sp.y, sp.yaw, obs) : false;
            const bool hitT = OBS ? ((__ballot(hit_lane) >> ll) & 1ull) != 0ull : false;
            const VecT4<R> row = sh_row[min(lane, nc - 1)];
            const R rr[4] = {row.x, row.y, row.z, row.w};
            R st_c = tracking_cost_row<R, MODEL_DIFF>(P, P.ws, r.wrap_stage(), rr, xT, yT, yawT, R(0));
            if (hitT) st_c += P.penalty;
            const R stage = st_c + P.gamma * ctrlT;
            R term = tracking_cost_row<R, MODEL_DIFF>(P, P.wt, r.wrap_term(), rr, xT, yT, yawT, R(0));
            if (hitT) term += P.penalty;
            cost_j = stage + term;
        }
        if (lane < LB_CAND) sh_cost[wid][lane] = cost_j;
    }
    STAMP(14);
    if (wid == 0) {
        // ---- look-back: the words of the workgroups before this one ----------------------------------------------
        int E = 0;
        bool bad_e = false;
        if (!lb_wait(slots, b, tag, lane, E, bad_e)) bad_e = true;
        const bool reach = lb_reach(max(E, Mb), P.window, P.n_ref, c);
        // (what a later workgroup reads of this word is the offset; the bad bit is k_finalize's: copy 0)
        if ((bad_e || !reach) && !bad_b && lane == 0) __hip_atomic_fetch_or(&slots[b], (unsigned)LB_BAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) sh_E = E;
        STAMP(13);
    }
    __syncthreads();
    STAMP(15);
    // ---- pass B: every wave prices all sixteen samples (lane l: sample l & 15) -- no second exchange of costs -------
    const int E = sh_E;
    const int w15 = lane & 15;
    const int before_l = wv::dpp<wv::DPP_ROW_SHR1>(incl, 0);  // the samples before sample l & 15 (row-wise shift; 0 for the first)
    const int a = min(max(max(E, before_l), mv & (LB_BAD - 1)), nc - 1);
    const R S_l = sh_cost[w15][a];
    if (valid && lane == wid) {  // (lane wid holds this wave's own sample)
        P.S[k] = S_l;
        P.pout[k] = c + a;
    }
    return S_l;
}

// ------------------------------------------------------------------------------------------
// MPPI_WAYPOINT_PER_ROLLOUT, T <= 64: the index of every cost call when it threads through each sample's OWN calls
// (mppi_differential_drive.py:228,:244) and restarts from the x0 call's index at every sample.  The T + 1 searches of a
// sample are a dependent chain (call t searches from where call t-1 ended), so the lanes-over-the-horizon layout has
// nothing to offer it: threaded by the sample's own wave (Rollout::chunk's serial loop, one wave-wide search per call)
// it cost 33 us per iteration at config 2 against 12 us with a frozen index.  Here every wave leaves its 64 positions in
// LDS and FOUR waves (one per SIMD) thread the indices of four samples each, a row of 16 lanes per sample over the
// candidates of a window: per call one distance per lane (two for windows of up to 32), the row's first minimum by DPP
// row rotations, the index advanced -- the four chains of a wave run in lockstep and the four waves side by side.
// Out: sh_ix[sample][t] = the index call t used (t = T: the terminal call).
// ------------------------------------------------------------------------------------------
constexpr int PR_REF_LDS = 512;  // waypoints of the path kept in LDS for the threading (longer paths: read through the cache)
template <typename R> __device__ __forceinline__ R row_all_min(R x) {  // min over the 16 lanes of each DPP row, in every lane
    x = fmin(x, wv::dpp<0x128>(x, x));  // row_ror:8
    x = fmin(x, wv::dpp<0x124>(x, x));
    x = fmin(x, wv::dpp<0x122>(x, x));
    x = fmin(x, wv::dpp<0x121>(x, x));
    return x;
}
__device__ __forceinline__ int row_all_min(int x) {
    x = min(x, wv::dpp<0x128>(x, x));
    x = min(x, wv::dpp<0x124>(x, x));
    x = min(x, wv::dpp<0x122>(x, x));
    x = min(x, wv::dpp<0x121>(x, x));
    return x;
}
template <typename R>
__device__ __forceinline__ void per_rollout_thread(const KParams<R> &P, int c, const R (*sh_x)[64], const R (*sh_y)[64],
                                                   const R *sh_ref /* [n][2] or null */, short (*sh_ix)[66]) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (wid >= 4) return;
    const int s = 4 * wid + (lane >> 4), sub = lane & 15, T = P.T;
    int p = c;
    for (int t = 0; t <= T; ++t) {  // t == T: the terminal call, on the state the last stage call saw (:244)
        const int ts = t < T ? t : T - 1;
        const R x = sh_x[s][ts], y = sh_y[s][ts];
        const int wl = min(P.window, P.n_ref - p);
        R best = R(INFINITY);
        int bj = INT_MAX;
        for (int j = sub; j < wl; j += 16) {
            R rx, ry;
            if (sh_ref) { rx = sh_ref[2 * (p + j)]; ry = sh_ref[2 * (p + j) + 1]; }
            else { rx = P.ref[4 * (p + j)]; ry = P.ref[4 * (p + j) + 1]; }
            const R dx = x - rx, dy = y - ry;
            const R d = dx * dx + dy * dy;
            if (d < best) { best = d; bj = j; }
        }
        const R m = row_all_min(best);
        p += row_all_min(best == m ? bj : INT_MAX);  // first minimum: the smallest offset among the nearest
        if (sub == 0) sh_ix[s][t] = (short)p;
    }
}

// MULTI: several agents per launch, one row of workgroups (blockIdx.y) each; a single agent compiles to the
// offset-free code (the offsets cost config 2 half a microsecond per iteration when they were unconditional)
// SPEC: 0 general, 1 no obstacles, 2 no obstacles + the PLAIN switches (see Rollout)
// HYPK: the instantiation that can resolve the sequential index in one launch (fused_lookback).  A separate instantiation,
// picked by the host while the waypoint index can still move (KParams::hyp): carrying that code costs the lean kernel
// 0.17 us per launch in registers and LDS (A/B on one box), and at the end of the path nothing moves any more.
template <typename R, int MODEL, int NCH, bool MULTI, int SPEC, bool HYPK = false>
__global__ __launch_bounds__(64 * FUSED_WAVES) void k_rollout_fused(const DevState *st_pre, const KParams<R> P,
                                                                    R *__restrict__ partials) {
    constexpr bool OBS = SPEC == 0, PLAIN = SPEC == 2;
    const int agent = MULTI ? (int)blockIdx.y : 0;
    __shared__ R sh_S[FUSED_WAVES];
    __shared__ R sh_e[FUSED_WAVES];
    __shared__ R sh_acc[FUSED_WAVES][128 * NCH];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int k = blockIdx.x * FUSED_WAVES + wid;  // wave-uniform
    STAMP(0);
    const DevState sv = load_state(P, st_pre + agent);
    // (a workgroup whose samples are all final -- below k_start in a repair round -- rebuilds the same record from
    // the stored costs; an early exit here would keep the compiler from fetching the kernel arguments up front)
    const int k_start = sv.k_start;
    STAMP(1);
    const bool valid = k < P.K;
    const int wlen0 = window_len<R>(P.window, P.n_ref, sv.c);
    // HYPK, the first round of an iteration whose waypoint index can move: resolved in this launch (fused_lookback)
    const bool lookback = HYPK && sv.round == 0 && wlen0 > 1;
    __shared__ RefPair<R> sh_win[WINDOW_LDS_MAX / 2];
    const ObsLanes<R> obs = OBS ? load_obstacles(P, lane) : ObsLanes<R>{R(0), R(0), R(0)};
    // Both waypoint modes search the window at c first.  A window of ONE candidate (the robot holds the end of the
    // path) leaves nothing to search or to move: no staging and no barrier behind its loads then (0.7 us per launch
    // at config 2).
    const bool seq_search = !lookback && P.sequential && wlen0 > 1;
    const bool use_win = !lookback && wlen0 > 1 && wlen0 <= WINDOW_LDS_MAX;  // (a window of one candidate: nothing to stage or to search)
    __shared__ int sh_mover[FUSED_WAVES];
    __shared__ int sh_first;
    if (use_win) stage_window(sh_win, P.ref, sv.c, wlen0, (int)threadIdx.x, (int)blockDim.x);
    if (seq_search && threadIdx.x == 0) sh_first = NO_TRIGGER;
    if (use_win || seq_search) __syncthreads();
    float e0[NCH], e1[NCH];
    R S_k = R(INFINITY);
    int mover = NO_TRIGGER;  // this wave's sample if it moved the waypoint index (sequential mode)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) { e0[ch] = 0.f; e1[ch] = 0.f; }
    // MPPI_WAYPOINT_PER_ROLLOUT on horizons of one chunk: positions -> LDS, four waves thread the indices, costs (see
    // per_rollout_thread); a window of one candidate leaves nothing to thread
    const bool pr_fast = !PLAIN && NCH == 1 && P.per_rollout && wlen0 > 1 && P.n_ref < 32768;
    R S_all = R(INFINITY);  // (look-back: every wave holds all sixteen costs, sample l & 15 on lane l)
    if (lookback) {
        if constexpr (HYPK && MODEL == MODEL_DIFF && NCH == 1) S_all = fused_lookback<R, OBS, PLAIN>(P, sv, k, valid, e0[0], e1[0]);
        S_k = wv::read_lane(S_all, wid);
    } else if (pr_fast) {
        __shared__ R sh_px[FUSED_WAVES][64], sh_py[FUSED_WAVES][64];
        __shared__ short sh_ix[FUSED_WAVES][66];
        __shared__ R sh_pref[2 * PR_REF_LDS];
        const bool ref_lds = P.n_ref <= PR_REF_LDS;
        if (ref_lds)
            for (int i = threadIdx.x; i < P.n_ref; i += blockDim.x) { sh_pref[2 * i] = P.ref[4 * i]; sh_pref[2 * i + 1] = P.ref[4 * i + 1]; }
        Rollout<R, MODEL, OBS, PLAIN> r(P, sv, k, lane, nullptr, obs, agent);
        typename Rollout<R, MODEL, OBS, PLAIN>::Step sp{R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0), false};
        if (valid) sp = r.dynamics(0, e0[0], e1[0]);
        sh_px[wid][lane] = sp.x;
        sh_py[wid][lane] = sp.y;
        __syncthreads();
        per_rollout_thread<R>(P, sv.c, sh_px, sh_py, ref_lds ? sh_pref : nullptr, sh_ix);
        __syncthreads();
        if (valid) {
            r.costs(0, sp, (int)sh_ix[wid][lane < P.T ? lane : 0], true, (int)sh_ix[wid][P.T]);
            S_k = r.finish(false);
        }
    } else if (valid) {
        Rollout<R, MODEL, OBS, PLAIN> r(P, sv, k, lane, use_win ? sh_win : nullptr, obs, agent);
        r.sh_first = seq_search ? &sh_first : nullptr;
        if (k >= k_start) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                if (ch < r.n_chunk) r.chunk(ch, e0[ch], e1[ch]);
            S_k = r.finish(false);
            if (r.moved()) mover = k;
        } else {
            S_k = P.S[(size_t)agent * P.K + k];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) r.load_eps(ch, e0[ch], e1[ch]);
        }
    }
    STAMP(2);
    R sv_l = S_all;
    if (!lookback) {
        if (lane == 0) {
            sh_S[wid] = S_k;
            if (seq_search) sh_mover[wid] = mover;
        }
        __syncthreads();
        sv_l = sh_S[lane & 15];
    }
    STAMP(3);
    // rho_b = min over the workgroup's 16 costs: lane w of a row reads sample w's, four DPP steps fold the row (every wave
    // needs it: 16 reads + 15 minima per wave took a twelfth of the launch's VALU instructions)
    static_assert(FUSED_WAVES == 16, "one DPP row of costs");
    const R rho = wv::read_lane(wv::scan_incl_row<wv::OpMin>(sv_l), 15);
    // (samples whose cost carries a collision penalty, mppi_stats::n_collided: one ballot over the costs read above)
    const int n_hit_i = OBS ? __popcll(__ballot(sv_l >= P.penalty && sv_l < R(INFINITY)) & 0xffffull) : 0;
    const R e = valid ? mf::exp_(-P.beta * (S_k - rho)) : R(0);  // :175
    if (lane == 0) sh_e[wid] = e;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        sh_acc[wid][ch * 128 + 2 * lane] = e * (R)e0[ch];
        sh_acc[wid][ch * 128 + 2 * lane + 1] = e * (R)e1[ch];
    }
    __syncthreads();
    const size_t slot = (size_t)agent * P.slots + blockIdx.x;  // this workgroup's record
    R *out = partials + slot * record_len(P.T, (int)sizeof(R));
    for (int i = threadIdx.x; i < 2 * P.T; i += blockDim.x) {  // W_b[t] = sum_k e_k eps[k, t], :132-135
        R s = 0;
#pragma unroll
        for (int w = 0; w < FUSED_WAVES; ++w) s += sh_acc[w][i];
        out[4 + i] = s;
    }
    if (threadIdx.x == 64 * (FUSED_WAVES - 1)) {  // a lane of the last wave: the first ones carry the column sums
        // (the head's fourth word: how many of the workgroup's samples carry a collision penalty in their cost -- mppi_stats::n_collided)
        R eta = 0, eta2 = 0;
        const R n_hit = (R)n_hit_i;
#pragma unroll
        for (int w = 0; w < FUSED_WAVES; ++w) {
            const R ew = sh_e[w];
            eta += ew;
            eta2 += ew * ew;
        }
        out[0] = rho;
        out[1] = eta;
        out[2] = eta2;
        *reinterpret_cast<VecT4<R> *>(P.heads + 4 * slot) = VecT4<R>{rho, eta, eta2, n_hit};
        if (seq_search) publish_first_mover<FUSED_WAVES>(sh_mover, &(P.st + agent)->first_k);
    }
    STAMP(4);
}

// ------------------------------------------------------------------------------------------
// T <= 64: two samples per wave, two steps per lane.  Each 32-lane half of a wave owns one sample and
// lane l of the half owns steps 2l and 2l+1, so ONE Philox4x32 call feeds both of its steps (words r0,r1
// and r2,r3 -- the sampler's layout), the scans are 5 DPP steps over lane totals plus one add, and a
// workgroup leaves one record per 32 samples (half as many for k_finalize to merge).  Same arithmetic per
// step as Rollout::chunk; only the association of the prefix sums differs.
// ------------------------------------------------------------------------------------------
// SPW = samples per wave.  2: the layout above (T <= 64).  1: one sample per wave with two steps per lane, T <= 128 --
// for 64 < T <= 128 this replaces two 64-step chunks of k_rollout_fused (the second one mostly idle lanes: T = 75 runs
// 11 of 64) by one pass, ~40 % fewer instructions per sample.
constexpr int DUAL_WAVES = 16, DUAL_SAMPLES = 2 * DUAL_WAVES;

// SEQ = samples each (half-)wave rolls out one after the other (1 or 2).  Past one workgroup per CU a second wave of
// workgroups only repeats the prologue, the block reduction and the record: with SEQ = 2 the same waves do both
// samples and the launch leaves half as many records for k_finalize to merge.
// PLAIN: the noise is drawn in the kernel, the rollout clamps its controls and the yaw is wrapped in the costs exactly
// when the model is the race car -- the reference's controllers as they come; those run-time switches become constants
// (config 3: 16.2 -> 15.5 us per iteration in the hold phase, config 4 shard: 20.6 -> 20.2 us).
// Batched agents (MULTI) bring thousands of workgroups per launch: the f32 diff-drive instantiation is held to 64 VGPRs (8
// waves per SIMD) so that TWO workgroups share a CU and one's state wait, barrier and record stores run under the other's
// arithmetic -- 32 agents of K = 4096: 1.0e11 -> 1.2e11 trajectory-steps/s.  (The race car would spill to scratch.)
// LB: the instantiation that can resolve the sequential waypoint index in this launch (the look-back of fused_lookback for
// two samples per wave and two steps per lane: KParams::hyp; one pass, one agent) -- picked by the host while the index
// can still move, like k_rollout_fused's HYPK.
template <typename R, int MODEL, int SPW, bool MULTI, int SEQ, bool PLAIN, bool LB = false>
__global__ __launch_bounds__(64 * DUAL_WAVES, (sizeof(R) == 4 && MODEL == MODEL_DIFF && SPW == 2 && (MULTI || SEQ == 1)) ? 8 : 1) void k_rollout_dual(const DevState *st_pre, const KParams<R> P,
                                                                  R *__restrict__ partials) {
    const int agent = MULTI ? (int)blockIdx.y : 0;  // several agents per launch (see k_rollout_fused)
    const bool use_philox = PLAIN || P.use_philox, clamp_rollout = PLAIN || P.clamp_rollout;
    const bool wrap_stage = PLAIN ? MODEL == MODEL_RACE : (bool)P.wrap_stage, wrap_term = PLAIN ? MODEL == MODEL_RACE : (bool)P.wrap_term;
    const R *__restrict__ u_ = P.u + (size_t)agent * 2 * P.T;
    R *__restrict__ S_ = P.S + (size_t)agent * P.K;
    int *__restrict__ pout_ = P.pout + (size_t)agent * P.K;
    // lanes per sample; samples in flight (one row of sh_acc each); samples per workgroup
    constexpr int HL = 64 / SPW, ROWS = DUAL_WAVES * SPW, SAMPLES = ROWS * SEQ;
    __shared__ R sh_S[SAMPLES];
    __shared__ R sh_e[SAMPLES];
    __shared__ __attribute__((aligned(16))) R sh_acc[ROWS][4 * HL];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, h = SPW == 2 ? lane >> 5 : 0, l32 = lane & (HL - 1);
    STAMP(0);
    const DevState sv = load_state(P, st_pre + agent);
    const int k_start = sv.k_start;  // (no early exit for all-final workgroups, see k_rollout_fused)
    STAMP(1);
    const int c = sv.c;
    const unsigned iter = (unsigned)sv.iter;
    const R *__restrict__ ref = P.ref;
    const int T = P.T, t0 = 2 * l32, t1 = t0 + 1;
    const bool a0 = t0 < T, a1 = t1 < T;
    __shared__ RefPair<R> sh_win[WINDOW_LDS_MAX / 2];
    const ObsLanes<R> obs = load_obstacles(P, lane);
    const int wlen0 = window_len<R>(P.window, P.n_ref, c);
    // LB, the first round of an iteration whose waypoint index can move: resolved in this launch (see fused_lookback)
    constexpr bool LBK = LB && MODEL == MODEL_DIFF && SPW == 2 && SEQ == 1 && !MULTI;
    const bool lookback = LBK && P.hyp && sv.round == 0 && wlen0 > 1;
    __shared__ RefPair<R> sh_c[LBK ? LB_CAND / 2 : 1];
    __shared__ VecT4<R> sh_row[LBK ? LB_CAND : 1];
    __shared__ R sh_cost[LBK ? ROWS : 1][LBK ? LB_CAND : 1];
    __shared__ int sh_M[LBK ? ROWS : 1];
    __shared__ int sh_E;
    const int nc = min(LB_CAND, P.n_ref - c);
    VecT4<R> lb_mine{R(1e30), R(1e30), R(0), R(0)};  // thread j < LB_CAND: candidate j's row (absent: far away), in flight during the draw
    if (LBK && lookback && (int)threadIdx.x < nc) lb_mine = *reinterpret_cast<const VecT4<R> *>(ref + 4 * (c + (int)threadIdx.x));
    const bool seq_search = !lookback && P.sequential && wlen0 > 1;  // (see k_rollout_fused)
    const bool use_win = !lookback && wlen0 > 1 && wlen0 <= WINDOW_LDS_MAX;  // (a window of one candidate: nothing to stage or to search)
    __shared__ int sh_first;  // smallest sample of the workgroup that moved the index so far (see Rollout::chunk)
    if (use_win) stage_window(sh_win, ref, c, wlen0, (int)threadIdx.x, (int)blockDim.x);
    if (seq_search && threadIdx.x == 0) sh_first = NO_TRIGGER;
    if (use_win || seq_search) __syncthreads();

    float e00 = 0.f, e01 = 0.f, e10 = 0.f, e11 = 0.f;  // e<step><channel> of the sample in hand
    R S_k = R(INFINITY);
    bool valid = false;
    float f00 = 0.f, f01 = 0.f, f10 = 0.f, f11 = 0.f;  // the same of the first pass (SEQ == 2)
    R S_first = R(INFINITY);
    bool valid_first = false;
    int mover = NO_TRIGGER;  // the smaller of this half-wave's samples that moved the waypoint index (sequential mode)
    __shared__ int sh_mover[ROWS];
    // `S[k] += ...` in the reference's order (f32 race car, see Rollout::chunk) for ALL samples of the workgroup at once:
    // every (half-)wave leaves its per-step costs and the terminal cost in a row of LDS, then ONE wave adds them up with
    // a lane per sample -- T + 1 dependent additions for the workgroup instead of for each of its waves (150 of the
    // ~1000 VALU instructions a sample cost: the launch is VALU-bound).  Row pitch 129: a column is conflict-free.
    constexpr bool F32 = sizeof(R) == 4;
    const bool lds_sum = F32 && P.accumulate;
    __shared__ float sh_st[F32 ? SAMPLES : 1][F32 ? 129 : 1];
    const int sidx = wid * SPW + h;  // row of this (half-)wave; its first-pass sample sits ROWS further
#pragma unroll
    for (int pass = 0; pass < SEQ; ++pass) {
    const int k = ((blockIdx.x * SEQ + pass) * DUAL_WAVES + wid) * SPW + h;  // this lane's sample
    valid = k < P.K;
    const bool live = valid && k >= k_start;

    // ---- S1: this lane's noise for its two steps ---------------------------------------------------------
    e00 = 0.f; e01 = 0.f; e10 = 0.f; e11 = 0.f;
    STAMP(8);
    if (valid && a0) {
        if (use_philox) {
            unsigned r[4];
            px::philox4x32_10((unsigned)(k + P.k_offset), (unsigned)l32, iter, (unsigned)(P.noise_stream + agent), P.seed_lo, P.seed_hi, r);
            px::box_muller(r[0], r[1], P.chol, e00, e01);
            px::box_muller(r[2], r[3], P.chol, e10, e11);
            if (!a1) { e10 = 0.f; e11 = 0.f; }
        } else {
            const float *pe = eps_tensor(P, iter, agent) + ((size_t)k * T + t0) * 2;
            const float2 ea = *reinterpret_cast<const float2 *>(pe);
            e00 = ea.x;
            e01 = ea.y;
            if (a1) {
                const float2 eb = *reinterpret_cast<const float2 *>(pe + 2);
                e10 = eb.x;
                e11 = eb.y;
            }
        }
    }
    STAMP(9);
    S_k = R(INFINITY);
    if (__ballot(live) != 0ull || lookback) {  // at least one of the wave's two samples still needs its rollout (look-back: every wave takes part)
        const bool exploit = (k + P.k_offset) < P.n_exploit;
        R u00 = 0, u01 = 0, u10 = 0, u11 = 0;  // u<step><channel>
        if (a0) { u00 = u_[2 * t0]; u01 = u_[2 * t0 + 1]; }
        if (a1) { u10 = u_[2 * t1]; u11 = u_[2 * t1 + 1]; }
        R v00 = exploit ? u00 + (R)e00 : (R)e00, v01 = exploit ? u01 + (R)e01 : (R)e01;  // :116-119
        R v10 = exploit ? u10 + (R)e10 : (R)e10, v11 = exploit ? u11 + (R)e11 : (R)e11;
        if (clamp_rollout) {  // `_g` :285-289
            v00 = mf::clamp(v00, P.umax0); v01 = mf::clamp(v01, P.umax1);
            v10 = mf::clamp(v10, P.umax0); v11 = mf::clamp(v11, P.umax1);
        }
        if (!a0) { v00 = 0; v01 = 0; }
        if (!a1) { v10 = 0; v11 = 0; }

        // ---- dynamics: scans over the lane totals of each half, then one add for the lane's second step ----
        const R x_0 = (R)sv.x0[0], y_0 = (R)sv.x0[1], yaw_0 = (R)sv.x0[2];
        R px0, py0, yw0, vl0 = 0, px1, py1, yw1, vl1 = 0;  // state after the lane's first / second step
        R sin_w0 = 0, cos_w0 = 1;  // sin/cos(yw0): the dynamics of the second step need it, so does the outline test of the first
        if (MODEL == MODEL_DIFF) {  // :194-196
            const R d0 = v01 * P.dt, d1 = v11 * P.dt;
            const R yb0 = yaw_0 + wv::shift_up1_seg<SPW>(wv::scan_incl_seg<wv::OpAdd, SPW>(d0 + d1), R(0));
            yw0 = yb0 + d0;
            yw1 = yw0 + d1;
            R s0, c0, s1, c1;
            mf::sincos_(yb0, s0, c0);
            mf::sincos_(yw0, s1, c1);
            sin_w0 = s1;
            cos_w0 = c1;
            const R dx0 = v00 * c0 * P.dt, dx1 = v10 * c1 * P.dt, dy0 = v00 * s0 * P.dt, dy1 = v10 * s1 * P.dt;
            px0 = x_0 + wv::shift_up1_seg<SPW>(wv::scan_incl_seg<wv::OpAdd, SPW>(dx0 + dx1), R(0)) + dx0;
            px1 = px0 + dx1;
            py0 = y_0 + wv::shift_up1_seg<SPW>(wv::scan_incl_seg<wv::OpAdd, SPW>(dy0 + dy1), R(0)) + dy0;
            py1 = py0 + dy1;
        } else {  // mppi_race_car.py:190-193, controls = [steer, accel]
            const R vel_0 = (R)sv.x0[3];
            const R dv0 = a0 ? v01 * P.dt : R(0), dv1 = a1 ? v11 * P.dt : R(0);
            const R vb0 = vel_0 + wv::shift_up1_seg<SPW>(wv::scan_incl_seg<wv::OpAdd, SPW>(dv0 + dv1), R(0));
            vl0 = vb0 + dv0;
            vl1 = vl0 + dv1;
            const R dp0 = a0 ? vb0 / P.wheel_base * mf::tan_(v00) * P.dt : R(0);
            const R dp1 = a1 ? vl0 / P.wheel_base * mf::tan_(v10) * P.dt : R(0);
            const R yb0 = yaw_0 + wv::shift_up1_seg<SPW>(wv::scan_incl_seg<wv::OpAdd, SPW>(dp0 + dp1), R(0));
            yw0 = yb0 + dp0;
            yw1 = yw0 + dp1;
            R s0, c0, s1, c1;
            mf::sincos_(yb0, s0, c0);
            mf::sincos_(yw0, s1, c1);
            sin_w0 = s1;
            cos_w0 = c1;
            const R dx0 = a0 ? vb0 * c0 * P.dt : R(0), dx1 = a1 ? vl0 * c1 * P.dt : R(0);
            const R dy0 = a0 ? vb0 * s0 * P.dt : R(0), dy1 = a1 ? vl0 * s1 * P.dt : R(0);
            px0 = x_0 + wv::shift_up1_seg<SPW>(wv::scan_incl_seg<wv::OpAdd, SPW>(dx0 + dx1), R(0)) + dx0;
            px1 = px0 + dx1;
            py0 = y_0 + wv::shift_up1_seg<SPW>(wv::scan_incl_seg<wv::OpAdd, SPW>(dy0 + dy1), R(0)) + dy0;
            py1 = py0 + dy1;
        }
        STAMP(10);

        // ---- waypoint index of the lane's two calls ---------------------------------------------------------
        const int t_last = T - 1, lane_last = t_last >> 1, sub_last = t_last & 1;
        int idx0 = c, idx1 = c, p_half = c, idx_term = c;  // p_half / idx_term: uniform within a half
        R lb_total = R(INFINITY);
        bool lb_done = false;
        if constexpr (LBK) {
          if (lookback) {
            lb_done = true;
            // ---- the sequential index in this launch: fused_lookback for a half-wave per sample, two calls per lane --------
            static_assert(!LBK || (HL == LB_CAND && ROWS == 32), "a half-wave prices its sample under every offset, a lane per offset");
            const unsigned tag = lb_tag(P.lb_seq);
            const int b = blockIdx.x;
            if (threadIdx.x < LB_CAND) {
                const int j = threadIdx.x;
                sh_row[j] = lb_mine;
                R *pair = reinterpret_cast<R *>(&sh_c[j >> 1]);  // {x_2q, x_2q+1, y_2q, y_2q+1}
                pair[j & 1] = lb_mine.x;
                pair[2 + (j & 1)] = lb_mine.y;
            }
            __syncthreads();
            // pass A: the lane's two calls
            const R xs[2] = {(valid && a0) ? px0 : R(NAN), (valid && a1) ? px1 : R(NAN)}, ys[2] = {py0, py1};
            int m2[2];
            bool bad_l;
            lb_scan<2, R>(sh_c, xs, ys, m2, bad_l);
            const unsigned long long bad = __ballot(bad_l && valid);
            STAMP(27);
            int m = max((valid && a0) ? m2[0] : 0, (valid && a1) ? m2[1] : 0);
            m = wv::scan_incl_half<wv::OpMaxInt>(m);
            const int Mh = h ? wv::read_lane(m, 63) : wv::read_lane(m, 31);  // this half's (sample's) largest offset
            const bool bad_h = ((h ? (bad >> 32) : bad) & 0xffffffffull) != 0ull;
            if (l32 == 0) sh_M[sidx] = Mh | (bad_h ? LB_BAD : 0);
            __syncthreads();
            STAMP(28);
            // the workgroup's maximum and, for sample s, the maximum of the samples before it (a half-wave of 32 values)
            const int mv = sh_M[l32];
            const int incl = wv::scan_incl_half<wv::OpMaxInt>(mv & (LB_BAD - 1));
            const int Mb = wv::read_lane(incl, 31);
            const bool bad_b = (__ballot((mv & LB_BAD) != 0) & 0xffffffffull) != 0ull;
            unsigned *slots = P.hyp_slots;
            if (wid == 0) lb_publish(slots, b, tag | (bad_b ? (unsigned)LB_BAD : 0u) | (unsigned)Mb, lane);
            // this half's sample priced under every offset, lanes over the offset (`S[k] =`: the last step's stage cost
            // and the terminal cost, both against the same waypoint)
            {
                const int src = (SPW == 2 ? 32 * h : 0) + lane_last;
                const R lxs = sub_last ? px1 : px0, lys = sub_last ? py1 : py0, lyaws = sub_last ? yw1 : yw0;
                const R xT = __shfl(lxs, src), yT = __shfl(lys, src), yawT = __shfl(lyaws, src);
                const R ua = sub_last ? u10 : u00, ub = sub_last ? u11 : u01, va = sub_last ? v10 : v00, vb = sub_last ? v11 : v01;
                const R ctrl = (ua * P.sinv[0] + ub * P.sinv[2]) * va + (ua * P.sinv[1] + ub * P.sinv[3]) * vb;  // :124
                const R ctrlT = __shfl(ctrl, src);
                // the collision test of that one state (:301-313): a lane per circle -- lanes m < 32 of the wave hold circles
                // 0 .. 31 -- instead of every lane walking all circles for a result only one lane's state needs
                bool hitT;
                if (P.obstacle_model == OBS_CIRCLE && P.n_obs <= 32) {
                    const R ox = __shfl(obs.x, l32), oy = __shfl(obs.y, l32), o2 = __shfl(obs.r2, l32);
                    const R ddx = xT - ox, ddy = yT - oy;
                    const unsigned long long hm = __ballot(l32 < P.n_obs && ddx * ddx + ddy * ddy < o2);
                    hitT = ((h ? (hm >> 32) : hm) & 0xffffffffull) != 0ull;
                } else {
                    const bool hit_lane = collided<false>(P, lxs, lys, lyaws, obs);
                    hitT = ((__ballot(hit_lane) >> src) & 1ull) != 0ull;
                }
                const VecT4<R> row = sh_row[min(l32, nc - 1)];
                const R rr[4] = {row.x, row.y, row.z, row.w};
                R st_c = tracking_cost_row<R, MODEL>(P, P.ws, wrap_stage, rr, xT, yT, yawT, R(0));
                if (hitT) st_c += P.penalty;
                R term = tracking_cost_row<R, MODEL>(P, P.wt, wrap_term, rr, xT, yT, yawT, R(0));
                if (hitT) term += P.penalty;
                sh_cost[sidx][l32] = valid ? (st_c + P.gamma * ctrlT) + term : R(INFINITY);
            }
            STAMP(30);
            if (wid == 0) {
                int E = 0;
                bool bad_e = false;
                if (!lb_wait(slots, b, tag, lane, E, bad_e)) bad_e = true;
                const bool reach = lb_reach(max(E, Mb), P.window, P.n_ref, c);
                if ((bad_e || !reach) && !bad_b && lane == 0) __hip_atomic_fetch_or(&slots[b], (unsigned)LB_BAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) sh_E = E;
                STAMP(29);
            }
            __syncthreads();
            STAMP(31);
            // pass B: lane l of a half holds sample l's index and cost; this half's own sample is sample sidx
            const int shifted = wv::dpp<wv::DPP_WAVE_SHR1>(incl, 0);  // (every lane takes part: a lane switched off is no source)
            const int before_l = l32 == 0 ? 0 : shifted;
            const int a_l = min(max(max(sh_E, before_l), mv & (LB_BAD - 1)), nc - 1);
            const R S_l = sh_cost[l32][a_l];
            lb_total = __shfl(S_l, sidx);
            p_half = c + __shfl(a_l, sidx);
          }
        }
        if (lb_done) {
        } else if (!P.sequential) {
            const int wlen = window_len<R>(P.window, P.n_ref, c);
            idx0 = use_win ? nearest_in_window_lds(sh_win, c, wlen, px0, py0) : nearest_in_window(ref, c, wlen, px0, py0);
            idx1 = use_win ? nearest_in_window_lds(sh_win, c, wlen, px1, py1) : nearest_in_window(ref, c, wlen, px1, py1);
            idx_term = sub_last ? idx1 : idx0;  // meaningful on the lane that holds the last step
        } else {
            const int wlen = window_len<R>(P.window, P.n_ref, c);
            const R d00 = dist2(ref, c, px0, py0), d01 = dist2(ref, c, px1, py1);
            int fm0 = c, fm1 = c;  // first minimum of each of the lane's two calls over the window at c (see Rollout::chunk)
            bool trig = false;
            if (use_win) {
                fm0 = nearest_in_window_lds(sh_win, c, wlen, px0, py0);
                fm1 = nearest_in_window_lds(sh_win, c, wlen, px1, py1);
                trig = (a0 && fm0 != c) || (a1 && fm1 != c);
            } else {
#pragma unroll 4
                for (int j = 1; j < wlen; ++j)
                    trig |= (a0 && dist2(ref, c + j, px0, py0) < d00) || (a1 && dist2(ref, c + j, px1, py1) < d01);
            }
            unsigned long long m = __ballot(trig && live);
            for (int hh = 0; hh < SPW; ++hh) {  // a smaller sample of the workgroup moved the index too: skip (see Rollout::chunk)
                const unsigned long long half = SPW == 2 ? 0xffffffffull << (32 * hh) : ~0ull;
                if ((m & half) == 0ull) continue;
                const int k_hh = k - h + hh;
                int old = 0;
                if (lane == 0) old = atomicMin(&sh_first, k_hh);
                if (wv::read_lane(old, 0) < k_hh) m &= ~half;
            }
            int p_a = c, p_b = c, term_a = c, term_b = c;
            const bool to_end = c + wlen >= P.n_ref;
            if (m != 0ull && !use_win) {
                R b0 = d00, b1 = d01;
#pragma unroll 4
                for (int j = 1; j < wlen; ++j) {
                    const R e0 = dist2(ref, c + j, px0, py0), e1 = dist2(ref, c + j, px1, py1);
                    if (e0 < b0) { b0 = e0; fm0 = c + j; }
                    if (e1 < b1) { b1 = e1; fm1 = c + j; }
                }
            }
            for (int hh = 0; hh < SPW; ++hh) {  // thread the index through that sample's calls in order
                if ((SPW == 2 ? ((hh ? (m >> 32) : m) & 0xffffffffull) : m) == 0ull) continue;
                int p = c;
                for (int t = 0; t < T; ++t) {
                    const int src = hh * HL + (t >> 1);
                    const int a = (t & 1) ? wv::read_lane(fm1, src) : wv::read_lane(fm0, src);
                    const bool direct = a >= p && (to_end || p == c);
                    const R xt = (t & 1) ? wv::read_lane(px1, src) : wv::read_lane(px0, src);
                    const R yt = (t & 1) ? wv::read_lane(py1, src) : wv::read_lane(py0, src);
                    if (direct) p = a;
                    else p = nearest_uniform(ref, p, window_len<R>(P.window, P.n_ref, p), xt, yt, lane);
                    if (lane == src) { if (t & 1) idx1 = p; else idx0 = p; }
                    if (t == t_last) {  // the terminal call searches once more from the same state (:244)
                        // (from a direct answer a the suffix [a, ..) has its first minimum at a itself)
                        if (!(direct && (to_end || p == c)))
                            p = nearest_uniform(ref, p, window_len<R>(P.window, P.n_ref, p), xt, yt, lane);
                        if (hh) { term_b = p; p_b = p; } else { term_a = p; p_a = p; }
                    }
                }
            }
            p_half = h ? p_b : p_a;
            idx_term = h ? term_b : term_a;
        }
        STAMP(11);

        // ---- costs -----------------------------------------------------------------------------------------------
        auto stage_cost = [&](R x, R y, R yaw, R vel, int idx, R ua, R ub, R va, R vb, bool &hit, bool have_sc = false) {
            hit = collided<MODEL == MODEL_RACE>(P, x, y, yaw, obs, have_sc, sin_w0, cos_w0);
            R st_c = tracking_cost<R, MODEL>(P, P.ws, wrap_stage, idx, x, y, yaw, vel);
            if (hit) st_c += P.penalty;
            R ctrl;
            if (MODEL == MODEL_DIFF) ctrl = (ua * P.sinv[0] + ub * P.sinv[2]) * va + (ua * P.sinv[1] + ub * P.sinv[3]) * vb;
            else ctrl = ua * (P.sinv[0] * va + P.sinv[1] * vb) + ub * (P.sinv[2] * va + P.sinv[3] * vb);
            return st_c + P.gamma * ctrl;
        };
        // state of the last step as held by lane_last of each half
        const R lx = sub_last ? px1 : px0, ly = sub_last ? py1 : py0, lyaw = sub_last ? yw1 : yw0, lvel = sub_last ? vl1 : vl0;
        R total;
        if (lb_done) {  // (priced above, under its exact index)
            total = lb_total;
        } else if (P.accumulate) {
            bool hit0, hit1;
            const R st0 = stage_cost(px0, py0, yw0, vl0, idx0, u00, u01, v00, v01, hit0, true);
            STAMP(12);
            const R st1 = stage_cost(px1, py1, yw1, vl1, idx1, u10, u11, v10, v11, hit1);
            STAMP(13);
            const bool hit_l = sub_last ? hit1 : hit0;
            R term = tracking_cost<R, MODEL>(P, P.wt, wrap_term, idx_term, lx, ly, lyaw, lvel);
            if (hit_l) term += P.penalty;
            STAMP(14);
            if (F32) {  // (lds_sum) summed after the passes, a lane per sample
                float *row = sh_st[(pass == SEQ - 1 ? 0 : ROWS) + sidx];
                if (a0) row[t0] = (float)st0;
                if (a1) row[t1] = (float)st1;
                if (l32 == lane_last) row[T] = (float)term;
                total = R(0);
            } else {
                const R part = wv::scan_incl_seg<wv::OpAdd, SPW>((a0 ? st0 : R(0)) + (a1 ? st1 : R(0)));
                const R acc_a = wv::read_lane(part, HL - 1), acc_b = wv::read_lane(part, 63);
                total = h ? acc_b + wv::read_lane(term, (SPW == 2 ? 32 : 0) + lane_last) : acc_a + wv::read_lane(term, lane_last);
            }
        } else {  // `S[k] =`: only the last step's stage cost survives (:124)
            bool hit_l;
            const int idx_l = sub_last ? idx1 : idx0;
            const R st_l = stage_cost(lx, ly, lyaw, lvel, idx_l, sub_last ? u10 : u00, sub_last ? u11 : u01,
                                      sub_last ? v10 : v00, sub_last ? v11 : v01, hit_l);
            R term = tracking_cost<R, MODEL>(P, P.wt, wrap_term, idx_term, lx, ly, lyaw, lvel);
            if (hit_l) term += P.penalty;
            const R both = st_l + term;
            total = h ? wv::read_lane(both, (SPW == 2 ? 32 : 0) + lane_last) : wv::read_lane(both, lane_last);
        }
        S_k = total;
        if (l32 == 0 && live) {
            if (!lds_sum) S_[k] = total;
            pout_[k] = p_half;
        }
        if (live && P.sequential && !lb_done && p_half != c) mover = min(mover, k);  // (uniform within the half; SEQ passes ascend)
    }
    if (valid && !live) S_k = S_[k];  // final from an earlier speculation round
    if (SEQ == 2 && pass == 0) {
        f00 = e00; f01 = e01; f10 = e10; f11 = e11;
        S_first = S_k;
        valid_first = valid;
    }
    }  // pass
    STAMP(2);

    // ---- the workgroup's softmin record over its samples (S5-S6) -----------------------------------------------
    if (lds_sum) {
        __syncthreads();
        if (wid == 0 && lane < SAMPLES) {  // slot = lane: rows of the last pass first, then those of the first pass
            const int pass_s = SEQ == 2 && lane < ROWS ? 1 : 0;
            const int k_s = (blockIdx.x * SEQ + pass_s) * (DUAL_WAVES * SPW) + (lane % ROWS);
            float S = INFINITY;
            if (k_s < P.K) {
                if (k_s >= k_start) {
                    const float *row = sh_st[lane];
                    S = 0.f;
                    for (int t = 0; t <= T; ++t) S += row[t];
                    S_[k_s] = (R)S;
                } else {
                    S = (float)S_[k_s];  // final from an earlier speculation round
                }
            }
            sh_S[lane] = (R)S;
        }
    } else if (l32 == 0) {
        sh_S[sidx] = S_k;
        if (SEQ == 2) sh_S[ROWS + sidx] = S_first;
    }
    if (l32 == 0 && seq_search) sh_mover[sidx] = mover;
    __syncthreads();
    if (lds_sum) {
        S_k = sh_S[sidx];
        if (SEQ == 2) S_first = sh_S[ROWS + sidx];
    }
    STAMP(3);
    R s_min = R(INFINITY);
    int n_hit_i = 0;  // samples of the workgroup whose cost carries a collision penalty (mppi_stats::n_collided): a ballot per
                      // pass over the costs every lane reads here anyway (a loop by the record's thread cost config 3 1.3 us)
#pragma unroll
    for (int i = 0; i < SAMPLES; i += HL) {
        const R sv_i = i + l32 < SAMPLES ? sh_S[i + l32] : R(INFINITY);
        s_min = fmin(s_min, sv_i);
        const unsigned long long hb = __ballot(sv_i >= P.penalty && sv_i < R(INFINITY));
        n_hit_i += __popcll(HL == 64 ? hb : (hb & 0xffffffffull));
    }
    const R rho = wv::read_lane(wv::scan_incl_seg<wv::OpMin, SPW>(s_min), HL - 1);
    const R e = valid ? mf::exp_(-P.beta * (S_k - rho)) : R(0);  // :175
    const R e_first = SEQ == 2 && valid_first ? mf::exp_(-P.beta * (S_first - rho)) : R(0);
    if (l32 == 0) {
        sh_e[sidx] = e;
        if (SEQ == 2) sh_e[ROWS + sidx] = e_first;
    }
    {
        R *dst = &sh_acc[sidx][4 * l32];
        if (SEQ == 2) {
            dst[0] = e * (R)e00 + e_first * (R)f00; dst[1] = e * (R)e01 + e_first * (R)f01;
            dst[2] = e * (R)e10 + e_first * (R)f10; dst[3] = e * (R)e11 + e_first * (R)f11;
        } else {
            dst[0] = e * (R)e00; dst[1] = e * (R)e01; dst[2] = e * (R)e10; dst[3] = e * (R)e11;
        }
    }
    __syncthreads();
    const size_t slot = (size_t)agent * P.slots + blockIdx.x;  // this workgroup's record
    R *out = partials + slot * record_len(T, (int)sizeof(R));
    for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) {  // W_b[t] = sum_k e_k eps[k, t], :132-135
        R acc = 0;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) acc += sh_acc[q][i];
        out[4 + i] = acc;
    }
    if (threadIdx.x == 64 * (DUAL_WAVES - 1)) {
        R eta = 0, eta2 = 0;
        const R n_hit = P.obstacle_model != OBS_NONE ? (R)n_hit_i : R(0);
#pragma unroll
        for (int q = 0; q < SAMPLES; ++q) {
            const R ew = sh_e[q];
            eta += ew;
            eta2 += ew * ew;
        }
        out[0] = rho;
        out[1] = eta;
        out[2] = eta2;
        *reinterpret_cast<VecT4<R> *>(P.heads + 4 * slot) = VecT4<R>{rho, eta, eta2, n_hit};
        if (seq_search) publish_first_mover<ROWS>(sh_mover, &(P.st + agent)->first_k);
    }
    STAMP(4);
}

// ------------------------------------------------------------------------------------------
// 64 < T <= 96, race car, f32: two samples per wave and THREE steps per lane.  With two steps per lane and one sample per
// wave (k_rollout_dual<.., 1, ..>) a horizon of 75 steps keeps 38 of 64 lanes busy; here a half-wave owns a sample and lane l
// of the half the steps 3l, 3l+1, 3l+2 -- 25 of 32 lanes at T = 75.  A lane's three steps straddle two Philox blocks (the
// sampler's counter is (k, t >> 1): words r0,r1 for even t, r2,r3 for odd t), so it draws two blocks and uses six of
// their eight words.  Same arithmetic per step as k_rollout_dual (the association of the prefix sums differs); frozen
// waypoint index, `S[k] +=` summed in the reference's order through LDS (a lane per sample), one record per 32 samples.
// ------------------------------------------------------------------------------------------
constexpr int TRI_STEPS = 3, TRI_SAMPLES = 2 * DUAL_WAVES;
// SHARE: held to 64 VGPRs so that two workgroups share a CU (one's barriers, one-wave sum and record stores run under the
// other's arithmetic): K = 65536 59 -> 46 us per launch.  A launch of at most one workgroup per CU takes the instantiation
// without the cap (two registers more, 6 % faster at K = 8192).
template <bool PLAIN, bool SHARE>
__global__ __launch_bounds__(64 * DUAL_WAVES, SHARE ? 8 : 1) void k_rollout_tri(const DevState *st_pre, const KParams<float> P,
                                                                                float *__restrict__ partials) {
    using R = float;
    constexpr int NS = TRI_STEPS, HL = 32, ROWS = TRI_SAMPLES;
    const bool use_philox = PLAIN || P.use_philox, clamp_rollout = PLAIN || P.clamp_rollout;
    const bool wrap_stage = PLAIN || P.wrap_stage, wrap_term = PLAIN || P.wrap_term;  // (PLAIN: the race car's own switches)
    __shared__ R sh_S[ROWS];
    __shared__ R sh_e[ROWS];
    __shared__ __attribute__((aligned(16))) R sh_acc[ROWS][2 * NS * HL];
    __shared__ float sh_st[ROWS][129];  // per-step costs + the terminal cost of each sample; pitch 129: a column is conflict-free
    __shared__ RefPair<R> sh_win[WINDOW_LDS_MAX / 2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, h = lane >> 5, l32 = lane & 31;
    const int sidx = wid * 2 + h;
    const DevState sv = load_state(P, st_pre);
    const int c = sv.c, T = P.T;
    const unsigned iter = (unsigned)sv.iter;
    const R *__restrict__ ref = P.ref;
    int t[NS];
    bool a[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        t[i] = NS * l32 + i;
        a[i] = t[i] < T;
    }
    const ObsLanes<R> obs = load_obstacles(P, lane);
    const int wlen = window_len<R>(P.window, P.n_ref, c);
    const bool use_win = wlen > 1 && wlen <= WINDOW_LDS_MAX;
    if (use_win) {
        stage_window(sh_win, ref, c, wlen, (int)threadIdx.x, (int)blockDim.x);
        __syncthreads();
    }
    const int k = (blockIdx.x * DUAL_WAVES + wid) * 2 + h;  // this half-wave's sample
    const bool valid = k < P.K;

    // ---- S1: the lane's noise for its three steps ------------------------------------------------------------
    float e[NS][2];
#pragma unroll
    for (int i = 0; i < NS; ++i) { e[i][0] = 0.f; e[i][1] = 0.f; }
    if (valid && a[0]) {
        if (use_philox) {
            const unsigned jA = (unsigned)(NS * l32) >> 1;  // the block of step 3l; the lane's last step sits in block jA + 1
            unsigned rA[4], rB[4];
            px::philox4x32_10((unsigned)(k + P.k_offset), jA, iter, (unsigned)P.noise_stream, P.seed_lo, P.seed_hi, rA);
            px::philox4x32_10((unsigned)(k + P.k_offset), jA + 1u, iter, (unsigned)P.noise_stream, P.seed_lo, P.seed_hi, rB);
            const bool odd = (l32 & 1) != 0;  // 3l odd: the lane's steps take words {A2,A3}, {B0,B1}, {B2,B3}; else {A0,A1}, {A2,A3}, {B0,B1}
            px::box_muller(odd ? rA[2] : rA[0], odd ? rA[3] : rA[1], P.chol, e[0][0], e[0][1]);
            px::box_muller(odd ? rB[0] : rA[2], odd ? rB[1] : rA[3], P.chol, e[1][0], e[1][1]);
            px::box_muller(odd ? rB[2] : rB[0], odd ? rB[3] : rB[1], P.chol, e[2][0], e[2][1]);
        } else {
            const float *pe = eps_tensor(P, iter, 0) + ((size_t)k * T + t[0]) * 2;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                if (a[i]) {
                    const float2 ev = *reinterpret_cast<const float2 *>(pe + 2 * i);
                    e[i][0] = ev.x;
                    e[i][1] = ev.y;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (!a[i]) { e[i][0] = 0.f; e[i][1] = 0.f; }
    }

    // ---- S2: controls (:116-121), S3: dynamics as scans over the lane totals of each half -----------------
    const bool exploit = (k + P.k_offset) < P.n_exploit;
    R u[NS][2], v[NS][2];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        u[i][0] = a[i] ? P.u[2 * t[i]] : R(0);
        u[i][1] = a[i] ? P.u[2 * t[i] + 1] : R(0);
        v[i][0] = exploit ? u[i][0] + e[i][0] : e[i][0];
        v[i][1] = exploit ? u[i][1] + e[i][1] : e[i][1];
        if (clamp_rollout) {
            v[i][0] = mf::clamp(v[i][0], P.umax0);
            v[i][1] = mf::clamp(v[i][1], P.umax1);
        }
        if (!a[i]) { v[i][0] = 0; v[i][1] = 0; }
    }
    auto before = [&](R lane_total, R start) {  // `start` + the totals of the lanes before this one in its half
        return start + wv::shift_up1_half(wv::scan_incl_half<wv::OpAdd>(lane_total), R(0));
    };
    // mppi_race_car.py:190-193, controls = [steer, accel]
    R vb[NS], vl[NS], yb[NS], yw[NS], px_[NS], py_[NS], sn[NS], cs[NS];
    {
        R d[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) d[i] = a[i] ? v[i][1] * P.dt : R(0);
        R run = before((d[0] + d[1]) + d[2], (R)sv.x0[3]);
#pragma unroll
        for (int i = 0; i < NS; ++i) { vb[i] = run; run += d[i]; vl[i] = run; }
#pragma unroll
        for (int i = 0; i < NS; ++i) d[i] = a[i] ? vb[i] / P.wheel_base * mf::tan_(v[i][0]) * P.dt : R(0);
        run = before((d[0] + d[1]) + d[2], (R)sv.x0[2]);
#pragma unroll
        for (int i = 0; i < NS; ++i) { yb[i] = run; run += d[i]; yw[i] = run; }
#pragma unroll
        for (int i = 0; i < NS; ++i) mf::sincos_(yb[i], sn[i], cs[i]);
        R dx[NS], dy[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            dx[i] = a[i] ? vb[i] * cs[i] * P.dt : R(0);
            dy[i] = a[i] ? vb[i] * sn[i] * P.dt : R(0);
        }
        run = before((dx[0] + dx[1]) + dx[2], (R)sv.x0[0]);
#pragma unroll
        for (int i = 0; i < NS; ++i) { run += dx[i]; px_[i] = run; }
        run = before((dy[0] + dy[1]) + dy[2], (R)sv.x0[1]);
#pragma unroll
        for (int i = 0; i < NS; ++i) { run += dy[i]; py_[i] = run; }
    }

    // ---- the frozen waypoint index of the lane's three calls (mppi_race_car.py:157-174) and the costs ---------
    float *row = sh_st[sidx];
    int idx[NS];
    bool hit[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        idx[i] = use_win ? nearest_in_window_lds(sh_win, c, wlen, px_[i], py_[i]) : nearest_in_window(ref, c, wlen, px_[i], py_[i]);
        // (the sine / cosine of the yaw after step i are those the dynamics of step i + 1 took)
        hit[i] = i + 1 < NS ? collided<true>(P, px_[i], py_[i], yw[i], obs, true, sn[i + 1 < NS ? i + 1 : 0], cs[i + 1 < NS ? i + 1 : 0])
                            : collided<true>(P, px_[i], py_[i], yw[i], obs);
        R st_c = tracking_cost<R, MODEL_RACE>(P, P.ws, wrap_stage, idx[i], px_[i], py_[i], yw[i], vl[i]);
        if (hit[i]) st_c += P.penalty;
        const R ctrl = u[i][0] * (P.sinv[0] * v[i][0] + P.sinv[1] * v[i][1]) + u[i][1] * (P.sinv[2] * v[i][0] + P.sinv[3] * v[i][1]);  // mppi_race_car.py:84
        if (valid && a[i]) row[t[i]] = st_c + P.gamma * ctrl;
    }
    {
        const int t_last = T - 1, lane_last = t_last / NS, sub_last = t_last - NS * lane_last;
        if (valid && l32 == lane_last) {  // the terminal cost: the state after the last step (:97-99)
            const R lx = sub_last == 0 ? px_[0] : sub_last == 1 ? px_[1] : px_[2], ly = sub_last == 0 ? py_[0] : sub_last == 1 ? py_[1] : py_[2];
            const R lyaw = sub_last == 0 ? yw[0] : sub_last == 1 ? yw[1] : yw[2], lvel = sub_last == 0 ? vl[0] : sub_last == 1 ? vl[1] : vl[2];
            const int li = sub_last == 0 ? idx[0] : sub_last == 1 ? idx[1] : idx[2];
            const bool lh = sub_last == 0 ? hit[0] : sub_last == 1 ? hit[1] : hit[2];
            R term = tracking_cost<R, MODEL_RACE>(P, P.wt, wrap_term, li, lx, ly, lyaw, lvel);
            if (lh) term += P.penalty;
            row[T] = term;
        }
    }
    __syncthreads();
    // `S[k] += ...` in the reference's order, a lane per sample (see k_rollout_dual)
    if (wid == 0 && lane < ROWS) {
        const int k_s = blockIdx.x * ROWS + lane;
        float S = INFINITY;
        if (k_s < P.K) {
            const float *rw = sh_st[lane];
            S = 0.f;
            for (int tt = 0; tt <= T; ++tt) S += rw[tt];
            P.S[k_s] = S;
            P.pout[k_s] = c;
        }
        sh_S[lane] = S;
    }
    __syncthreads();

    // ---- the workgroup's softmin record over its 32 samples (S5-S6) -------------------------------------------
    const R S_k = sh_S[sidx];
    const R sv_l = sh_S[l32];
    const unsigned long long hb = __ballot(sv_l >= P.penalty && sv_l < R(INFINITY));
    const int n_hit_i = __popcll(hb & 0xffffffffull);
    const R rho = wv::read_lane(wv::scan_incl_half<wv::OpMin>(sv_l), HL - 1);
    const R ew = valid ? mf::exp_(-P.beta * (S_k - rho)) : R(0);  // mppi_race_car.py:197-209
    if (l32 == 0) sh_e[sidx] = ew;
    {
        R *dst = &sh_acc[sidx][2 * NS * l32];  // columns 2 t + channel of the lane's steps: contiguous
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            dst[2 * i] = ew * e[i][0];
            dst[2 * i + 1] = ew * e[i][1];
        }
    }
    __syncthreads();
    const size_t slot = blockIdx.x;
    R *out = partials + slot * record_len(T, (int)sizeof(R));
    for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) {  // W_b[t] = sum_k e_k eps[k, t]
        R acc = 0;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) acc += sh_acc[q][i];
        out[4 + i] = acc;
    }
    if (threadIdx.x == 64 * (DUAL_WAVES - 1)) {
        R eta = 0, eta2 = 0;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            const R w = sh_e[q];
            eta += w;
            eta2 += w * w;
        }
        out[0] = rho;
        out[1] = eta;
        out[2] = eta2;
        *reinterpret_cast<VecT4<R> *>(P.heads + 4 * slot) =
            VecT4<R>{rho, eta, eta2, P.obstacle_model != OBS_NONE ? (R)n_hit_i : R(0)};
    }
}

// ------------------------------------------------------------------------------------------
// The memory-bound form of the analytic path (north star: "coalesced HBM loads of the [K,T,nu] noise tensor"): the noise
// of `_calc_epsilon` (mppi_differential_drive.py:273-283) arrives as a tensor in HBM -- the caller's, or a slot of the
// noise ring (mppi_set_noise_ring) -- instead of being drawn in registers.  Read by the general kernels above, one
// workgroup per 32 samples, the launch is SLOWER than drawing the noise (32 batched config-2 agents: 61 against 41 us):
// every wave issues one load, waits an HBM round trip for it, computes, and leaves; nothing is in flight meanwhile.
// Here a workgroup STREAMS: it owns `n_pass` consecutive batches of 32 samples and requests batch i+1's rows (16 bytes per
// lane: the two steps a lane owns, eps[k, 2l .. 2l+1, :]) before it rolls out batch i, so the HBM latency is spent under
// arithmetic; a half-wave's 25 lanes read 400 contiguous bytes, a workgroup 12.8 KB per batch.  With the grid sized to
// two workgroups per CU (64 VGPRs), every CU keeps 25 KB of noise in flight throughout the launch.
//
// Diff-drive, two samples per wave / two steps per lane (T <= 64), frozen waypoint index, `S[k] =` (:124) -- the batched
// and large-K forms the memory roof matters for.  No workgroup barrier inside the loop: every half-wave merges ITS samples
// online into a private softmin record -- after a sample with cost S: rho <- min(rho, S), what is accumulated so far is
// rescaled by exp(-beta (rho_old - rho)) and the sample enters with e = exp(-beta (S - rho)) (:175); the lane that owns
// steps 2l, 2l+1 keeps the four sums W[2l .. 2l+1][0..1] (:132-135) in registers -- so the waves of a workgroup drift
// apart and cover each other's waits; the 32 private records meet once, at the end, in LDS, and the workgroup leaves ONE
// record for all its batches (k_finalize merges n_pass times fewer).
// The frozen index makes every cost call of a sample search the SAME window [c, c + wlen) and `S[k] =` keeps only the
// last step's: one search per sample, its <= 32 candidates spread over the lanes of the half-wave.
// ------------------------------------------------------------------------------------------
// PHILOX: the same loop with the noise drawn in registers (one Philox4x32 block per lane feeds its two steps, as in
// k_rollout_dual) -- the frozen-index `S[k] =` form then also pays one search per sample instead of one per lane and step,
// no workgroup barrier per 32 samples and one record per workgroup: 32 batched config-2 agents 63 -> ~35 us per launch.
template <typename R, bool MULTI, bool OBS, bool PHILOX>
__global__ __launch_bounds__(64 * DUAL_WAVES, sizeof(R) == 4 ? 8 : 1) void k_rollout_stream(const DevState *st_pre, const KParams<R> P,
                                                                                            R *__restrict__ partials, int n_pass) {
    constexpr int ROWS = DUAL_SAMPLES;  // half-waves of the workgroup = samples in flight
    const int agent = MULTI ? (int)blockIdx.y : 0;
    const R *__restrict__ u_ = P.u + (size_t)agent * 2 * P.T;
    R *__restrict__ S_ = P.S + (size_t)agent * P.K;
    __shared__ R sh_rho[ROWS];
    __shared__ R sh_eta[ROWS][3];
    __shared__ __attribute__((aligned(16))) R sh_acc[ROWS][128];
    __shared__ __attribute__((aligned(16))) float sh_raw[ROWS][128];
    __shared__ __attribute__((aligned(16))) R sh_u[128];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int sidx = wid * 2 + h;
    const DevState sv = load_state(P, st_pre + agent);
    const int c = sv.c, T = P.T, K = P.K, t0 = 2 * l32, t1 = t0 + 1;
    const bool a0 = t0 < T, a1 = t1 < T;
    const int n_batch = (K + ROWS - 1) / ROWS, b0 = (int)blockIdx.x * n_pass;
    const float *__restrict__ eps = PHILOX ? nullptr : eps_tensor(P, (unsigned)sv.iter, agent);
    const unsigned iter = (unsigned)sv.iter;
    // Unconditional loads from clamped addresses (what lies beyond the horizon or the last sample is masked where it is
    // used): a load under a per-lane condition makes the compiler wait for it at the end of the branch, i.e. right away.
    // Two 8-byte loads per lane (a row of odd length leaves its second half 8-byte aligned only).
    const float *__restrict__ row0 = eps + (size_t)(t0 < T ? t0 : T - 1) * 2, *__restrict__ row1 = eps + (size_t)(t1 < T ? t1 : T - 1) * 2;
    auto request = [&](int batch, float4 &e) {
        int k = batch * ROWS + sidx;
        if (PHILOX) {  // (drawn, not fetched: see the loop)
            e = make_float4(0.f, 0.f, 0.f, 0.f);
            return;
        }
        k = k < K ? k : K - 1;
        const size_t off = (size_t)k * T * 2;
        const float2 ea = *reinterpret_cast<const float2 *>(row0 + off), eb = *reinterpret_cast<const float2 *>(row1 + off);
        e = make_float4(ea.x, ea.y, eb.x, eb.y);
    };
    float4 e_next;
    request(b0, e_next);  // the first batch's rows fly while the prologue below runs
    const ObsLanes<R> obs = OBS ? load_obstacles(P, lane) : ObsLanes<R>{R(0), R(0), R(0)};
    const int wlen = window_len<R>(P.window, P.n_ref, c);
    // the frozen window's candidates, one per lane of each half-wave (the host sends windows of up to 32 candidates here)
    R cand_x = R(0), cand_y = R(0), cand_yaw = R(0);
    if (l32 < wlen) { cand_x = P.ref[4 * (c + l32)]; cand_y = P.ref[4 * (c + l32) + 1]; cand_yaw = P.ref[4 * (c + l32) + 2]; }
    if (tid < 128) sh_u[tid] = tid < 2 * T ? u_[tid] : R(0);  // u[t][channel]: the same for every batch
    __syncthreads();
    const R x_0 = (R)sv.x0[0], y_0 = (R)sv.x0[1], yaw_0 = (R)sv.x0[2];
    const int t_last = T - 1, lane_last = t_last >> 1, sub_last = t_last & 1;
    // the control cost of the last step, u^T Sigma^-1 v (:124), as two wave-uniform coefficients of v
    const R u_la = sh_u[2 * t_last], u_lb = sh_u[2 * t_last + 1];
    const R ca = wv::read_lane(u_la * P.sinv[0] + u_lb * P.sinv[2], 0), cb = wv::read_lane(u_la * P.sinv[1] + u_lb * P.sinv[3], 0);
    // this half-wave's private record (rho, eta, eta2 uniform within the half; w*: this lane's four columns)
    R rho_h = R(INFINITY), eta_h = R(0), eta2_h = R(0), w0 = R(0), w1 = R(0), w2 = R(0), w3 = R(0);
    R hit_h = R(0);  // samples of this half-wave whose cost carries a collision penalty
    for (int pass = 0; pass < n_pass; ++pass) {
        const int batch = b0 + pass;
        if (batch >= n_batch) break;  // (uniform over the workgroup)
        const int k = batch * ROWS + sidx;
        const bool valid = k < K;
        const bool exploit = (k + P.k_offset) < P.n_exploit;
        R v00, v01, v10, v11;
        {   // (the noise waits in this lane's own LDS slot until the sample's weight is known; the nominal controls come from
            // LDS per batch: eight registers fewer across the arithmetic below, which keeps the kernel at 64 VGPRs)
            float4 e = e_next;
            request(batch + 1, e_next);  // (past the last batch: a clamped re-read, cheaper than a branch around the loads)
            if (PHILOX) {
                unsigned r[4];
                px::philox4x32_10((unsigned)(k + P.k_offset), (unsigned)l32, iter, (unsigned)(P.noise_stream + agent), P.seed_lo, P.seed_hi, r);
                px::box_muller(r[0], r[1], P.chol, e.x, e.y);
                px::box_muller(r[2], r[3], P.chol, e.z, e.w);
            }
            *reinterpret_cast<float4 *>(&sh_raw[sidx][4 * l32]) = e;
            const VecT4<R> uu = *reinterpret_cast<const VecT4<R> *>(&sh_u[4 * l32]);  // u<step><channel>
            v00 = exploit ? uu.x + (R)e.x : (R)e.x; v01 = exploit ? uu.y + (R)e.y : (R)e.y;  // :116-119
            v10 = exploit ? uu.z + (R)e.z : (R)e.z; v11 = exploit ? uu.w + (R)e.w : (R)e.w;
        }
        if (P.clamp_rollout) {  // `_g` :285-289
            v00 = mf::clamp(v00, P.umax0); v01 = mf::clamp(v01, P.umax1);
            v10 = mf::clamp(v10, P.umax0); v11 = mf::clamp(v11, P.umax1);
        }
        if (!a0) { v00 = 0; v01 = 0; }
        if (!a1) { v10 = 0; v11 = 0; }
        // dynamics (:194-196), exactly as k_rollout_dual<.., MODEL_DIFF, 2, ..> associates them
        const R d0 = v01 * P.dt, d1 = v11 * P.dt;
        const R yb0 = yaw_0 + wv::shift_up1_half(wv::scan_incl_half<wv::OpAdd>(d0 + d1), R(0));
        const R yw0 = yb0 + d0, yw1 = yw0 + d1;
        R s0, c0, s1, c1;
        mf::sincos_(yb0, s0, c0);
        mf::sincos_(yw0, s1, c1);
        const R dx0 = v00 * c0 * P.dt, dx1 = v10 * c1 * P.dt, dy0 = v00 * s0 * P.dt, dy1 = v10 * s1 * P.dt;
        const R px0 = x_0 + wv::shift_up1_half(wv::scan_incl_half<wv::OpAdd>(dx0 + dx1), R(0)) + dx0, px1 = px0 + dx1;
        const R py0 = y_0 + wv::shift_up1_half(wv::scan_incl_half<wv::OpAdd>(dy0 + dy1), R(0)) + dy0, py1 = py0 + dy1;
        // the state after the last step, held by lane_last of each half; broadcast within the half
        const R lx_l = sub_last ? px1 : px0, ly_l = sub_last ? py1 : py0, lyaw_l = sub_last ? yw1 : yw0;
        const R lx = h ? wv::read_lane(lx_l, 32 + lane_last) : wv::read_lane(lx_l, lane_last);
        const R ly = h ? wv::read_lane(ly_l, 32 + lane_last) : wv::read_lane(ly_l, lane_last);
        const R lyaw = h ? wv::read_lane(lyaw_l, 32 + lane_last) : wv::read_lane(lyaw_l, lane_last);
        // the one search this sample's cost needs (`_get_nearest_waypoint`, :201-220: first minimum over [c, c + wlen))
        // (the waypoint the cost is taken against comes out of the candidate registers: a load from the path here would
        // make the wave wait for the batch requested above as well -- vector-memory loads return in order)
        R wx, wy, wyaw;
        {
            const R ddx = lx - cand_x, ddy = ly - cand_y;
            const R d = l32 < wlen ? ddx * ddx + ddy * ddy : R(INFINITY);
            const R ms = wv::scan_incl_half<wv::OpMin>(d);
            const R m = h ? wv::read_lane(ms, 63) : wv::read_lane(ms, 31);
            const int js = wv::scan_incl_half<wv::OpMinInt>(d == m ? l32 : 0x7fffffff);
            const int ja = wv::read_lane(js, 31), jb = wv::read_lane(js, 63);
            wx = h ? wv::read_lane(cand_x, jb) : wv::read_lane(cand_x, ja);
            wy = h ? wv::read_lane(cand_y, jb) : wv::read_lane(cand_y, ja);
            wyaw = h ? wv::read_lane(cand_yaw, jb) : wv::read_lane(cand_yaw, ja);
        }
        // costs (`S[k] =`: the last step's stage cost + the terminal cost of the same state, :124,:128; `_compute_cost`
        // :222-236 as tracking_cost<.., MODEL_DIFF> writes it)
        auto track = [&](const R (&w)[4], bool wrap) {
            const R yw = wrap ? mf::pymod(lyaw + P.two_pi, P.two_pi) : lyaw;
            const R ex = lx - wx, ey = ly - wy, eyaw = yw - wyaw;
            return w[0] * (ex * ex) + w[1] * (ey * ey) + w[2] * (eyaw * eyaw);
        };
        const bool hit = OBS ? collided<false>(P, lx, ly, lyaw, obs) : false;
        R st_c = track(P.ws, P.wrap_stage);
        if (hit) st_c += P.penalty;
        const R va = sub_last ? v10 : v00, vb = sub_last ? v11 : v01;
        const R ctrl_l = ca * va + cb * vb;
        const R ctrl = h ? wv::read_lane(ctrl_l, 32 + lane_last) : wv::read_lane(ctrl_l, lane_last);
        R term = track(P.wt, P.wrap_term);
        if (hit) term += P.penalty;
        const R S_k = (st_c + P.gamma * ctrl) + term;
        if (l32 == 0 && valid) S_[k] = S_k;
        if (OBS && valid && hit) hit_h += R(1);
        if (valid) {  // (uniform within the half) merge the sample into the half-wave's record
            const R rho_new = fmin(rho_h, S_k);
            const R scale = rho_h < R(INFINITY) ? mf::exp_(-P.beta * (rho_h - rho_new)) : R(0);
            const R e_w = mf::exp_(-P.beta * (S_k - rho_new));  // :175
            const float4 e = *reinterpret_cast<const float4 *>(&sh_raw[sidx][4 * l32]);  // (own slot: no barrier needed)
            w0 = w0 * scale + e_w * (R)e.x; w1 = w1 * scale + e_w * (R)e.y;  // :132-135
            w2 = w2 * scale + e_w * (R)e.z; w3 = w3 * scale + e_w * (R)e.w;
            eta_h = eta_h * scale + e_w;
            eta2_h = eta2_h * scale * scale + e_w * e_w;
            rho_h = rho_new;
        }
    }
    // the 32 private records -> the workgroup's record
    if (l32 == 0) sh_rho[sidx] = rho_h;
    __syncthreads();
    const R rho = wv::read_lane(wv::scan_incl_half<wv::OpMin>(sh_rho[l32]), 31);
    const R sc = rho_h < R(INFINITY) ? mf::exp_(-P.beta * (rho_h - rho)) : R(0);
    if (l32 == 0) { sh_eta[sidx][0] = sc * eta_h; sh_eta[sidx][1] = sc * sc * eta2_h; sh_eta[sidx][2] = hit_h; }
    *reinterpret_cast<VecT4<R> *>(&sh_acc[sidx][4 * l32]) = VecT4<R>{sc * w0, sc * w1, sc * w2, sc * w3};
    __syncthreads();
    const size_t slot = (size_t)agent * P.slots + blockIdx.x;  // this workgroup's record
    R *out = partials + slot * record_len(T, (int)sizeof(R));
    if (tid < 2 * T) {
        R acc = 0;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) acc += sh_acc[q][tid];
        out[4 + tid] = acc;
    }
    if (tid == 64 * (DUAL_WAVES - 1)) {
        R eta = 0, eta2 = 0, n_hit = 0;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) { eta += sh_eta[q][0]; eta2 += sh_eta[q][1]; n_hit += sh_eta[q][2]; }
        out[0] = rho;
        out[1] = eta;
        out[2] = eta2;
        *reinterpret_cast<VecT4<R> *>(P.heads + 4 * slot) = VecT4<R>{rho, eta, eta2, n_hit};
    }
}

// ------------------------------------------------------------------------------------------
// S5-S6: block-local softmin partials {rho_b, eta_b, eta2_b, W_b[T][2]}.
// ------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(256) void k_reduce(const KParams<R> P, R *__restrict__ partials) {
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

    R eta = 0, eta2 = 0, n_hit = 0;
    for (int i = tid; i < nk; i += blockDim.x) {
        const R e = mf::exp_(-P.beta * (P.S[k0 + i] - rho));  // :175
        sh_e[i] = e;
        eta += e;
        eta2 += e * e;
        if (P.obstacle_model != OBS_NONE && P.S[k0 + i] >= P.penalty) n_hit += R(1);
    }
    eta = wv::reduce<wv::OpAdd>(eta);
    eta2 = wv::reduce<wv::OpAdd>(eta2);
    n_hit = wv::reduce<wv::OpAdd>(n_hit);
    if (lane == 0) { sh_red[4 + wid] = eta; sh_red[8 + wid] = eta2; sh_red[12 + wid] = n_hit; }
    __syncthreads();
    R *out = partials + (size_t)blockIdx.x * record_len(P.T, (int)sizeof(R));
    if (tid == 0) {
        R a = 0, b = 0, nh = 0;
        for (int w = 0; w < nw; ++w) { a += sh_red[4 + w]; b += sh_red[8 + w]; nh += sh_red[12 + w]; }
        out[0] = rho;
        out[1] = a;
        out[2] = b;
        *reinterpret_cast<VecT4<R> *>(P.heads + 4 * (size_t)blockIdx.x) = VecT4<R>{rho, a, b, nh};
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
                    px::sample(P.seed_lo, P.seed_hi, iter, (unsigned)(k0 + i + P.k_offset), t, P.chol, e0, e1, (unsigned)P.noise_stream);
                } else {
                    const float2 e = *reinterpret_cast<const float2 *>(eps_tensor(P, iter, 0) + ((size_t)(k0 + i) * P.T + t) * 2);
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
            out[4 + ch * 128 + tid] = s;
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
// S7: merge partials, moving-average filter, update, clamp, shift, plant.  One block.
//
// This kernel is the serial tail of every iteration (one workgroup, nothing to overlap with), so it
// is written for latency: 4 waves (one per SIMD, so each issues at the full rate), ONE memory round
// trip -- every thread issues all of its loads (record heads + 32 16-byte W vectors) before the
// first use -- and arithmetic in the handle's precision A (fp32 handles merge in fp32: DPP
// reductions are one VALU op per step and exp is the hardware exp2; fp64 handles keep f64
// throughout for the 1e-9 parity tests).
//
// Block records (written by k_rollout_fused / k_reduce / k_merge) use the INTERNAL layout
// {rho, eta, eta2, pad, W[2T] padded to a 16-byte multiple}, so W is read as aligned 16-byte vectors;
// the per-rank record of the split step uses the ABI layout {rho, eta, eta2, W[2T]} in doubles.
// ------------------------------------------------------------------------------------------
// NT threads merge 256 records: thread = (16-byte column vc = tid % 32, group grp = tid / 32), NT/32 groups of
// 256/(NT/32) consecutive records.  NT = 256: the merge kernels' own launches; NT = 1024: the prologue of k_iter.
template <int NT> struct MergeShape {
    static constexpr int GROUPS = NT / 32, MAXJ = MERGE_MAX_RECORDS / GROUPS, WAVES = NT / 64;
};

template <typename A> struct alignas(16) VecT { A v[16 / sizeof(A)]; };

// LDS of the merge code.  w: the weighted noise in the filter's padded layout, [2 (T + W + 1)] (k_merge: W = 0,
// plain); u: the updated controls [2T]; s: 64 record scales per wave; red: block reductions; part: per-group
// partial sums.  Every region starts on a 16-byte boundary.
template <typename A, int NT = MERGE_THREADS> struct MergeLds {
    A *w, *u, *s, *red, *part;
    __device__ __forceinline__ MergeLds(char *smem, int T, int W) {
        w = reinterpret_cast<A *>(smem);
        u = w + ((2 * (T + W + 1) + 3) & ~3);
        s = u + ((2 * T + 3) & ~3);
        red = s + MERGE_MAX_WINDOWS * NT;
        part = red + 64;
    }
};

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ double fast_exp(double x) { return exp(x); }
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }  // 1 ulp
__device__ __forceinline__ double fast_div(double a, double b) { return a / b; }

// ---- the moving average of the weighted noise (`_moving_average_filter`), shared by k_finalize and k_eval_filter ----
// Sample t of channel d sits at sh_w[2 (t + H) + d], H = W / 2: H leading and W - H trailing slots hold zeros
// (np.convolve 'same', mppi_differential_drive.py:257-263) or copies of the first / last H samples
// (mppi_race_car.py:211-222: the padded signal is xx[:H] + xx + xx[-H:])
template <typename A> __device__ __forceinline__ void filter_store(A *sh_w, int i, A v, int T, int H, bool pad_copy) {
    const int t = i >> 1, d = i & 1;
    const A pv = pad_copy ? v : A(0);
    sh_w[2 * (t + H) + d] = v;
    if (t < H) sh_w[2 * t + d] = pv;
    if (t >= T - H) sh_w[2 * (t + 2 * H) + d] = pv;
    if (t == T - 1) sh_w[2 * (T + 2 * H) + d] = pv;  // odd W only
}
// filtered sample t of channel d (window W): taps padded[t + W - 1 - q], q = 0 .. W-1, in that order
template <typename A> __device__ __forceinline__ A filter_at(const A *sh_w, int t, int d, int T, int W, int f_filter) {
    const int H = W / 2;
    const A inv_w = fast_div(A(1), (A)W);
    if (f_filter == FILTER_NONE) return sh_w[2 * (t + H) + d];
    if (f_filter == FILTER_TORCH) {
        // conv1d(padding = H) over the padded signal, first T outputs (mppi_race_car_torch.py:211-222):
        // output t = padded rows t-H .. t-H+W-1, rows before the start are the convolution's zero padding
        A sacc = 0;
        for (int q = 0; q < W; ++q) {
            const int m = t - H + q;
            if (m >= 0) sacc += sh_w[2 * m + d] * inv_w;
        }
        return sacc;
    }
    const A *tap = sh_w + 2 * t + d;
    A sacc = 0;
    if (W == 10) {  // every reference variant; static LDS offsets, reads issue back to back
#pragma unroll
        for (int q = 0; q < 10; ++q) sacc += tap[2 * (9 - q)] * inv_w;
    } else {
        for (int q = 0; q < W; ++q) sacc += tap[2 * (W - 1 - q)] * inv_w;
    }
    if (f_filter == FILTER_DIFF) {  // the edge factors of mppi_differential_drive.py:265-269
        const int n_conv = (W + 1) / 2;
        if (t == 0) sacc *= fast_div((A)W, (A)n_conv);
        else if (t < n_conv) sacc *= fast_div((A)W, (A)(t + n_conv));
        if (t == T - 1)
            for (int q = 1; q < n_conv; ++q) sacc *= fast_div((A)W, (A)(q + n_conv - (W % 2)));
    }
    return sacc;
}

template <typename A> struct BlockRed {  // block-wide reductions through one LDS exchange each (256 threads)
    static __device__ __forceinline__ A min1(A v, A *sh, int tid) {
        v = wv::reduce<wv::OpMin>(v);
        __syncthreads();
        if ((tid & 63) == 0) sh[tid >> 6] = v;
        __syncthreads();
        A r = sh[0];
#pragma unroll
        for (int w = 1; w < MERGE_THREADS / 64; ++w) r = fmin(r, sh[w]);
        return r;
    }
    static __device__ __forceinline__ void add2(A &a, A &b, A *sh, int tid) {
        a = wv::reduce<wv::OpAdd>(a);
        b = wv::reduce<wv::OpAdd>(b);
        __syncthreads();
        if ((tid & 63) == 0) { sh[tid >> 6] = a; sh[32 + (tid >> 6)] = b; }
        __syncthreads();
        a = 0;
        b = 0;
#pragma unroll
        for (int w = 0; w < MERGE_THREADS / 64; ++w) { a += sh[w]; b += sh[32 + w]; }
    }
};

// Merge n <= 256 NWIN records with the rescale trick: rho = min rho_b,
// s_b = exp(-beta (rho_b - rho)), eta = sum s_b eta_b, W = sum s_b W_b.  Records come from a PREVIOUS
// launch: ordinary loads are coherent.  Written for latency -- the caller issues every load first thing
// (merge_load_*), before it touches anything else, so that one memory round trip covers them all; the
// heads are reduced per wave (each wave reads all 256 heads, DPP reductions, no block barrier).
// Record b lives in slot b; slots >= n are never written by a producer and the buffers are zero-filled and
// padded by 256 records at creation, so every load is unconditional and in bounds, and an absent slot enters
// with a zero scale.
// NWIN windows of 256 records: the same thread mapping in every window, all windows' loads in flight at once.
template <typename A, int NT = MERGE_THREADS, int NWIN = 1> struct MergeRegs {
    VecT<A> w[NWIN][MergeShape<NT>::MAXJ];
    A hr[4 * NWIN], he[4 * NWIN], he2[4 * NWIN];  // heads of records 256 win + lane + {0, 64, 128, 192}
    A hc[4 * NWIN];                                // their fourth word: samples that carry a collision penalty
};

template <typename A, int NT, int NWIN>
__device__ __forceinline__ void merge_load_heads(const A *__restrict__ heads, MergeRegs<A, NT, NWIN> &m) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 4 * NWIN; ++i) {  // the compact copy: consecutive lanes read consecutive 16 / 32 bytes
        const int b = lane + 64 * i;
        const size_t r = (size_t)b;
        const VecT4<A> hd = *reinterpret_cast<const VecT4<A> *>(heads + 4 * r);
        m.hr[i] = hd.x;
        m.he[i] = hd.y;
        m.he2[i] = hd.z;
        m.hc[i] = hd.w;
    }
}

template <typename A, int NT, int NWIN>
__device__ __forceinline__ void merge_load_tile(const A *__restrict__ recs, int T, int vt, MergeRegs<A, NT, NWIN> &m) {
    constexpr int VW = 16 / sizeof(A), MAXJ = MergeShape<NT>::MAXJ;
    const int tid = threadIdx.x, vc = tid & 31, grp = tid >> 5;
    const unsigned rbytes = (unsigned)record_len(T, (int)sizeof(A)) * (unsigned)sizeof(A);
    const int nvc = (2 * T + VW - 1) / VW;  // 16-byte columns of W
    const char *base = reinterpret_cast<const char *>(recs);
    const unsigned col = (unsigned)(4 + min(vt * 32 + vc, nvc - 1) * VW) * (unsigned)sizeof(A);
    const unsigned off0 = (unsigned)(grp * MAXJ) * rbytes + col;
#pragma unroll
    for (int win = 0; win < NWIN; ++win)
#pragma unroll
        for (int j = 0; j < MAXJ; ++j)
            m.w[win][j] = *reinterpret_cast<const VecT<A> *>(base + (off0 + (unsigned)(win * MERGE_MAX_RECORDS + j) * rbytes));
}

// `store(i, v)` receives w_eps[i] = W[i] / eta for i in [0, 2T); rho/eta/eta2 end up in every thread.
template <typename A, int NT, int NWIN, typename Store>
__device__ __forceinline__ void merge_combine(const A *__restrict__ recs, int n, int T, A beta,
                                              MergeRegs<A, NT, NWIN> &m, A *sh_s, A *sh_part, A &rho, A &eta, A &eta2,
                                              Store store, A *n_hit = nullptr) {
    constexpr int VW = 16 / sizeof(A), MAXJ = MergeShape<NT>::MAXJ, GROUPS = MergeShape<NT>::GROUPS;
    using V = VecT<A>;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int vc = tid & 31, grp = tid >> 5;
    const int nvc = (2 * T + VW - 1) / VW;
    A lr = A(INFINITY);
#pragma unroll
    for (int i = 0; i < 4 * NWIN; ++i) {
        if (lane + 64 * i >= n) m.hr[i] = A(INFINITY);
        lr = fmin(lr, m.hr[i]);
    }
    rho = wv::reduce<wv::OpMin>(lr);
    STAMP(25);
    A sc[4 * NWIN];
    eta = 0;
    eta2 = 0;
#pragma unroll
    for (int i = 0; i < 4 * NWIN; ++i) {
        sc[i] = lane + 64 * i < n ? fast_exp(-beta * (m.hr[i] - rho)) : A(0);
        eta += sc[i] * m.he[i];
        eta2 += sc[i] * sc[i] * m.he2[i];
    }
    eta = wv::reduce<wv::OpAdd>(eta);
    eta2 = wv::reduce<wv::OpAdd>(eta2);
    if (n_hit && *n_hit >= A(0)) {  // (the caller asks for the count by passing 0, and leaves it out with -1: no obstacles)
        A cnt = 0;
#pragma unroll
        for (int i = 0; i < 4 * NWIN; ++i) cnt += lane + 64 * i < n ? m.hc[i] : A(0);
        *n_hit = wv::reduce<wv::OpAdd>(cnt);
    }
    // this wave's two groups use the scales of records r0 .. r0 + 2 MAXJ - 1 only (a run inside one of the four
    // 64-record slots every lane holds): a wave-local exchange through 64 private LDS words, no block barrier
    // (per window)
    const int r0 = wid * 2 * MAXJ, slot = r0 >> 6;
    A *my_s = sh_s + wid * 64;  // window win at + win * NT
#pragma unroll
    for (int win = 0; win < NWIN; ++win)
        my_s[win * NT + lane] = slot == 0 ? sc[4 * win] : slot == 1 ? sc[4 * win + 1] : slot == 2 ? sc[4 * win + 2] : sc[4 * win + 3];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    STAMP(26);
    const A *sj_base = my_s + (r0 & 63) + (grp & 1) * MAXJ;
    const A inv_eta = fast_div(A(1), eta);
    const int n_tiles = (nvc + 31) / 32;
    for (int vt = 0; vt < n_tiles; ++vt) {
        V acc;
#pragma unroll
        for (int q = 0; q < VW; ++q) acc.v[q] = 0;
#pragma unroll
        for (int win = 0; win < NWIN; ++win)
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const A sj = sj_base[win * NT + j];
#pragma unroll
                for (int q = 0; q < VW; ++q) acc.v[q] += sj * m.w[win][j].v[q];
            }
        if (vt > 0) __syncthreads();  // the previous tile's readers of sh_part are done
        *reinterpret_cast<V *>(sh_part + (grp * 32 + vc) * VW) = acc;
        __syncthreads();
        if (__builtin_expect(vt + 1 < n_tiles, 0)) merge_load_tile<A, NT, NWIN>(recs, T, vt + 1, m);  // T > 64 (f64: 32) only
        for (int e = tid; e < 32 * VW; e += NT) {
            const int i = vt * 32 * VW + e;
            if (i < 2 * T) {
                A t = 0;
#pragma unroll
                for (int g = 0; g < GROUPS; ++g) t += sh_part[g * 32 * VW + e];
                store(i, t * inv_eta);  // w_eps = W / eta, :132-135
            }
        }
    }
    __syncthreads();
}

// ABI layout (doubles, {rho, eta, eta2, W[2T]}), n = number of ranks: few records, plain loops.
template <typename A, typename Store>
__device__ __forceinline__ void merge_abi(const double *recs, int n, int T, A beta, A *sh_s, A *sh_red, A &rho, A &eta,
                                          A &eta2, Store store, int stride = 0) {
    const int tid = threadIdx.x, plen = stride ? stride : partial_len(T);
    A hr = A(INFINITY), he = 0, he2 = 0;
    if (tid < n) {
        const double *pb = recs + tid * plen;
        hr = (A)pb[0];
        he = (A)pb[1];
        he2 = (A)pb[2];
    }
    rho = BlockRed<A>::min1(hr, sh_red, tid);
    const A sc = tid < n ? fast_exp(-beta * (hr - rho)) : A(0);
    if (tid < n) sh_s[tid] = sc;
    eta = sc * he;
    eta2 = sc * sc * he2;
    BlockRed<A>::add2(eta, eta2, sh_red, tid);
    const A inv_eta = fast_div(A(1), eta);
    for (int i = tid; i < 2 * T; i += MERGE_THREADS) {
        A t = 0;
        for (int b = 0; b < n; ++b) t += sh_s[b] * (A)recs[b * plen + 3 + i];
        store(i, t * inv_eta);  // w_eps = W / eta, :132-135
    }
    __syncthreads();
}

// groups of `group` <= 256 internal records -> one record each: internal layout (large K) or the ABI
// layout in doubles (the per-rank record of the split step)
template <typename A, bool ABI_OUT>
__global__ __launch_bounds__(MERGE_THREADS) void k_merge(const A *__restrict__ recs, const A *__restrict__ heads, int n,
                                                         int group, int T, A beta, void *__restrict__ out,
                                                         A *__restrict__ out_heads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MergeLds<A> L(smem, T, 0);
    A *sh_w = L.w;
    const int b0 = blockIdx.x * group, nb = min(group, n - b0);
    const A *mine = recs + (size_t)b0 * record_len(T, (int)sizeof(A));
    MergeRegs<A, MERGE_THREADS, 1> mr;
    merge_load_heads<A, MERGE_THREADS, 1>(heads + 4 * (size_t)b0, mr);
    merge_load_tile<A, MERGE_THREADS, 1>(mine, T, 0, mr);
    A rho, eta, eta2, n_hit = 0;
    merge_combine<A, MERGE_THREADS, 1>(mine, nb, T, beta, mr, L.s, L.part, rho, eta, eta2, [&](int i, A v) { sh_w[i] = v; }, &n_hit);
    // merge_combine leaves W / eta; a record carries W itself
    if (ABI_OUT) {
        double *o = reinterpret_cast<double *>(out) + (size_t)blockIdx.x * partial_len(T);
        for (int i = threadIdx.x; i < 2 * T; i += MERGE_THREADS) o[3 + i] = (double)(sh_w[i] * eta);
        if (threadIdx.x == 0) { o[0] = (double)rho; o[1] = (double)eta; o[2] = (double)eta2; }
    } else {
        A *o = reinterpret_cast<A *>(out) + (size_t)blockIdx.x * record_len(T, (int)sizeof(A));
        for (int i = threadIdx.x; i < 2 * T; i += MERGE_THREADS) o[4 + i] = sh_w[i] * eta;
        if (threadIdx.x == 0) {
            o[0] = rho; o[1] = eta; o[2] = eta2;
            *reinterpret_cast<VecT4<A> *>(out_heads + 4 * (size_t)blockIdx.x) = VecT4<A>{rho, eta, eta2, n_hit};
        }
    }
}

// raise this rank's flag in every rank's buffer (release: the records stored before the preceding barrier
// are visible to whoever sees the flag), then wait for every rank's flag in the own buffer.  Returns false
// after F.x_timeout ticks without them (and makes that sticky).  Every thread of the block calls this.
// `peer_ptr`: lane p (< x_nranks) of every wave holds the base of rank p's exchange buffer, fetched once up front
// (F.x_peers lives in device memory: indexing it inside the loops below would put a dependent load before every store)
__device__ __forceinline__ char *peer_base(unsigned long long peer_ptr, int p) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)peer_ptr, p);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(peer_ptr >> 32), p);
    return reinterpret_cast<char *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned long long load_peer_ptrs(const FinalizeParams &F) {
    const int lane = threadIdx.x & 63;
    return lane < F.x_nranks ? reinterpret_cast<unsigned long long>(F.x_peers[lane]) : 0ull;
}

__device__ __forceinline__ bool exchange_flags(const FinalizeParams &F, unsigned long long peer_ptr, size_t slot_off) {
    const int tid = threadIdx.x;
    char *own = peer_base(peer_ptr, F.x_rank);
    __syncthreads();
    if (tid < F.x_nranks) {  // (wave 0: tid == lane, so peer_ptr is this thread's peer)
        long long *theirs = reinterpret_cast<long long *>(reinterpret_cast<char *>(peer_ptr) + slot_off) + F.x_rank;
        __hip_atomic_store(theirs, F.x_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long *mine = reinterpret_cast<const long long *>(own + slot_off) + tid;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != F.x_seq) {
            if (wall_clock64() - t0 > (unsigned long long)F.x_timeout) {
                __hip_atomic_store(F.x_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the records behind the flags
    }
    __syncthreads();
    return __hip_atomic_load(F.x_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// wiring test: the full protocol of one iteration with a known pattern instead of a record (run it at least
// four times, so that both slots are REUSED and a stale cached copy of a peer's stores would show)
__global__ __launch_bounds__(64) void k_exchange_probe(const FinalizeParams F, int *ok_out) {
    const int tid = threadIdx.x, n_chk = 8;
    const size_t slot_off = (size_t)(F.x_seq & 1) * xchg_slot_bytes(F.T, F.x_nranks);
    const size_t recs_off = slot_off + sizeof(long long) * XCHG_MAX_RANKS;
    auto pattern = [&](int rank, int i) { return (double)(F.x_seq * 4096 + rank * 64 + i); };
    const unsigned long long peer_ptr = load_peer_ptrs(F);
    for (int p = 0; p < F.x_nranks; ++p) {
        double *rec = reinterpret_cast<double *>(peer_base(peer_ptr, p) + recs_off) + (size_t)F.x_rank * xchg_rec_len(F.T);
        if (tid < n_chk) rec[tid] = pattern(F.x_rank, tid);
    }
    bool ok = exchange_flags(F, peer_ptr, slot_off);
    const double *recs = reinterpret_cast<const double *>(peer_base(peer_ptr, F.x_rank) + recs_off);
    bool match = true;
    for (int r = 0; r < F.x_nranks; ++r)
        if (tid < n_chk) match &= recs[(size_t)r * xchg_rec_len(F.T) + tid] == pattern(r, tid);
    ok = ok && __ballot(!match) == 0ull;
    if (tid == 0) *ok_out = ok ? 1 : 0;
}

// MODE 0: F.partials = this GPU's block records (handle precision); 1: = the ranks' records, gathered by the
// caller (doubles, ABI layout); 2: block records + peer-to-peer exchange of the per-rank record
// S7: merge the records of the finished rollouts, moving average, update, shift, plant, next x0 call -- one
// workgroup of NT threads; state and controls are updated where they are or into F.st_out / F.u_out.
// The *_pre arguments repeat F.partials, F.heads, F.st, F.u and F.T: leading kernel arguments that the dispatcher
// preloads into SGPRs (-amdgpu-kernarg-preload-count), so the first loads do not wait for the argument fetch.
// PLAIN: the closed loop of the diff-drive NumPy controller on one GPU (sequential index, its moving average, the plant
// on the device, no host arguments, no trace) -- the run-time switches below are constants then (0.15 us per iteration at
// config 2, A/B on one box); launch_finalize picks it when every condition holds.
template <typename A, int MODE, int NT, int NWIN, bool PLAIN = false, bool HYPK = false>
__device__ __forceinline__ void finalize_body(const void *partials_pre, const void *heads_pre, const DevState *st_pre,
                                              const void *u_pre, int T_pre, const FinalizeParams &F, char *smem,
                                              int agent) {
    constexpr bool ABI_RECS = MODE == 1, XCHG = MODE == 2;
    const bool f_use_args = !PLAIN && F.use_args, f_sequential = PLAIN || F.sequential, f_plant = PLAIN || F.plant;
    const bool f_raise = !PLAIN && F.raise_at_path_end, f_clamp_u = !PLAIN && F.clamp_u, f_trace = !PLAIN && F.u0_trace != nullptr;
    const int f_model = PLAIN ? (int)MODEL_DIFF : F.model, f_filter = PLAIN ? (int)FILTER_DIFF : F.filter_mode;
    static_assert(MODE == 0 || NT == MERGE_THREADS, "the ABI / exchange variants are 256-thread kernels");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int T = F.T, W = F.filter_window, H = W / 2;
    const MergeLds<A, NT> L(smem, T, W);
    A *sh_w = L.w;  // weighted noise in the filter's padded layout: sample t of channel d at [2 (t + H) + d]
    A *sh_u = L.u;  // [2T] updated u
    // (`agent`: 0, or this workgroup's row of a batched launch -- every per-agent buffer is offset by it; the *_pre
    // pointers arrive already offset)
    DevState *st_out = F.st_out + agent;
    StepResult *res = reinterpret_cast<StepResult *>(reinterpret_cast<char *>(F.res) + (size_t)agent * F.res_stride);
    double *res_u = reinterpret_cast<double *>(res + 1);
    const A *u_in = reinterpret_cast<const A *>(u_pre);
    A *u_out = reinterpret_cast<A *>(F.u_out) + (size_t)agent * 2 * T;
    A *u_hist = reinterpret_cast<A *>(F.u_before) + (size_t)agent * 4 * T;
    const int *pout = F.pout + (size_t)agent * F.K;
    const void *partials = partials_pre;  // == F.partials + this agent's offset
    const A *ref = reinterpret_cast<const A *>(F.ref);

    STAMP(16);
    // ---- every load that nothing below produces, issued before anything waits: this thread's u, the
    // records (their addresses depend on nothing but the kernel arguments) and the state -- a single memory
    // round trip covers all of them
    // (the state as one VECTOR load, lane i <- dword i: vector loads return in issue order, so it arrives
    // before the records queued behind it; scalar loads would be issued after them and wait for the queue)
    static_assert(sizeof(DevState) == 72, "DevState layout");
    const int st_word = reinterpret_cast<const int *>(st_pre)[lane < 18 ? lane : 0];
    const A u_old = tid < 2 * T_pre ? u_in[tid] : A(0);  // elements >= NT: re-read below
    unsigned long long peer_ptr = 0;
    if (XCHG) peer_ptr = load_peer_ptrs(F);
    MergeRegs<A, NT, NWIN> mr;
    // (HYPK: every wave fetches the workgroups' look-back words as well, four per lane and load -- each wave reduces them
    // itself, which saves an exchange through LDS and a barrier)
    constexpr int HW = HYP_MAX_BLOCKS / 256;
    uint4 hwords[HW];
#pragma unroll
    for (int i = 0; i < HW; ++i) {
        hwords[i] = uint4{0, 0, 0, 0};
        if (HYPK && 4 * (lane + 64 * i) < F.hyp_blocks) hwords[i] = reinterpret_cast<const uint4 *>(F.hyp_slots)[lane + 64 * i];
    }
    if (!ABI_RECS) {
        merge_load_heads<A, NT, NWIN>(reinterpret_cast<const A *>(heads_pre), mr);
        merge_load_tile<A, NT, NWIN>(reinterpret_cast<const A *>(partials_pre), T_pre, 0, mr);
    }
    STAMP(24);
    DevState sv;
    {
        auto word = [&](int i) { return __builtin_amdgcn_readlane(st_word, i); };
        auto dbl = [&](int i) {
            return __builtin_bit_cast(double, ((long long)word(i + 1) << 32) | (unsigned int)word(i));
        };
        sv.x0[0] = dbl(0); sv.x0[1] = dbl(2); sv.x0[2] = dbl(4); sv.x0[3] = dbl(6);
        sv.p = word(8); sv.c = word(9); sv.k_start = word(10); sv.first_k = word(11);
        sv.round = word(12); sv.path_end = word(13); sv.idx_start = word(14); sv.pad = 0;
        sv.iter = ((long long)word(17) << 32) | (unsigned int)word(16);
    }
    const int fk = sv.first_k, round = sv.round;
    // first round of a synchronous step: the observed state and its x0 index came as kernel arguments
    const bool args = f_use_args && round == 0;
    const int c_state = args ? F.c_arg : sv.c;
    const int p_state = (args && !f_sequential) ? F.c_arg : sv.p;  // update_prev_idx=True at x0 (mppi_race_car.py:61)
    const int idx_start = args ? F.c_arg : sv.idx_start, path_end = args ? (F.c_arg >= F.n_ref - 1) : sv.path_end;
    const long long iter = sv.iter;
    const double x0v[4] = {args ? F.x0_arg[0] : sv.x0[0], args ? F.x0_arg[1] : sv.x0[1],
                           args ? F.x0_arg[2] : sv.x0[2], args ? F.x0_arg[3] : sv.x0[3]};
    // the state as this call found it, with the observed state folded in: what every exit starts from
    DevState nx = sv;
    nx.x0[0] = x0v[0]; nx.x0[1] = x0v[1]; nx.x0[2] = x0v[2]; nx.x0[3] = x0v[3];
    nx.c = c_state; nx.p = p_state; nx.idx_start = idx_start; nx.path_end = path_end;
    nx.first_k = NO_TRIGGER;
    auto publish = [&]() {  // completion word for a polling host: every result store first, system-wide
        __syncthreads();
        if (tid == 0 && F.seq) {
            __threadfence_system();
            *reinterpret_cast<volatile long long *>(&res->seq) = F.seq;
        }
    };
    // an exit that leaves the controls as they are: the state (and, when they go elsewhere, the controls) out
    auto leave = [&]() {
        if (tid == 0) *st_out = nx;
        if (u_out != u_in)
            for (int i = tid; i < 2 * T; i += NT) u_out[i] = i == tid ? u_old : u_in[i];
        publish();
    };

    if (XCHG && *reinterpret_cast<volatile int *>(F.x_err)) {  // an earlier exchange failed: do not wait again
        if (tid == 0) { res->status = STATUS_EXCHANGE_FAILED; res->iter = iter; }
        return leave();
    }
    // --- the sequential index resolved in the rollout launch (fused_lookback): the largest offset any workgroup saw ----
    int c_final = c_state;
    bool hyp_done = false;
    if (HYPK) {  // (launched with the HYPK rollout kernel: the same condition there)
        static_assert(!HYPK || (MODE == 0 && NWIN <= 2 && NT == MERGE_THREADS), "the resolution is part of the plain 256-thread finalize");
        const bool hyp_round = round == 0 && min(F.window, F.n_ref - c_state) > 1;
        if (hyp_round) {
            const unsigned tag = lb_tag(F.lb_seq);
            int mx = 0;
            bool bd = false;
#pragma unroll
            for (int i = 0; i < HW; ++i) {
                const unsigned hw[4] = {hwords[i].x, hwords[i].y, hwords[i].z, hwords[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (4 * (lane + 64 * i) + j < F.hyp_blocks) {
                        bd |= (hw[j] & LB_TAG_MASK) != tag || (hw[j] & LB_BAD) != 0u;
                        mx = max(mx, (int)(hw[j] & (LB_BAD - 1)));
                    }
                }
            }
            const int lb_max = wv::reduce<wv::OpMaxInt>(mx);
            const bool lb_bad = __ballot(bd) != 0ull;
            if (lb_bad) {
                // a call that was not unimodal, an index beyond the candidates' reach or a look-back that timed out: the
                // speculation rounds redo the iteration from its first sample
                nx.k_start = 0;
                nx.c = c_state;
                nx.round = round + 1;
                if (tid == 0) {
                    res->status = STATUS_NEED_ROUND;
                    res->k_next = 0;
                    res->c_next = c_state;
                    res->rounds = round + 1;
                    res->iter = iter;
                }
                return leave();
            }
            c_final = c_state + lb_max;
            hyp_done = true;
        }
    }
    // --- sequential-waypoint speculation: did a sample move the index? ---------------------
    if (!hyp_done && f_sequential && fk != NO_TRIGGER) {
        const int c_new = pout[fk];
        if (fk + 1 < F.K) {  // samples after fk were evaluated from a stale index: another round
            nx.k_start = fk + 1;
            nx.c = c_new;
            nx.round = round + 1;
            if (tid == 0) {
                res->status = STATUS_NEED_ROUND;
                res->k_next = fk + 1;
                res->c_next = c_new;
                res->rounds = round + 1;
                res->iter = iter;
            }
            return leave();
        }
        c_final = c_new;
    }
    const int p_now = f_sequential ? c_final : p_state;  // prev_way_point_idx after this iteration

    // wave 0 prefetches the candidates of the next x0 call so that they overlap the merge
    A cand_x = 0, cand_y = 0;
    const int wlen_next = min(F.window, F.n_ref - p_now);
    if (wid == 0 && f_plant && lane < wlen_next) {
        cand_x = ref[4 * (p_now + lane)];
        cand_y = ref[4 * (p_now + lane) + 1];
    }
    A sn_yaw = 0, cs_yaw = 1;  // the plant's trigonometry, while the loads are in flight
    if (f_plant) mf::sincos_((A)x0v[2], sn_yaw, cs_yaw);
    STAMP(17);

    // w_eps lands in the padded layout the filter reads (filter_store)
    const bool pad_copy = f_filter == FILTER_RACE || f_filter == FILTER_TORCH;
    auto store_w = [&](int i, A v) { filter_store<A>(sh_w, i, v, T, H, pad_copy); };
    A rho, eta, eta2;
    A n_hit = A(-1);  // (-1: not known -- the records of other ranks carry no count)
    const bool count_hits = F.count_hits != 0;  // (a handle with obstacles: the block records carry the count)
    if (ABI_RECS) {
        merge_abi<A>(reinterpret_cast<const double *>(F.partials), F.n_part, T, (A)F.beta, L.s, L.red, rho, eta, eta2,
                     store_w);
    } else if (!XCHG) {
        if (count_hits) n_hit = A(0);
        merge_combine<A, NT, NWIN>(reinterpret_cast<const A *>(partials), F.n_part, T, (A)F.beta, mr, L.s, L.part, rho,
                                   eta, eta2, store_w, &n_hit);
        if (!count_hits) n_hit = A(0);
    } else {
        // this rank's record {rho, eta, eta2, W} from its block records, stored into every rank's buffer
        merge_combine<A, NT, NWIN>(reinterpret_cast<const A *>(partials), F.n_part, T, (A)F.beta, mr, L.s, L.part, rho,
                                   eta, eta2, [&](int i, A v) { sh_u[i] = v; });
        const size_t slot_off = (size_t)(F.x_seq & 1) * xchg_slot_bytes(T, F.x_nranks);
        const size_t rec_off = slot_off + sizeof(long long) * XCHG_MAX_RANKS + sizeof(double) * (size_t)F.x_rank * xchg_rec_len(T);
        for (int p = 0; p < F.x_nranks; ++p) {
            double *rec = reinterpret_cast<double *>(peer_base(peer_ptr, p) + rec_off);
            for (int i = tid; i < 2 * T; i += NT) rec[3 + i] = (double)(sh_u[i] * eta);  // W itself
            if (tid == 0) { rec[0] = (double)rho; rec[1] = (double)eta; rec[2] = (double)eta2; }
        }
        // every wave's stores into the peers' buffers are acknowledged before the flags go up (the release store in
        // exchange_flags is made by wave 0 alone: across a link it orders that wave's own stores only)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (!exchange_flags(F, peer_ptr, slot_off)) {  // a peer never arrived: nothing is updated
            if (tid == 0) {
                res->status = STATUS_EXCHANGE_FAILED;
                res->rounds = round + 1; res->iter = iter;
            }
            return leave();
        }
        const double *recs = reinterpret_cast<const double *>(peer_base(peer_ptr, F.x_rank) + slot_off + sizeof(long long) * XCHG_MAX_RANKS);
        if (F.x_nranks <= XCHG_LDS_RANKS) {
            // the exchange buffer is fine-grained (uncached) memory: fetch all ranks' records with independent,
            // coalesced loads -- one memory round trip -- instead of one dependent load per rank in the merge loop
            double *sh_x = reinterpret_cast<double *>(smem + sizeof(A) * merge_lds_elems(T, W, sizeof(A), NT));
            const int tot = F.x_nranks * xchg_rec_len(T);
            for (int i = tid; i < tot; i += NT) sh_x[i] = recs[i];
            __syncthreads();
            recs = sh_x;
        }
        merge_abi<A>(recs, F.x_nranks, T, (A)F.beta, L.s, L.red, rho, eta, eta2, store_w, xchg_rec_len(T));
    }
    STAMP(18);

    if (f_raise && path_end) {  // mppi_race_car.py:63-65: nothing is updated (but the index was, :61)
        nx.k_start = 0;
        nx.round = 0;
        if (tid == 0) {
            res->status = STATUS_PATH_END;
            res->idx_start = idx_start; res->idx_after = p_state; res->path_end = 1;
            res->rounds = round + 1; res->iter = iter;
        }
        return leave();
    }

    // --- moving average of w_eps (window W) ------------------------------------------------------------------
    for (int i = tid; i < 2 * T; i += NT) {
        const int t = i >> 1, d = i & 1;
        const A f = filter_at<A>(sh_w, t, d, T, W, f_filter);
        const A uo = i == tid ? u_old : u_in[i];
        A un = uo + f;                                                        // u += w_epsilon, :141
        if (f_clamp_u) un = mf::clamp(un, d == 0 ? (A)F.umax0 : (A)F.umax1);  // :145-149
        sh_u[i] = un;
    }
    __syncthreads();
    STAMP(19);
    // --- shift (:162-163); the returned sequence aliases u_prev (:165) ---------------------
    for (int i = tid; i < 2 * T; i += NT) {
        const int t = i >> 1, d = i & 1;
        const A uo = i == tid ? u_old : u_in[i];
        const A shifted = sh_u[2 * (t < T - 1 ? t + 1 : T - 1) + d];
        u_hist[i] = uo;               // u before the update and the updated, unshifted u: viz rollouts
        u_hist[2 * T + i] = sh_u[i];
        res_u[i] = (double)shifted;
        u_out[i] = shifted;  // element i is read and written by this thread only
    }

    STAMP(20);
    if (wid == 0) {
        const A u0a = sh_u[2 * (T > 1 ? 1 : 0)], u0b = sh_u[2 * (T > 1 ? 1 : 0) + 1];
        double xn[4] = {x0v[0], x0v[1], x0v[2], x0v[3]};
        if (f_plant) {  // the driver's plant with the returned control
            if (f_model == MODEL_DIFF) {  // DifferentialDrive.update_state :33-40
                xn[0] += (double)(u0a * cs_yaw) * F.dt;
                xn[1] += (double)(u0a * sn_yaw) * F.dt;
                xn[2] += (double)u0b * F.dt;
            } else {  // Vehicle.update models/vehicle.py:85-114
                const A steer = mf::clamp(u0a, (A)F.umax0), accel = mf::clamp(u0b, (A)F.umax1);
                const double v = xn[3];
                xn[0] += v * (double)cs_yaw * F.dt;
                xn[1] += v * (double)sn_yaw * F.dt;
                xn[2] += v / F.wheel_base * (double)mf::tan_(steer) * F.dt;
                xn[3] += (double)accel * F.dt;
            }
        }
        nx.iter = iter + 1;
        nx.k_start = 0;
        nx.round = 0;
        nx.p = p_now;
        if (lane == 0) {
            res->status = STATUS_DONE;
            res->k_next = 0; res->c_next = c_final;
            res->idx_start = idx_start;
            res->idx_after = p_now;
            res->path_end = path_end;
            res->rounds = round + 1;
            res->costs_hyp = hyp_done ? 1 : 0;
            res->n_collided = (int)n_hit;
            res->rho = (double)rho; res->eta = (double)eta; res->ess = (double)(eta * eta / eta2);
            res->u0[0] = (double)u0a; res->u0[1] = (double)u0b;
            for (int q = 0; q < 4; ++q) res->x_next[q] = xn[q];
            if (f_trace && agent == 0) { F.u0_trace[2 * iter] = (double)u0a; F.u0_trace[2 * iter + 1] = (double)u0b; }
            res->iter = iter + 1;
        }
        if (f_plant) {  // next iteration's x0 call (:96-99), so the next slot needs no host input
            // in f64 whatever the handle's precision, like k_set_state and the host side of mppi_step: the closed loop on
            // the device and the same loop stepped from the host must take the same index at a near tie
            double best = INFINITY;
            int bj = INT_MAX;
            if (lane < wlen_next) {
                const double dx = xn[0] - (double)cand_x, dy = xn[1] - (double)cand_y;
                best = dx * dx + dy * dy;
                bj = lane;
            }
            for (int j = lane + 64; j < wlen_next; j += 64) {  // windows wider than a wave (race car: 200)
                const double dx = xn[0] - (double)ref[4 * (p_now + j)], dy = xn[1] - (double)ref[4 * (p_now + j) + 1];
                const double dd = dx * dx + dy * dy;
                if (dd < best) { best = dd; bj = j; }
            }
            wv::argmin_first(best, bj);
            const int c = p_now + bj;
            nx.x0[0] = xn[0]; nx.x0[1] = xn[1]; nx.x0[2] = xn[2]; nx.x0[3] = xn[3];
            nx.c = c;
            nx.idx_start = c;
            nx.path_end = c >= F.n_ref - 1;
            if (!f_sequential) nx.p = c;
        }
        if (lane == 0) *st_out = nx;
    }
    publish();
    STAMP(21);
}

template <typename A, int MODE, int NWIN, bool MULTI, bool PLAIN = false, bool HYPK = false>
__global__ __launch_bounds__(MERGE_THREADS) void k_finalize(const void *partials_pre, const void *heads_pre,
                                                            const DevState *st_pre, const void *u_pre, int T_pre,
                                                            const FinalizeParams F) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (MULTI) {  // several agents per launch: workgroup row blockIdx.y finishes agent blockIdx.y
        const int a = blockIdx.y;
        const size_t rec = (size_t)a * F.slots * record_len(T_pre, (int)sizeof(A));
        finalize_body<A, MODE, MERGE_THREADS, NWIN>(reinterpret_cast<const A *>(partials_pre) + rec,
                                                    reinterpret_cast<const A *>(heads_pre) + (size_t)a * F.slots * 4,
                                                    st_pre + a, reinterpret_cast<const A *>(u_pre) + (size_t)a * 2 * T_pre,
                                                    T_pre, F, smem, a);
    } else {
        finalize_body<A, MODE, MERGE_THREADS, NWIN, PLAIN, HYPK>(partials_pre, heads_pre, st_pre, u_pre, T_pre, F, smem, 0);
    }
}

// ------------------------------------------------------------------------------------------
// S1 materialised (`_calc_epsilon`), and the visualisation rollouts (:144-159)
// ------------------------------------------------------------------------------------------
__global__ void k_sample(unsigned seed_lo, unsigned seed_hi, unsigned iter, int K, int T, int k_offset, float l00,
                         float l10, float l11, float *__restrict__ eps, unsigned stream_word) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)K * T) return;
    const int k = (int)(i / T), t = (int)(i % T);
    const float chol[3] = {l00, l10, l11};
    float e0, e1;
    px::sample(seed_lo, seed_hi, iter, (unsigned)(k + k_offset), t, chol, e0, e1, stream_word);
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
                    px::sample(P.seed_lo, P.seed_hi, iter, (unsigned)(row + P.k_offset), tc, P.chol, e0, e1, (unsigned)P.noise_stream);
                } else {
                    const float2 e = *reinterpret_cast<const float2 *>(eps_tensor(P, iter, 0) + ((size_t)row * P.T + tc) * 2);
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
// The controller classes' stage methods, batched over n items (`_state_transition` / `_F`,
// `_compute_cost` / `_c`, `_terminal_cost` / `_phi`, `_is_collided`, `_get_nearest_waypoint`, `_g`,
// `_moving_average_filter`, `_compute_weight`) -- the same device functions the rollout kernels call, behind
// mppi_eval_* of the C ABI.  Inspection / test entry points, not a hot path.
// ------------------------------------------------------------------------------------------
// waypoint index of n calls: every call searches [p0, p0 + window) (`update_prev_idx=False`, mppi_race_car.py:157-174), or
// the index threads through the calls in order as n successive reference calls with `update_prev_idx=True` would
// (mppi_differential_drive.py:201-220); xy = the first two columns of rows of `stride` elements
template <typename R>
__global__ __launch_bounds__(64) void k_eval_index(const R *__restrict__ ref, int n_ref, int window, const R *__restrict__ xy,
                                                   int stride, int n, int p0, int sequential, int *__restrict__ idx_out,
                                                   int *__restrict__ p_out) {
    const int lane = threadIdx.x;
    if (sequential) {  // one wave, candidates over the lanes, calls one after the other
        int p = p0;
        for (int i = 0; i < n; ++i) {
            p = nearest_uniform(ref, p, window_len<R>(window, n_ref, p), xy[(size_t)i * stride], xy[(size_t)i * stride + 1], lane);
            if (lane == 0) idx_out[i] = p;
        }
        if (lane == 0) *p_out = p;
    } else {
        const int i = blockIdx.x * 64 + lane;
        if (i < n) idx_out[i] = nearest_in_window(ref, p0, window_len<R>(window, n_ref, p0), xy[(size_t)i * stride], xy[(size_t)i * stride + 1]);
        if (i == 0) *p_out = p0;
    }
}

template <typename R, int MODEL>
__global__ __launch_bounds__(256) void k_eval(const KParams<R> P, int what, const R *__restrict__ x, const R *__restrict__ v,
                                              const int *__restrict__ idx, int n, R *__restrict__ out) {
    constexpr int NX = MODEL == MODEL_RACE ? 4 : 3;
    const int lane = threadIdx.x & 63, i = blockIdx.x * blockDim.x + threadIdx.x;
    const ObsLanes<R> obs = load_obstacles(P, lane);  // (lane m holds circle m: all lanes of the wave take part)
    const bool act = i < n;
    R s[4] = {R(0), R(0), R(0), R(0)}, v0 = 0, v1 = 0;
    if (act && x)
        for (int q = 0; q < NX; ++q) s[q] = x[(size_t)i * NX + q];
    if (act && v) { v0 = v[2 * (size_t)i]; v1 = v[2 * (size_t)i + 1]; }
    if (what == EVAL_TRANSITION) {
        R sn, cs;
        mf::sincos_(s[2], sn, cs);
        if (MODEL == MODEL_DIFF) {  // mppi_differential_drive.py:182-198
            s[0] += v0 * cs * P.dt;
            s[1] += v0 * sn * P.dt;
            s[2] += v1 * P.dt;
        } else {  // mppi_race_car.py:183-197, controls = [steer, accel]
            const R vel = s[3];
            s[0] += vel * cs * P.dt;
            s[1] += vel * sn * P.dt;
            s[2] += vel / P.wheel_base * mf::tan_(v0) * P.dt;
            s[3] += v1 * P.dt;
        }
        if (act)
            for (int q = 0; q < NX; ++q) out[(size_t)i * NX + q] = s[q];
    } else if (what == EVAL_CLAMP) {  // `_g`
        if (act) {
            out[2 * (size_t)i] = mf::clamp(v0, P.umax0);
            out[2 * (size_t)i + 1] = mf::clamp(v1, P.umax1);
        }
    } else {
        const bool hit = collided<MODEL == MODEL_RACE>(P, s[0], s[1], s[2], obs);
        if (what == EVAL_COLLIDED) {
            if (act) out[i] = hit ? R(1) : R(0);
        } else {  // `_compute_cost` / `_c`, `_terminal_cost` / `_phi` at the given waypoint index
            const bool term = what == EVAL_COST_TERMINAL;
            R c = tracking_cost<R, MODEL>(P, term ? P.wt : P.ws, term ? P.wrap_term : P.wrap_stage, act ? idx[i] : 0, s[0], s[1],
                                          s[2], s[3]);
            if (hit) c += P.penalty;
            if (act) out[i] = c;
        }
    }
}

// `_moving_average_filter` of a [T,2] signal in the handle's filter mode (one workgroup)
template <typename A>
__global__ __launch_bounds__(256) void k_eval_filter(const A *__restrict__ xx, A *__restrict__ out, int T, int W, int f_filter) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    A *sh_w = reinterpret_cast<A *>(smem);
    const bool pad_copy = f_filter == FILTER_RACE || f_filter == FILTER_TORCH;
    for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) filter_store<A>(sh_w, i, xx[i], T, W / 2, pad_copy);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) out[i] = filter_at<A>(sh_w, i >> 1, i & 1, T, W, f_filter);
}

// `_compute_weight` of a given cost vector: w = exp(-beta (S - min S)) / sum (one workgroup, f64)
__global__ __launch_bounds__(256) void k_eval_weights(const double *__restrict__ S, int n, double beta, double *__restrict__ w) {
    __shared__ double sh[8];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double m = INFINITY;
    for (int i = tid; i < n; i += 256) m = fmin(m, S[i]);
    m = wv::reduce<wv::OpMin>(m);
    if (lane == 0) sh[wid] = m;
    __syncthreads();
    const double rho = fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
    double a = 0.0;
    for (int i = tid; i < n; i += 256) a += exp(-beta * (S[i] - rho));
    a = wv::reduce<wv::OpAdd>(a);
    if (lane == 0) sh[4 + wid] = a;
    __syncthreads();
    const double eta = sh[4] + sh[5] + sh[6] + sh[7];
    for (int i = tid; i < n; i += 256) w[i] = exp(-beta * (S[i] - rho)) / eta;
}

// x0 of a synchronous step handed over in device memory (mppi_step_device_x0): into the state, then the x0 call
template <typename R>
__global__ __launch_bounds__(64) void k_set_state_dev(const R *ref, int n_ref, int window, int sequential, DevState *st,
                                                      const double *__restrict__ x_dev, int nx) {
    const int lane = threadIdx.x;
    const double x0 = x_dev[0], x1 = x_dev[1];
    if (lane < 4) st->x0[lane] = lane < nx ? x_dev[lane] : 0.0;
    x0_call<R>(st, ref, n_ref, window, sequential, lane, x0, x1, st->p);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// Which instantiation the last rollout launch of this thread took, spelled as rocprofv3 prints it (bench.py looks its
// counters up by this name): one static string per launch site, a pointer store per launch.
static thread_local const char *g_last_rollout_kernel = "";
const char *last_rollout_kernel() { return g_last_rollout_kernel; }
template <typename R> static const char *type_name() { return sizeof(R) == 8 ? "double" : "float"; }
static std::string kernel_name(const char *fmt, ...) {
    char b[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(b, sizeof(b), fmt, ap);
    va_end(ap);
    return b;
}
#define MPPI_NOTE_KERNEL(...)                                       \
    do {                                                            \
        static const std::string _nm = kernel_name(__VA_ARGS__);    \
        g_last_rollout_kernel = _nm.c_str();                        \
    } while (0)
static const char *tf(bool b) { return b ? "true" : "false"; }

int reduce_blocks(int K, int traj_per_block) { return (K + traj_per_block - 1) / traj_per_block; }

template <typename R> void launch_set_state(const KParams<R> &P, const double *x, hipStream_t s) {
    const int agents = x ? 1 : (P.n_agents > 1 ? P.n_agents : 1);  // (an x0 passed by value: single agent only)
    const double z[4] = {0, 0, 0, 0};
    const double *v = x ? x : z;
    hipLaunchKernelGGL(k_set_state<R>, dim3(agents), dim3(64), 0, s, P.ref, P.n_ref, P.window, P.sequential, P.st, v[0],
                       v[1], v[2], v[3], x ? 1 : 0);
}

template <typename R> void launch_rollout(const KParams<R> &P, hipStream_t s) {
    const int waves_per_block = 4, blocks = (P.K + waves_per_block - 1) / waves_per_block;
    if (P.model == MODEL_DIFF) {
        MPPI_NOTE_KERNEL("k_rollout<%s, 0>", type_name<R>());
        hipLaunchKernelGGL((k_rollout<R, MODEL_DIFF>), dim3(blocks), dim3(64 * waves_per_block), 0, s, P.st, P);
    } else {
        MPPI_NOTE_KERNEL("k_rollout<%s, 1>", type_name<R>());
        hipLaunchKernelGGL((k_rollout<R, MODEL_RACE>), dim3(blocks), dim3(64 * waves_per_block), 0, s, P.st, P);
    }
}

bool fused_supported(int T) { return T <= 128; }
// Two samples per wave pays once there are enough samples to keep >= 4 waves per SIMD with it (K >= 8192):
// it issues ~45 % fewer instructions per sample but halves the number of waves, and at K = 4096 the
// rollout is latency-bound (2 waves per SIMD cannot cover the ~10-cycle dependent-issue latency; measured
// 5.7 us against 5.0 us for one sample per wave).  MPPI_DUAL=0/1 overrides for experiments.
// Several agents per launch (blockIdx.y) count together: what matters is the number of waves the LAUNCH brings (32 agents
// of K = 4096: 6.7e10 -> 1.0e11 trajectory-steps/s with two samples per wave, 4 agents: 4.4e10 -> 5.6e10).
static bool dual_layout(int K, int T, int n_agents) {
    if (T > 64) return false;
    if (const char *e = getenv("MPPI_DUAL")) return atoi(e) != 0;
    return (long long)K * n_agents >= 8192;
}
// 64 < T <= 128: one sample per wave, two steps per lane (k_rollout_dual<.., 1>) instead of two 64-step chunks
static bool pair_layout(int T) {
    if (T <= 64 || T > 128) return false;
    if (const char *e = getenv("MPPI_PAIR")) return atoi(e) != 0;
    return true;
}
// 64 < T <= 96, race car: two samples per wave, three steps per lane (k_rollout_tri).  MPPI_TRI=0 keeps the pair layout (A/B).
static bool tri_layout(int T, bool tri_ok) {
    if (!tri_ok || T <= 64 || T > 32 * TRI_STEPS) return false;
    if (const char *e = getenv("MPPI_TRI")) return atoi(e) != 0;
    return true;
}
int rollout_layout(int K, int T, int n_agents, int model, bool f64, bool per_rollout, bool tri_ok) {
    if (!per_rollout && model == MODEL_RACE && !f64 && n_agents <= 1 && tri_layout(T, tri_ok)) return LAYOUT_TRI;
    // (per-rollout index threading lives in the one-sample-per-wave kernels: Rollout::chunk)
    const int kind = per_rollout ? LAYOUT_FUSED : dual_layout(K, T, n_agents) ? LAYOUT_DUAL : pair_layout(T) ? LAYOUT_PAIR : LAYOUT_FUSED;
    if (kind == LAYOUT_FUSED) return kind;
    // k_rollout_dual's SEQ: two samples per (half-)wave once one sample each would need more than one workgroup per
    // CU, and at most 512 records after halving (what k_finalize merges directly).  Beyond that the longer live
    // ranges of the doubled body cost the rollout more than k_merge gains from fewer records: K = 65536 x T = 75
    // measured 158 us per iteration against 153 us with one pass.  The f32 diff-drive form with two samples per wave
    // fits 64 VGPRs, so two of its workgroups share a CU and overlap their phases: up to 512 workgroups it stays at one
    // pass (config 3: 21.7 -> 21.0 us per iteration).  MPPI_SEQ=1/2 overrides for experiments.
    const int blocks = fused_blocks(K, T, kind);
    const bool shares_cu = kind == LAYOUT_DUAL && model == MODEL_DIFF && !f64;
    bool twice = blocks > (shares_cu ? 512 : 256) && blocks <= 1024;
    if (const char *e = getenv("MPPI_SEQ")) twice = atoi(e) == 2;
    return kind | (twice ? LAYOUT_TWICE : 0);
}
int fused_blocks(int K, int T, int layout) {
    // (the pair layout: DUAL_WAVES = 16 = FUSED_WAVES samples per pass)
    const int kind = layout & LAYOUT_KIND;
    const int per_block = ((kind == LAYOUT_DUAL || kind == LAYOUT_TRI) ? DUAL_SAMPLES : FUSED_WAVES) * (layout & LAYOUT_TWICE ? 2 : 1);
    return (K + per_block - 1) / per_block;
}

// The streaming kernel serves: diff-drive, the two-samples-per-wave layout, frozen index, `S[k] =`, windows of <= 32 candidates.
// Batches of 32 samples per workgroup: as many as leave about 1024 workgroups (two rounds of two per CU), at most 16.
template <typename R> static int stream_passes(const KParams<R> &P) {
    if (P.model != MODEL_DIFF || P.sequential || P.accumulate || P.T > 64 || P.window > 32 ||
        (P.layout & LAYOUT_KIND) != LAYOUT_DUAL)
        return 0;
    static const bool off = getenv("MPPI_NO_STREAM") != nullptr;  // (A/B runs)
    if (off) return 0;
    const long long batches = (long long)((P.K + DUAL_SAMPLES - 1) / DUAL_SAMPLES) * (P.n_agents > 1 ? P.n_agents : 1);
    static const int forced = getenv("MPPI_STREAM_PASSES") ? atoi(getenv("MPPI_STREAM_PASSES")) : 0;  // (experiments)
    if (forced > 0) return forced;
    int np = (int)(batches / 1024);  // (measured, 32 agents of K = 4096: 2 / 4 / 8 / 16 batches -> 32.8 / 30.8 / 31.3 / 35.6 us)
    return np < 1 ? 1 : np > 16 ? 16 : np;
}
template <typename R> int fused_records(const KParams<R> &P) {
    const int np = stream_passes(P);
    if (np == 0) return fused_blocks(P.K, P.T, P.layout);
    const int batches = (P.K + DUAL_SAMPLES - 1) / DUAL_SAMPLES;
    return (batches + np - 1) / np;
}
template int fused_records<float>(const KParams<float> &);
template int fused_records<double>(const KParams<double> &);

template <typename R, int MODEL, bool MULTI> static void launch_fused_mm(const KParams<R> &P, R *partials, hipStream_t s) {
    if (const int np = stream_passes(P)) {
        const dim3 sgrid(fused_records(P), MULTI ? P.n_agents : 1);
#define MPPI_LAUNCH_STREAM(OBS_, PHILOX_)                                                                              \
    do {                                                                                                               \
        MPPI_NOTE_KERNEL("k_rollout_stream<%s, %s, %s, %s>", type_name<R>(), tf(MULTI), tf(OBS_), tf(PHILOX_));        \
        hipLaunchKernelGGL((k_rollout_stream<R, MULTI, OBS_, PHILOX_>), sgrid, dim3(64 * DUAL_WAVES), 0, s, P.st, P, partials, np); \
    } while (0)
        const bool obs = P.obstacle_model != OBS_NONE;
        if (P.use_philox) { if (obs) MPPI_LAUNCH_STREAM(true, true); else MPPI_LAUNCH_STREAM(false, true); }
        else { if (obs) MPPI_LAUNCH_STREAM(true, false); else MPPI_LAUNCH_STREAM(false, false); }
#undef MPPI_LAUNCH_STREAM
        return;
    }
    const dim3 grid(fused_blocks(P.K, P.T, P.layout), MULTI ? P.n_agents : 1);
    const bool twice = (P.layout & LAYOUT_TWICE) != 0;
    const bool race = MODEL == MODEL_RACE;
    // (the PLAIN instantiations of k_rollout_dual: every single-agent form, and for batched agents the one the default
    // layout takes -- two samples per wave, one pass)
    const bool plain_ok = P.use_philox && P.clamp_rollout && (bool)P.wrap_stage == race && (bool)P.wrap_term == race;
    const bool plain_dual = !MULTI && plain_ok;
    switch (P.layout & LAYOUT_KIND) {
#define MPPI_LAUNCH_DUAL(SPW_, SEQ_, PLAIN_)                                                                                  \
    do {                                                                                                                      \
        MPPI_NOTE_KERNEL("k_rollout_dual<%s, %d, %d, %s, %d, %s, false>", type_name<R>(), MODEL, SPW_, tf(MULTI), SEQ_, tf(PLAIN_)); \
        hipLaunchKernelGGL((k_rollout_dual<R, MODEL, SPW_, MULTI, SEQ_, PLAIN_>), grid, dim3(64 * DUAL_WAVES), 0, s, P.st, P, \
                           partials);                                                                                         \
    } while (0)
    case LAYOUT_DUAL:
        if (P.hyp && !MULTI && MODEL == MODEL_DIFF && !twice) {  // the sequential index resolved in the launch (LB)
            if constexpr (!MULTI && MODEL == MODEL_DIFF) {
                if (plain_ok) {
                    MPPI_NOTE_KERNEL("k_rollout_dual<%s, 0, 2, false, 1, true, true>", type_name<R>());
                    hipLaunchKernelGGL((k_rollout_dual<R, MODEL_DIFF, 2, false, 1, true, true>), grid, dim3(64 * DUAL_WAVES), 0, s, P.st, P, partials);
                } else {
                    MPPI_NOTE_KERNEL("k_rollout_dual<%s, 0, 2, false, 1, false, true>", type_name<R>());
                    hipLaunchKernelGGL((k_rollout_dual<R, MODEL_DIFF, 2, false, 1, false, true>), grid, dim3(64 * DUAL_WAVES), 0, s, P.st, P, partials);
                }
            }
        } else if (plain_dual) { if (twice) MPPI_LAUNCH_DUAL(2, 2, !MULTI); else MPPI_LAUNCH_DUAL(2, 1, !MULTI); }
        else if (MULTI && plain_ok && !twice) MPPI_LAUNCH_DUAL(2, 1, true);
        else { if (twice) MPPI_LAUNCH_DUAL(2, 2, false); else MPPI_LAUNCH_DUAL(2, 1, false); }
        break;
    case LAYOUT_TRI:
        if constexpr (sizeof(R) == 4 && MODEL == MODEL_RACE && !MULTI) {
#define MPPI_LAUNCH_TRI(PLAIN_, SHARE_)                                                                                   \
    do {                                                                                                                  \
        MPPI_NOTE_KERNEL("k_rollout_tri<%s, %s>", tf(PLAIN_), tf(SHARE_));                                                \
        hipLaunchKernelGGL((k_rollout_tri<PLAIN_, SHARE_>), grid, dim3(64 * DUAL_WAVES), 0, s, P.st, P, partials);        \
    } while (0)
            const bool share = grid.x > 256;  // (more than one workgroup per CU)
            if (plain_ok) { if (share) MPPI_LAUNCH_TRI(true, true); else MPPI_LAUNCH_TRI(true, false); }
            else { if (share) MPPI_LAUNCH_TRI(false, true); else MPPI_LAUNCH_TRI(false, false); }
#undef MPPI_LAUNCH_TRI
        }
        break;
    case LAYOUT_PAIR:
        if (plain_dual) { if (twice) MPPI_LAUNCH_DUAL(1, 2, !MULTI); else MPPI_LAUNCH_DUAL(1, 1, !MULTI); }
        else { if (twice) MPPI_LAUNCH_DUAL(1, 2, false); else MPPI_LAUNCH_DUAL(1, 1, false); }
        break;
#undef MPPI_LAUNCH_DUAL
    default:
        {
            const bool plain = P.obstacle_model == OBS_NONE && P.use_philox && P.clamp_rollout && !P.wrap_stage && !P.wrap_term &&
                               !P.per_rollout;
            const int spec = plain ? 2 : P.obstacle_model == OBS_NONE ? 1 : 0;
            const dim3 block(64 * FUSED_WAVES);
#define MPPI_LAUNCH_FUSED(NCH_, SPEC_)                                                                                  \
    do {                                                                                                                \
        MPPI_NOTE_KERNEL("k_rollout_fused<%s, %d, %d, %s, %d, false>", type_name<R>(), MODEL, NCH_, tf(MULTI), SPEC_);  \
        hipLaunchKernelGGL((k_rollout_fused<R, MODEL, NCH_, MULTI, SPEC_>), grid, block, 0, s, P.st, P, partials);      \
    } while (0)
#define MPPI_LAUNCH_FUSED_HYP(SPEC_)                                                                                        \
    do {                                                                                                                    \
        MPPI_NOTE_KERNEL("k_rollout_fused<%s, 0, 1, false, %d, true>", type_name<R>(), SPEC_);                              \
        hipLaunchKernelGGL((k_rollout_fused<R, MODEL_DIFF, 1, false, SPEC_, true>), grid, block, 0, s, P.st, P, partials);  \
    } while (0)
            if (P.hyp && P.T <= 64 && !MULTI && MODEL == MODEL_DIFF) {
                if (spec == 2) MPPI_LAUNCH_FUSED_HYP(2);
                else if (spec == 1) MPPI_LAUNCH_FUSED_HYP(1);
                else MPPI_LAUNCH_FUSED_HYP(0);
            } else if (P.T <= 64) {
                if (spec == 2) MPPI_LAUNCH_FUSED(1, 2);
                else if (spec == 1) MPPI_LAUNCH_FUSED(1, 1);
                else MPPI_LAUNCH_FUSED(1, 0);
            } else {
                if (spec == 2) MPPI_LAUNCH_FUSED(2, 2);
                else if (spec == 1) MPPI_LAUNCH_FUSED(2, 1);
                else MPPI_LAUNCH_FUSED(2, 0);
            }
#undef MPPI_LAUNCH_FUSED
#undef MPPI_LAUNCH_FUSED_HYP
        }
    }
}
template <typename R, int MODEL> static void launch_fused_m(const KParams<R> &P, R *partials, hipStream_t s) {
    if (P.n_agents > 1) launch_fused_mm<R, MODEL, true>(P, partials, s);
    else launch_fused_mm<R, MODEL, false>(P, partials, s);
}

template <typename R> void launch_rollout_fused(const KParams<R> &P, void *partials, hipStream_t s) {
    if (P.model == MODEL_DIFF) launch_fused_m<R, MODEL_DIFF>(P, (R *)partials, s);
    else launch_fused_m<R, MODEL_RACE>(P, (R *)partials, s);
}

template <typename R> void launch_reduce(const KParams<R> &P, void *partials, int n_blocks, hipStream_t s) {
    const size_t shmem = sizeof(R) * ((size_t)P.traj_per_block + 16 + 4 * 128);
    hipLaunchKernelGGL(k_reduce<R>, dim3(n_blocks), dim3(256), shmem, s, P, (R *)partials);
}

static size_t merge_lds(int T, int W, size_t elem) { return elem * merge_lds_elems(T, W, elem); }

template <typename R>
void launch_merge(const void *recs, const void *heads, int n, int group, int T, double beta, void *out, void *out_heads,
                  bool out_abi, hipStream_t s) {
    const int blocks = (n + group - 1) / group;
    if (out_abi)
        hipLaunchKernelGGL((k_merge<R, true>), dim3(blocks), dim3(MERGE_THREADS), merge_lds(T, 0, sizeof(R)), s,
                           (const R *)recs, (const R *)heads, n, group, T, (R)beta, out, (R *)nullptr);
    else
        hipLaunchKernelGGL((k_merge<R, false>), dim3(blocks), dim3(MERGE_THREADS), merge_lds(T, 0, sizeof(R)), s,
                           (const R *)recs, (const R *)heads, n, group, T, (R)beta, out, (R *)out_heads);
}

template <typename R> void launch_finalize(const FinalizeParams &F, bool abi_recs, hipStream_t s) {
    // (+ the staging area of the peer-to-peer exchange; merge_lds_elems is a multiple of 4 elements: 16-byte aligned)
    const size_t lds = merge_lds(F.T, F.filter_window, sizeof(R)) +
                       (F.x_nranks > 1 ? sizeof(double) * XCHG_LDS_RANKS * xchg_rec_len(F.T) : 0) +
                       (F.hyp ? (size_t)HYP_MAX_BLOCKS + 16 : 0);
    const DevState *st = F.st;
    const bool two = F.n_part > MERGE_MAX_RECORDS;  // (at most MERGE_MAX_WINDOWS * 256: the caller merges above that)
    const bool multi = !abi_recs && F.x_nranks <= 1 && F.n_agents > 1;  // one workgroup per agent
    const dim3 grid(1, multi ? F.n_agents : 1);
#define MPPI_FIN(MODE, NWIN, MULTI)                                                                                    \
    hipLaunchKernelGGL((k_finalize<R, MODE, NWIN, MULTI>), grid, dim3(MERGE_THREADS), lds, s, F.partials, F.heads, st,  \
                       (const void *)F.u, F.T, F)
    if (abi_recs) MPPI_FIN(1, 1, false);
    else if (F.x_nranks > 1) { if (two) MPPI_FIN(2, 2, false); else MPPI_FIN(2, 1, false); }
    else if (multi) { if (two) MPPI_FIN(0, 2, true); else MPPI_FIN(0, 1, true); }
    else if (two) {
        if (F.hyp) hipLaunchKernelGGL((k_finalize<R, 0, 2, false, false, true>), grid, dim3(MERGE_THREADS), lds, s, F.partials, F.heads, st,
                                      (const void *)F.u, F.T, F);
        else MPPI_FIN(0, 2, false);
    } else {
        const bool plain = F.sequential && F.plant && !F.use_args && !F.raise_at_path_end && !F.clamp_u && F.model == MODEL_DIFF &&
                           !F.u0_trace && F.filter_mode == FILTER_DIFF;
#define MPPI_FIN_SINGLE(PLAIN_, HYPK_)                                                                                  \
    hipLaunchKernelGGL((k_finalize<R, 0, 1, false, PLAIN_, HYPK_>), grid, dim3(MERGE_THREADS), lds, s, F.partials, F.heads, \
                       st, (const void *)F.u, F.T, F)
        if (F.hyp) { if (plain) MPPI_FIN_SINGLE(true, true); else MPPI_FIN_SINGLE(false, true); }
        else { if (plain) MPPI_FIN_SINGLE(true, false); else MPPI_FIN_SINGLE(false, false); }
#undef MPPI_FIN_SINGLE
    }
#undef MPPI_FIN
}

void launch_exchange_probe(const FinalizeParams &F, int *ok_out, hipStream_t s) {
    hipLaunchKernelGGL(k_exchange_probe, dim3(1), dim3(64), 0, s, F, ok_out);
}

template <typename R>
void launch_eval_index(const KParams<R> &P, const R *xy, int stride, int n, int p0, int sequential, int *idx_out, int *p_out,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_eval_index<R>, dim3(sequential ? 1 : (n + 63) / 64), dim3(64), 0, s, P.ref, P.n_ref, P.window, xy,
                       stride, n, p0, sequential, idx_out, p_out);
}
template <typename R>
void launch_eval(const KParams<R> &P, int what, const R *x, const R *v, const int *idx, int n, R *out, hipStream_t s) {
    const dim3 grid((n + 255) / 256), block(256);
    if (P.model == MODEL_DIFF) hipLaunchKernelGGL((k_eval<R, MODEL_DIFF>), grid, block, 0, s, P, what, x, v, idx, n, out);
    else hipLaunchKernelGGL((k_eval<R, MODEL_RACE>), grid, block, 0, s, P, what, x, v, idx, n, out);
}
template <typename R> void launch_eval_filter(const R *xx, R *out, int T, int W, int mode, hipStream_t s) {
    hipLaunchKernelGGL(k_eval_filter<R>, dim3(1), dim3(256), sizeof(R) * 2 * (size_t)(T + W + 2), s, xx, out, T, W, mode);
}
void launch_eval_weights(const double *S, int n, double beta, double *w, hipStream_t s) {
    hipLaunchKernelGGL(k_eval_weights, dim3(1), dim3(256), 0, s, S, n, beta, w);
}
template <typename R> void launch_set_state_dev(const KParams<R> &P, const double *x_dev, int nx, hipStream_t s) {
    hipLaunchKernelGGL(k_set_state_dev<R>, dim3(1), dim3(64), 0, s, P.ref, P.n_ref, P.window, P.sequential, P.st, x_dev, nx);
}

template <typename R> void launch_weights(const KParams<R> &P, double rho, double eta, double *w, hipStream_t s) {
    hipLaunchKernelGGL(k_weights<R>, dim3((P.K + 255) / 256), dim3(256), 0, s, P.S, P.K, (double)P.beta, rho, eta, w);
}

void launch_sample(unsigned seed_lo, unsigned seed_hi, unsigned iter, int K, int T, int k_offset, const float *chol,
                   float *eps_out, hipStream_t s, unsigned stream_word) {
    const size_t n = (size_t)K * T;
    hipLaunchKernelGGL(k_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed_lo, seed_hi, iter, K, T,
                       k_offset, chol[0], chol[1], chol[2], eps_out, stream_word);
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

#ifdef MPPI_STAMPS
extern "C" int mppi_debug_stamps(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (n < 128 ? n : 128));
}
#endif

#define INSTANTIATE(R)                                                                                    \
    template void launch_set_state<R>(const KParams<R> &, const double *, hipStream_t);                   \
    template void launch_rollout<R>(const KParams<R> &, hipStream_t);                                     \
    template void launch_reduce<R>(const KParams<R> &, void *, int, hipStream_t);                         \
    template void launch_rollout_fused<R>(const KParams<R> &, void *, hipStream_t);                       \
    template void launch_merge<R>(const void *, const void *, int, int, int, double, void *, void *, bool, hipStream_t);        \
    template void launch_finalize<R>(const FinalizeParams &, bool, hipStream_t);                          \
    template void launch_weights<R>(const KParams<R> &, double, double, double *, hipStream_t);           \
    template void launch_eval_index<R>(const KParams<R> &, const R *, int, int, int, int, int *, int *, hipStream_t); \
    template void launch_eval<R>(const KParams<R> &, int, const R *, const R *, const int *, int, R *, hipStream_t);   \
    template void launch_eval_filter<R>(const R *, R *, int, int, int, hipStream_t);                                 \
    template void launch_set_state_dev<R>(const KParams<R> &, const double *, int, hipStream_t);                     \
    template void launch_viz<R>(const KParams<R> &, const R *, const R *, long long, float *, float *, hipStream_t);
INSTANTIATE(float)
INSTANTIATE(double)

}  // namespace mppi
