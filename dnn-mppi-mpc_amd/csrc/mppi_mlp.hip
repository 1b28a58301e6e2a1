// BASELINE config 5: the MPPI rollout with LEARNED residual dynamics on the gfx950 matrix cores.
//
//   x_{t+1} = x_t + dt * ( f(x_t, v_t) + MLP([x_t, v_t]) ),   f = [v cos(yaw), v sin(yaw), w]
//   (test/bullet_differential_drive_dnn.py:79-92; MLP = train/train_diff_mlp.py:13-36:
//    Linear(5,512) -> 3 x [Linear(512,512) + tanh] -> Linear(512,3), first layer without activation)
//
// The recurrence is serial in t and dense in K, so the mapping differs from the analytic kernels: a
// workgroup (4 waves, one per SIMD, the whole CU) owns a tile of 64 samples and keeps its activations
// [64 x 512] f32 in LDS (132 KB of the CU's 160 KB) across all layers and all T steps; every wave owns
// 128 output columns of every layer as 2x4 accumulator tiles of v_mfma_f32_32x32x2_f32 (exact f32, the
// rate of the f32 vector unit but one VGPR per operand).  Weights (3.2 MB, L2-resident) are streamed as
// fully coalesced 16-byte-per-lane loads from a fragment-ordered copy packed once on the host: for the
// k-group g (8 reduction indices) lane l of a wave reads W[n = 32 ct + (l & 31)][8 g + 4 (l >> 5) + s],
// s = 0..3, which are the B fragments of four consecutive MFMA k-steps; the A fragments of the same four
// steps are one ds_read_b128 of the activation row (row pitch 516 floats: conflict-free for 16-lane
// groups).  Per k-group: 2 LDS reads + 4 global loads feed 32 MFMAs (2048 cycles), so the loop is
// MFMA-bound; the next group's operands are loaded before the current group's MFMAs issue.
// 1 581 056 flop per trajectory-step.
#include <stdlib.h>
#include <string.h>

#include "mppi_device.h"

namespace mppi {

constexpr int MLP_M = 64, MLP_H = 512, MLP_PITCH = 516, MLP_WAVES = 4, MLP_GROUPS = MLP_H / 8;

using f32x16 = __attribute__((ext_vector_type(16))) float;
struct alignas(16) F4 { float v[4]; };

// Wave 0 searches the nearest waypoint of 64 samples every step while the other waves wait for it: the path (x, y, yaw, v
// per waypoint) is copied to LDS once per launch when it fits the 16 KB left beside the activations, so the search reads
// LDS instead of walking global memory (20 dependent loads per call, 5 % of the launch).
constexpr int MLP_REF_LDS_MAX = 768;  // (12 KB: what the 8-wave form of the f16-split kernel leaves beside its buffers)

// the path in LDS (returns the parameter block the per-sample code should use)
constexpr int H3_REF_LDS_32 = 256;  // the same beside a 32-sample tile (two workgroups share the CU's LDS)
__device__ __forceinline__ KParams<float> mlp_stage_path(const KParams<float> &P, float *ref_lds, int cap = MLP_REF_LDS_MAX) {
    KParams<float> PL = P;
    if (P.n_ref <= cap) {
        for (int i = threadIdx.x; i < P.n_ref; i += blockDim.x)
            reinterpret_cast<F4 *>(ref_lds)[i] = reinterpret_cast<const F4 *>(P.ref)[i];
        PL.ref = ref_lds;  // (first read behind the first workgroup barrier of the step loop)
    }
    return PL;
}

__device__ __forceinline__ float fast_tanh(float x) {  // 1 - 2 / (exp(2x) + 1): abs error ~2e-7
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);  // (v_rcp_f32, 1 ulp; `__fdividef` is the 10-instruction IEEE division here)
}

// acc[rt][ct] += A[rt] (64 x 8 slab of the activations) * B[ct] for one k-group
__device__ __forceinline__ void mfma_group(f32x16 (&acc)[2][4], const F4 (&a)[2], const F4 (&b)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt].v[s], b[ct].v[s], acc[rt][ct], 0, 0, 0);
}

// One Linear(n_groups*8 -> 512) of the tile: acc = act[64, :] @ W^T, this wave's 128 columns.
// `a_base` points at the A matrix in LDS (row pitch `pitch` floats), `wp` at this layer's packed weights.
__device__ __forceinline__ void gemm_layer(f32x16 (&acc)[2][4], const float *a_base, int pitch, const float *wp,
                                           int n_groups, int wid, int lane) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
    const float *arow0 = a_base + (lane & 31) * pitch + 4 * (lane >> 5);
    const float *arow1 = arow0 + 32 * pitch;
    const float *wl = wp + ((size_t)(wid * 4) * n_groups * 64 + lane) * 4;  // + (ct * n_groups + g) * 256
    F4 a0[2], b0[4], a1[2], b1[4];
    auto load = [&](int g, F4 (&a)[2], F4 (&b)[4]) {
        a[0] = *reinterpret_cast<const F4 *>(arow0 + 8 * g);
        a[1] = *reinterpret_cast<const F4 *>(arow1 + 8 * g);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) b[ct] = *reinterpret_cast<const F4 *>(wl + ((size_t)ct * n_groups + g) * 256);
    };
    load(0, a0, b0);
    for (int g = 0; g < n_groups; g += 2) {  // two groups per trip: static register names for the prefetch
        if (g + 1 < n_groups) load(g + 1, a1, b1);
        mfma_group(acc, a0, b0);
        if (g + 1 < n_groups) {
            if (g + 2 < n_groups) load(g + 2, a0, b0);
            mfma_group(acc, a1, b1);
        }
    }
}

// bias (+ tanh), then this wave's [64 x 128] slice of the activations back to LDS.
// C/D layout of 32x32 tiles: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
template <bool TANH>
__device__ __forceinline__ void store_layer(float *act, const f32x16 (&acc)[2][4], const float *bias, int wid, int lane) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int n = wid * 128 + ct * 32 + (lane & 31);
        const float bn = bias[n];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = acc[rt][ct][r] + bn;
                act[row * MLP_PITCH + n] = TANH ? fast_tanh(v) : v;
            }
    }
}

// ---- what the lanes of wave 0 (one lane = one sample of the tile) do around the network, shared by both kernels ----
struct MlpLane {
    float x, y, yaw, S;
    int p;
};

// this sample's controls at step t: noise (Philox or the caller's tensor), perturb, clamp (:116-121)
__device__ __forceinline__ void mlp_controls(const KParams<float> &P, unsigned iter, int k, int t, bool valid, bool exploit,
                                             float &u0, float &u1, float &v0, float &v1) {
    float e0 = 0.f, e1 = 0.f;
    if (valid) {
        if (P.use_philox) px::sample(P.seed_lo, P.seed_hi, iter, (unsigned)(k + P.k_offset), t, P.chol, e0, e1, (unsigned)P.noise_stream);
        else {
            const float2 e = *reinterpret_cast<const float2 *>(eps_tensor(P, iter, 0) + ((size_t)k * P.T + t) * 2);
            e0 = e.x;
            e1 = e.y;
        }
    }
    u0 = P.u[2 * t];
    u1 = P.u[2 * t + 1];
    v0 = exploit ? u0 + e0 : e0;  // mppi_differential_drive.py:116-119
    v1 = exploit ? u1 + e1 : e1;
    if (P.clamp_rollout) {
        v0 = mf::clamp(v0, P.umax0);
        v1 = mf::clamp(v1, P.umax1);
    }
}

// Euler step of f + MLP (bullet_differential_drive_dnn.py:79-92)
__device__ __forceinline__ void mlp_euler(const KParams<float> &P, float r0, float r1, float r2, float v0, float v1, MlpLane &L) {
    float sn, cs;
    mf::sincos_(L.yaw, sn, cs);
    L.x = L.x + P.dt * (v0 * cs + r0);
    L.y = L.y + P.dt * (v0 * sn + r1);
    L.yaw = L.yaw + P.dt * (v1 + r2);
}

// The visualisation rollouts with this model (mppi_differential_drive.py:144-159; the reference never wires the learned
// model into MPPI, so this is the analytic controllers' loop with the transition swapped, like the rollout itself): step t
// is driven by control (t-1) mod T -- the reference's `[t-1]` indexing --, always clamped (:148,:158); rows 0..K-1 the
// samples' perturbed controls of the iteration just finished (its noise regenerated), one extra row the updated nominal
// sequence.  `VIZ` instantiation of the rollout kernel: same network code, no costs, the states stored instead.
struct MlpViz {
    const float *u_before, *u_upd;  // [T][2]: nominal controls before the update / updated, unshifted
    float *opt, *smp;               // [T][3], [K][T][3] (either may be null)
    unsigned iter;                  // the iteration whose noise the samples used
    int block0;                     // first workgroup of the launch (the samples' workgroups are left out without `smp`)
    // `_state_transition` of the learned model, batched (mppi_eval_state_transition): row k takes ONE Euler step from
    // ex[k] under the control ev[k] as it is given
    const float *ex, *ev;           // [en][3], [en][2]
    float *eout;                    // [en][3]
    int en;
};
__device__ __forceinline__ void mlp_controls_viz(const KParams<float> &P, const MlpViz &V, int k, int t, bool sample_row,
                                                 bool opt_row, bool exploit, float &v0, float &v1) {
    const int tc = (t + P.T - 1) % P.T;
    v0 = 0.f;
    v1 = 0.f;
    if (opt_row) {
        v0 = V.u_upd[2 * tc];
        v1 = V.u_upd[2 * tc + 1];
    } else if (sample_row) {
        float e0, e1;
        if (P.use_philox) px::sample(P.seed_lo, P.seed_hi, V.iter, (unsigned)(k + P.k_offset), tc, P.chol, e0, e1, (unsigned)P.noise_stream);
        else {
            const float2 e = *reinterpret_cast<const float2 *>(eps_tensor(P, V.iter, 0) + ((size_t)k * P.T + tc) * 2);
            e0 = e.x;
            e1 = e.y;
        }
        v0 = exploit ? V.u_before[2 * tc] + e0 : e0;
        v1 = exploit ? V.u_before[2 * tc + 1] + e1 : e1;
    }
    v0 = mf::clamp(v0, P.umax0);
    v1 = mf::clamp(v1, P.umax1);
}

// this call's waypoint index and costs behind the Euler step
__device__ __forceinline__ void mlp_advance(const KParams<float> &P, const ObsLanes<float> &obs, int c, int t, float r0, float r1,
                                            float r2, float u0, float u1, float v0, float v1, MlpLane &L) {
    const float *__restrict__ ref = P.ref;
    mlp_euler(P, r0, r1, r2, v0, v1, L);
    // waypoint index of this call: per-lane, so the sequential index threads through the sample's own
    // calls in order by construction (mppi_differential_drive.py:228)
    auto nearest = [&](int from) {
        float best = dist2(ref, from, L.x, L.y);
        int bj = 0;
#pragma unroll 4
        for (int j = 1; j < P.window; ++j) {
            const bool ok = from + j < P.n_ref;
            const float d = ok ? dist2(ref, min(from + j, P.n_ref - 1), L.x, L.y) : INFINITY;
            if (d < best) { best = d; bj = j; }
        }
        return from + bj;
    };
    const bool threads = P.sequential || P.per_rollout;  // the index moves with this sample's calls (:228)
    const int idx = nearest(threads ? L.p : c);
    if (threads) L.p = idx;
    if (P.accumulate || t == P.T - 1) {
        const bool hit = collided<false>(P, L.x, L.y, L.yaw, obs);
        float st_c = tracking_cost<float, MODEL_DIFF>(P, P.ws, P.wrap_stage, idx, L.x, L.y, L.yaw, 0.f);
        if (hit) st_c += P.penalty;
        const float ctrl = (u0 * P.sinv[0] + u1 * P.sinv[2]) * v0 + (u0 * P.sinv[1] + u1 * P.sinv[3]) * v1;
        const float stage = st_c + P.gamma * ctrl;
        L.S = P.accumulate ? L.S + stage : stage;
        if (t == P.T - 1) {
            int idx_term = idx;
            if (threads) {  // the terminal call moves the index once more (:244)
                L.p = nearest(L.p);
                idx_term = L.p;
            }
            float term = tracking_cost<float, MODEL_DIFF>(P, P.wt, P.wrap_term, idx_term, L.x, L.y, L.yaw, 0.f);
            if (hit) term += P.penalty;
            L.S += term;
        }
    }
}

// this tile's softmin record {rho, eta, eta2, pad, W[T][2]} (wave 0)
__device__ __forceinline__ void mlp_record(const KParams<float> &P, float *__restrict__ partials, unsigned iter, int k, int c,
                                           bool valid, bool live, MlpLane &L, int lane) {
    float S = L.S;
    if (live) {
        P.S[k] = S;
        P.pout[k] = L.p;
        if (P.sequential && L.p != c) atomicMin(&P.st->first_k, k);
    } else if (valid) {
        S = P.S[k];  // final from an earlier speculation round
    }
    const float Sm = valid ? S : INFINITY;
    const float rho = wv::reduce<wv::OpMin>(Sm);
    const float e = valid ? mf::exp_(-P.beta * (S - rho)) : 0.f;
    const float eta = wv::reduce<wv::OpAdd>(e), eta2 = wv::reduce<wv::OpAdd>(e * e);
    const float n_hit = P.obstacle_model != OBS_NONE ? wv::reduce<wv::OpAdd>(valid && S >= P.penalty ? 1.f : 0.f) : 0.f;
    float *out = partials + (size_t)blockIdx.x * record_len(P.T, 4);
    if (lane == 0) {
        out[0] = rho; out[1] = eta; out[2] = eta2;
        float *hd = P.heads + 4 * (size_t)blockIdx.x;  // the compact copy the merge kernels read
        hd[0] = rho; hd[1] = eta; hd[2] = eta2; hd[3] = n_hit;  // (n_hit: samples that carry a collision penalty)
    }
    for (int t = 0; t < P.T; ++t) {  // second pass over this tile's noise rows (regenerated / re-read)
        float e0 = 0.f, e1 = 0.f;
        if (valid) {
            if (P.use_philox) px::sample(P.seed_lo, P.seed_hi, iter, (unsigned)(k + P.k_offset), t, P.chol, e0, e1, (unsigned)P.noise_stream);
            else {
                const float2 ee = *reinterpret_cast<const float2 *>(eps_tensor(P, iter, 0) + ((size_t)k * P.T + t) * 2);
                e0 = ee.x;
                e1 = ee.y;
            }
        }
        const float w0 = wv::reduce<wv::OpAdd>(e * e0), w1 = wv::reduce<wv::OpAdd>(e * e1);
        if (lane == 0) { out[4 + 2 * t] = w0; out[4 + 2 * t + 1] = w1; }
    }
}

__global__ __launch_bounds__(64 * MLP_WAVES, 1) void k_rollout_mlp(const KParams<float> P, const MlpParams Q,
                                                                    float *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *act = smem;                       // [64][516]
    float *zbuf = act + MLP_M * MLP_PITCH;   // [64][8]   layer-0 input rows {x, y, yaw, v, w, 0, 0, 0}
    float *ypart = zbuf + MLP_M * 8;         // [4][64][4] partial outputs of the last Linear per wave
    float *ref_lds = ypart + MLP_WAVES * MLP_M * 4;  // [n_ref][4] when the path fits
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int k0 = blockIdx.x * MLP_M, k = k0 + lane;
    const KParams<float> PL = mlp_stage_path(P, ref_lds);
    const DevState sv = load_state(P, P.st);
    const ObsLanes<float> obs = load_obstacles(P, lane);
    if (k0 + MLP_M <= sv.k_start) return;  // every sample of the tile is final: its record stands
    const bool valid = k < P.K, live = valid && k >= sv.k_start;
    const int c = sv.c;
    const unsigned iter = (unsigned)sv.iter;

    // sample state lives in the lanes of wave 0 (one lane = one sample of the tile)
    MlpLane L{(float)sv.x0[0], (float)sv.x0[1], (float)sv.x0[2], 0.f, c};
    const bool exploit = (k + P.k_offset) < P.n_exploit;
    f32x16 acc[2][4];

    for (int t = 0; t < P.T; ++t) {
        float u0 = 0, u1 = 0, v0 = 0, v1 = 0;
        if (wid == 0) {
            mlp_controls(P, iter, k, t, valid, exploit, u0, u1, v0, v1);
            F4 z0 = {{L.x, L.y, L.yaw, v0}}, z1 = {{v1, 0.f, 0.f, 0.f}};
            *reinterpret_cast<F4 *>(zbuf + lane * 8) = z0;
            *reinterpret_cast<F4 *>(zbuf + lane * 8 + 4) = z1;
        }
        __syncthreads();
        // input_layer: Linear(5 -> 512), no activation (train/train_diff_mlp.py:32)
        gemm_layer(acc, zbuf, 8, Q.w_in, 1, wid, lane);
        store_layer<false>(act, acc, Q.b_in, wid, lane);
        __syncthreads();
        // hidden_layer[i]: tanh(Linear(512 -> 512)) (:33-34)
        for (int l = 0; l < Q.n_hidden; ++l) {
            gemm_layer(acc, act, MLP_PITCH, Q.w_h[l], MLP_GROUPS, wid, lane);
            __syncthreads();  // every wave has read the previous activations
            store_layer<true>(act, acc, Q.b_h[l], wid, lane);
            __syncthreads();
        }
        // out_layer: Linear(512 -> 3) (:35): lane = sample, this wave's 128 of the 512 inputs
        {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            const float *hrow = act + lane * MLP_PITCH + wid * 128;
            const float *w = Q.w_out + wid * 128;
#pragma unroll 8
            for (int n = 0; n < 128; n += 4) {
                const F4 h = *reinterpret_cast<const F4 *>(hrow + n);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s0 = fmaf(h.v[q], w[n + q], s0);
                    s1 = fmaf(h.v[q], w[MLP_H + n + q], s1);
                    s2 = fmaf(h.v[q], w[2 * MLP_H + n + q], s2);
                }
            }
            F4 o = {{s0, s1, s2, 0.f}};
            *reinterpret_cast<F4 *>(ypart + (wid * MLP_M + lane) * 4) = o;
        }
        __syncthreads();
        if (wid == 0) {
            float r0 = Q.b_out[0], r1 = Q.b_out[1], r2 = Q.b_out[2];
#pragma unroll
            for (int w = 0; w < MLP_WAVES; ++w) {
                const F4 o = *reinterpret_cast<const F4 *>(ypart + (w * MLP_M + lane) * 4);
                r0 += o.v[0];
                r1 += o.v[1];
                r2 += o.v[2];
            }
            mlp_advance(PL, obs, c, t, r0, r1, r2, u0, u1, v0, v1, L);
        }
        // zbuf / ypart are rewritten only after the next barrier sequence: wave 0 writes zbuf at the top of
        // the next step while the others wait at that step's first barrier
    }
    if (wid == 0) mlp_record(P, partials, iter, k, c, valid, live, L, lane);
}

// ------------------------------------------------------------------------------------------
// The same rollout on the f16 matrix pipe, 16 x the rate of the f32-input MFMA, at f32-like accuracy: every operand
// is split into two f16 numbers, a = a_hi + a_lo (a_hi = f16(a), a_lo = f16(a - a_hi): 22 bits of significand, and
// below f16's normal range an absolute resolution of 2^-25), and a b is taken as a_hi b_hi + a_hi b_lo + a_lo b_hi --
// three `v_mfma_f32_32x32x16_f16` into the SAME f32 accumulator (every product of two f16 numbers is exact in f32).
// Relative error of a dot product ~2^-21 against ~2^-23 of the f32 chain; the parity tests (u within 1e-4 RMSE of the
// reference's class with the real checkpoint) hold with either kernel.  MPPI_MLP_F32=1 selects the f32-input kernel.
//
// Layout: activations as two f16 planes [64][520] in LDS (row pitch 1040 B: the 16-byte fragment reads of a 16-lane
// group fall into distinct banks); lane l of a wave (r = l & 31, h = l >> 5) reads A[row r][k = 16 s + 8 h + 0..7] for
// k-step s with one ds_read_b128 per plane and row tile.  Weights are packed on the host in fragment order
// [column tile][k-step][lane][8] per plane, so B[k = 16 s + 8 h + j][col = 32 ct + r] = W[col][k] is one coalesced
// 16-byte load per lane.  Per k-step and wave: 4 LDS reads + 8 global loads feed 24 MFMAs (768 cycles).
// ------------------------------------------------------------------------------------------
using half8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int H3_PITCH = 520, H3_ZPITCH = 24, H3_STEPS = MLP_H / 16;
constexpr int H3_FORM_DEFAULT = 1;  // (H3_FORM_8x64, see h3_form)

__device__ __forceinline__ void split_h3(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// The f16 range.  The input layer has no activation (train/train_diff_mlp.py:32), so raw [x, y, yaw, v, w] and the first
// layer's output h0 = W_in z + b_in travel through f16 halves at whatever magnitude the caller's coordinates have (a path in
// UTM metres: 1e5 .. 1e7): beyond 65504 a half is inf.  Per sample and step two powers of two keep every half below 2^15:
// z is split as z / s0, s0 = 2^max(0, exponent(|z|_inf) - 14), and h0 as h0 / s1 with s1 from the bound
// |h0|_inf <= in_gain |z|_inf + in_bias (host, mppi_set_mlp); the epilogues multiply the f32 accumulators back
// (store_layer_h3).  Exact scalings: only the absolute resolution of the low halves (2^-25) grows with them.  After the first
// tanh everything lies in [-1, 1].  s0 = s1 = 1 whenever |z|_inf < 2^15 and the bound stays below 2^15.
struct H3Scale { float s0, inv_s0, s1, inv_s1; };
__device__ __forceinline__ H3Scale h3_scale(const float (&z)[5], float in_gain, float in_bias) {
    const float zmax = fmaxf(fmaxf(fmaxf(fabsf(z[0]), fabsf(z[1])), fmaxf(fabsf(z[2]), fabsf(z[3]))), fabsf(z[4]));
    auto pow2_over = [](float m) {  // e = max(0, exponent(m) - 14): m / 2^e < 2^15
        const int e = max(0, (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127 - 14);
        return min(e, 100);  // (inf / NaN inputs stay what they are)
    };
    const int e0 = pow2_over(zmax), e1 = pow2_over(in_gain * zmax + in_bias);
    auto p2 = [](int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); };
    return H3Scale{p2(e0), p2(-e0), p2(e1), p2(-e1)};
}

// acc[pass][rt][c2] = (A[64, 16 n_steps] @ W^T)^T for this wave's 128 columns (column tile ct = 2 pass + c2), i.e. the MFMA's
// "A" operand is the WEIGHT fragment and its "B" operand the activation fragment (both index a row / column by lane & 31
// and eight reduction indices by lane >> 5, so the fragments serve either role as they are): in the 32 x 32 result a lane
// then holds ONE sample (lane & 31 of row tile rt) and 16 output features in four runs of four consecutive ones -- which is
// what the epilogue wants to write: four f16 numbers = one 8-byte LDS store instead of four 2-byte ones (the 2-byte stores
// of the sample-per-register form held the LDS write port for a fifth of the launch).  a_hi / a_lo:
// the planes in LDS (row pitch `pitch` halfs).  Two passes of two column tiles each: the weights come from L2 (1 MB per
// layer and workgroup, the same MB for every workgroup), at 8 KB per k-step and wave their latency is what the loop has
// to cover -- so B runs THREE k-steps ahead in a ring of four register sets, and with two column tiles per pass that ring
// is 64 VGPRs (with four it spilled); the activations come from LDS just in time, twice per layer.
//
// The weight stream does not stop between the two passes of a layer: the last trip of the first pass requests the first
// three k-steps of the second into the ring sets it has just used up (an L2 round trip waited for at the top of every
// pass before).  Across layers the ring is not kept THROUGH the epilogue (that pushed the kernel into scratch spills, 8.3
// against 6.8 ms): the next layer's first k-steps are requested right behind the epilogue (h3_prime_layer).
struct H3FragA { half8 h[2], l[2]; };
struct H3FragB { half8 h[2], l[2]; };
struct H3Ring { H3FragB b0, b1, b2; };  // k-steps 0, 1, 2 of the pass about to run

// this lane's view of the two column tiles of (layer `w`, pass): + (c2 * n_steps + s) * 64
struct H3Pass { const half8 *hi, *lo; };
// NW waves per workgroup (4: one per SIMD, each owns 128 output columns = two passes of two column tiles; 8: two per SIMD,
// each owns 64 columns = one pass -- the same MFMAs, weight bytes and LDS reads per SIMD, but the two waves of a SIMD cover
// each other's waits in the GEMMs and share the VALU in the epilogues, which a single wave issues at half the pipe's rate)
template <int NW> __device__ __forceinline__ H3Pass h3_pass(const unsigned short *layer, int n_steps, int wid, int pass, int lane) {
    const half8 *w = reinterpret_cast<const half8 *>(layer);
    const size_t off = (size_t)(wid * (16 / NW) + 2 * pass) * n_steps * 64 + lane;
    return H3Pass{w + off, w + (size_t)16 * n_steps * 64 + off};
}
__device__ __forceinline__ void h3_load_b(const H3Pass &w, int n_steps, int s, H3FragB &f) {
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
        f.h[c2] = w.hi[((size_t)c2 * n_steps + s) * 64];
        f.l[c2] = w.lo[((size_t)c2 * n_steps + s) * 64];
    }
}
__device__ __forceinline__ void h3_prime(const H3Pass &w, int n_steps, H3Ring &r) {
    h3_load_b(w, n_steps, 0, r.b0);
    h3_load_b(w, n_steps, 1, r.b1);
    h3_load_b(w, n_steps, 2, r.b2);
}
// (the three terms as three sweeps over the four tiles: independent accumulators between two MFMAs on the same one)
// RT = row tiles of 32 samples a workgroup owns (2: the 64-sample tile; 1: a 32-sample tile, two workgroups per CU)
// TERMS = 2 (opt-in, MPPI_MLP_TERMS=2): without the a_lo b_hi term, i.e. with the activations rounded to f16 -- a third
// less matrix work (config 5: 6.5 -> 5.0 ms per iteration); u stays within 1e-7 RMSE of the reference, S within 3e-3
// relative, which is outside the 1e-3 the default is held to (see test_config5_two_term_split_is_an_opt_in).
template <int RT, int TERMS = 3> __device__ __forceinline__ void h3_mma(f32x16 (&ac)[RT][2], const H3FragA &a, const H3FragB &b) {
    if (TERMS == 3) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) ac[rt][c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h[c2], a.l[rt], ac[rt][c2], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) ac[rt][c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.l[c2], a.h[rt], ac[rt][c2], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) ac[rt][c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h[c2], a.h[rt], ac[rt][c2], 0, 0, 0);
}
template <int RT> __device__ __forceinline__ void h3_zero(f32x16 (&ac)[RT][2]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int r = 0; r < 16; ++r) ac[rt][c2][r] = 0.f;
}

// one pass (two column tiles) of a 512-wide hidden layer: `ring` holds its k-steps 0..2 on entry and (HAS_NEXT) those of
// `next` on exit
template <bool HAS_NEXT, int RT, int TERMS = 3>
__device__ __forceinline__ void gemm_pass_h3(f32x16 (&ac)[RT][2], const _Float16 *a_hi, const _Float16 *a_lo, const H3Pass &w,
                                             const H3Pass &next, H3Ring &ring, int lane) {
    constexpr int n_steps = H3_STEPS, pitch = H3_PITCH;
    const int aoff = (lane & 31) * pitch + 8 * (lane >> 5);
    auto load_a = [&](int s, H3FragA &f) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f.h[rt] = *reinterpret_cast<const half8 *>(a_hi + rt * 32 * pitch + aoff + 16 * s);
            f.l[rt] = *reinterpret_cast<const half8 *>(a_lo + rt * 32 * pitch + aoff + 16 * s);
        }
    };
    // (`sched_barrier`: the instruction scheduler would otherwise sink every load to just above its first use -- fewer
    // live registers, and the prefetch gone)
#define H3_FENCE() __builtin_amdgcn_sched_barrier(0)
    // One k-step = 12 MFMAs (384 cycles of the matrix pipe) + the requests that keep the ring and the activation fragments
    // ahead: 4 global loads + 4 LDS reads.  Issued as a block in front of the MFMAs they left the pipe idle while the
    // vector-memory unit took them; the group barriers below spread them, one behind each of the first eight MFMAs.
#define H3_SPREAD(n_vmem, n_lds)                                                                          \
    do {                                                                                                  \
        for (int _i = 0; _i < (n_vmem); ++_i) {                                                           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* one MFMA */                             \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); /* one vector-memory read */               \
        }                                                                                                 \
        for (int _i = 0; _i < (n_lds); ++_i) {                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* one LDS read */                         \
        }                                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * RT - (n_vmem) - (n_lds), 0);                      \
        H3_FENCE();                                                                                       \
    } while (0)
    h3_zero<RT>(ac);
    H3FragB b3;
    H3FragA a0, a1;
    load_a(0, a0);
    H3_FENCE();
#pragma unroll 1
    for (int s = 0; s < n_steps - 4; s += 4) {  // n_steps is a multiple of 4 (512 / 16 = 32)
        h3_load_b(w, n_steps, s + 3, b3);
        load_a(s + 1, a1);
        h3_mma<RT, TERMS>(ac, a0, ring.b0);
        H3_SPREAD(4, 2 * RT);
        h3_load_b(w, n_steps, s + 4, ring.b0);
        load_a(s + 2, a0);
        h3_mma<RT, TERMS>(ac, a1, ring.b1);
        H3_SPREAD(4, 2 * RT);
        h3_load_b(w, n_steps, s + 5, ring.b1);
        load_a(s + 3, a1);
        h3_mma<RT, TERMS>(ac, a0, ring.b2);
        H3_SPREAD(4, 2 * RT);
        h3_load_b(w, n_steps, s + 6, ring.b2);
        load_a(s + 4, a0);
        h3_mma<RT, TERMS>(ac, a1, b3);
        H3_SPREAD(4, 2 * RT);
    }
    {   // the last four k-steps: the ring sets that fall free take the head of the next pass of the stream
        constexpr int s = n_steps - 4;
        h3_load_b(w, n_steps, s + 3, b3);
        load_a(s + 1, a1);
        h3_mma<RT, TERMS>(ac, a0, ring.b0);
        H3_SPREAD(4, 2 * RT);
        if (HAS_NEXT) h3_load_b(next, n_steps, 0, ring.b0);
        load_a(s + 2, a0);
        h3_mma<RT, TERMS>(ac, a1, ring.b1);
        H3_SPREAD(HAS_NEXT ? 4 : 0, 2 * RT);
        if (HAS_NEXT) h3_load_b(next, n_steps, 1, ring.b1);
        load_a(s + 3, a1);
        h3_mma<RT, TERMS>(ac, a0, ring.b2);
        H3_SPREAD(HAS_NEXT ? 4 : 0, 2 * RT);
        if (HAS_NEXT) h3_load_b(next, n_steps, 2, ring.b2);
        h3_mma<RT, TERMS>(ac, a1, b3);
        H3_SPREAD(HAS_NEXT ? 4 : 0, 0);
    }
#undef H3_SPREAD
#undef H3_FENCE
}

// a 512-wide hidden layer: this wave's 128 columns of all 64 samples.  `ring`: the first three k-steps of the layer,
// requested by the caller (h3_prime_layer) once the previous layer's epilogue has let go of its registers -- they fly
// while the workgroup meets at the barrier in front of this layer.
template <int NW> __device__ __forceinline__ void h3_prime_layer(const unsigned short *layer, int wid, int lane, H3Ring &ring) {
    h3_prime(h3_pass<NW>(layer, H3_STEPS, wid, 0, lane), H3_STEPS, ring);
}
template <int NW, int RT, int TERMS = 3>
__device__ __forceinline__ void gemm_layer_h3(f32x16 (&acc)[8 / NW][RT][2], const _Float16 *a_hi, const _Float16 *a_lo,
                                              const unsigned short *layer, int wid, int lane, H3Ring &ring) {
    const H3Pass w0 = h3_pass<NW>(layer, H3_STEPS, wid, 0, lane);
    if (NW == 4) {
        const H3Pass w1 = h3_pass<NW>(layer, H3_STEPS, wid, 1, lane);
        gemm_pass_h3<true, RT, TERMS>(acc[0], a_hi, a_lo, w0, w1, ring, lane);
        gemm_pass_h3<false, RT, TERMS>(acc[8 / NW - 1], a_hi, a_lo, w1, w1, ring, lane);
    } else {
        gemm_pass_h3<false, RT, TERMS>(acc[0], a_hi, a_lo, w0, w0, ring, lane);
    }
}

// Linear(5 -> 512) of the step's inputs: one k-step (z rows padded to 16), both passes
template <int NW, int RT>
__device__ __forceinline__ void gemm_input_h3(f32x16 (&acc)[8 / NW][RT][2], const _Float16 *z_hi, const _Float16 *z_lo,
                                              const unsigned short *layer, int wid, int lane) {
    const int aoff = (lane & 31) * H3_ZPITCH + 8 * (lane >> 5);
    H3FragA a;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        a.h[rt] = *reinterpret_cast<const half8 *>(z_hi + rt * 32 * H3_ZPITCH + aoff);
        a.l[rt] = *reinterpret_cast<const half8 *>(z_lo + rt * 32 * H3_ZPITCH + aoff);
    }
#pragma unroll
    for (int pass = 0; pass < 8 / NW; ++pass) {
        const H3Pass w = h3_pass<NW>(layer, 1, wid, pass, lane);
        H3FragB b;
        h3_load_b(w, 1, 0, b);
        h3_zero<RT>(acc[pass]);
        h3_mma<RT>(acc[pass], a, b);
    }
}

// bias (+ tanh), split, then this wave's [64 x 128] slice of both planes back to LDS.  Result layout (see gemm_pass_h3):
// register r of tile (rt, ct) in lane l = sample 32 rt + (l & 31), feature 128 wid + 32 ct + (r & 3) + 8 (r >> 2) + 4 (l >> 5).
// The launch spends a fifth of its time here (one wave per SIMD: nothing else issues meanwhile), so two elements at a
// time: packed f32 arithmetic for the bias, the tanh's linear parts and the residual, and `v_cvt_pkrtz_f16_f32` for
// both conversions (the high part may round towards zero: the low part takes what is left, exactly) -- a converted
// pair IS two consecutive features, two pairs one 8-byte store.
using f32x2 = __attribute__((ext_vector_type(2))) float;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using half4v = __attribute__((ext_vector_type(4))) _Float16;
// Written in stages over the 16 registers of a tile (eight independent pairs per stage: with one wave per SIMD a chain
// of dependent instructions -- exp, add, rcp, fma, convert, subtract, convert -- runs at its latency, 90 cycles per
// element as the compiler first scheduled it, one pair after the other), and with four base addresses (plane x row tile)
// plus immediate offsets for the stores.
// LAST (the third hidden layer): its activations feed only the 512 -> 3 output layer, so they are not split or stored at
// all -- every lane multiplies its 64 features of a sample by the output weights as they leave the tanh (f32, `yo[rt][j]`),
// and the caller adds up the two lane halves and the four waves.  (Reading them back from LDS on the vector unit took 6 %
// of the launch, splitting and storing them another 2 %.)
// SCALE (the f16 range, see H3Scale): 0 none; 1 the input layer -- its accumulators are those of z / s0 and its output is
// stored as h0 / s1: v = (acc s0 + b) / s1; 2 the first hidden layer -- its accumulators are those of h0 / s1: v = acc s1 + b.
// `scale`: the per-sample {s0, 1 / s1, s1, 0} in LDS.  All three are 1 unless an input exceeds 2^15, and then exact powers
// of two: the common case is bit for bit the unscaled arithmetic.
template <int NW, int RT, bool TANH, bool LAST = false, int SCALE = 0>
__device__ __forceinline__ void store_layer_h3(_Float16 *a_hi, _Float16 *a_lo, const f32x16 (&acc)[8 / NW][RT][2], const float *bias,
                                               int wid, int lane, const float *w_out = nullptr, float (*yo)[3] = nullptr,
                                               const float *scale = nullptr) {
    F4 sc[2] = {{{1.f, 1.f, 1.f, 0.f}}, {{1.f, 1.f, 1.f, 0.f}}};
    if (SCALE != 0) {
        sc[0] = *reinterpret_cast<const F4 *>(scale + 4 * (lane & 31));
        if (RT == 2) sc[1] = *reinterpret_cast<const F4 *>(scale + 4 * (32 + (lane & 31)));
    }
    constexpr int COLS = MLP_H / NW;  // this wave's output features: 16 / NW column tiles of 32
    const int lane_off = (lane & 31) * H3_PITCH + wid * COLS + 4 * (lane >> 5);
    _Float16 *const base[2][2] = {{a_hi + lane_off, a_hi + 32 * H3_PITCH + lane_off},
                                  {a_lo + lane_off, a_lo + 32 * H3_PITCH + lane_off}};
    const float *bl = bias + wid * COLS + 4 * (lane >> 5);
#pragma unroll
    for (int ct = 0; ct < 16 / NW; ++ct) {
        f32x2 bn[8];  // the biases of this lane's 16 features of the tile: four aligned 16-byte loads
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const F4 b4 = *reinterpret_cast<const F4 *>(bl + ct * 32 + 8 * q);
            bn[2 * q] = f32x2{b4.v[0], b4.v[1]};
            bn[2 * q + 1] = f32x2{b4.v[2], b4.v[3]};
        }
        F4 wo[3][4];  // LAST: the output layer's weights of the same 16 features, requested before the tanh arithmetic
        if (LAST) {
            const float *wl = w_out + wid * COLS + 4 * (lane >> 5) + ct * 32;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) wo[j][q] = *reinterpret_cast<const F4 *>(wl + j * MLP_H + 8 * q);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x2 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2 a = f32x2{acc[ct >> 1][rt][ct & 1][2 * i], acc[ct >> 1][rt][ct & 1][2 * i + 1]};
                if (SCALE == 1) v[i] = (a * sc[rt].v[0] + bn[i]) * sc[rt].v[1];
                else if (SCALE == 2) v[i] = a * sc[rt].v[2] + bn[i];
                else v[i] = a + bn[i];
            }
            if (TANH) {  // 1 - 2 / (exp(2x) + 1), exp(2x) = 2^(x * 2 log2(e))
                f32x2 e[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x2 a = v[i] * 2.8853900817779268f;
                    e[i] = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} + 1.0f;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = f32x2{__builtin_amdgcn_rcpf(e[i].x), __builtin_amdgcn_rcpf(e[i].y)};
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = 1.0f - 2.0f * e[i];
            }
            if (LAST) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    f32x2 a = f32x2{0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        a = a + v[2 * q] * f32x2{wo[j][q].v[0], wo[j][q].v[1]} + v[2 * q + 1] * f32x2{wo[j][q].v[2], wo[j][q].v[3]};
                    yo[rt][j] += a.x + a.y;
                }
                continue;
            }
            half2v hi[8], lo[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) hi[i] = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(v[i].x, v[i].y));
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = v[i] - f32x2{(float)hi[i].x, (float)hi[i].y};
#pragma unroll
            for (int i = 0; i < 8; ++i) lo[i] = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(v[i].x, v[i].y));
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // registers 4q .. 4q + 3: features 8q .. 8q + 3 (+ lane, tile and wave parts)
                const int off = ct * 32 + 8 * q;
                *reinterpret_cast<half4v *>(base[0][rt] + off) = half4v{hi[2 * q].x, hi[2 * q].y, hi[2 * q + 1].x, hi[2 * q + 1].y};
                *reinterpret_cast<half4v *>(base[1][rt] + off) = half4v{lo[2 * q].x, lo[2 * q].y, lo[2 * q + 1].x, lo[2 * q + 1].y};
            }
        }
        // (one column tile at a time: unfenced, the scheduler lifts the bias / weight loads of all four tiles to the top of
        // the unrolled block and the kernel spills)
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifdef MPPI_STAMPS
__device__ unsigned long long g_mlp_phase[16];
#define PH(i)                                             \
    do {                                                  \
        const unsigned long long _n = clock64();          \
        ph[i] += _n - ph_t;                               \
        ph_t = _n;                                        \
    } while (0)
#else
#define PH(i) \
    do {      \
    } while (0)
#endif
template <bool VIZ, int NW, int RT, int TERMS = 3>
__global__ __launch_bounds__(64 * NW, RT == 1 ? 2 : 1) void k_rollout_mlp_h3(const KParams<float> P, const MlpParams Q,
                                                               float *__restrict__ partials, const MlpViz V) {
#ifdef MPPI_STAMPS
    unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int M = 32 * RT;  // samples of this workgroup's tile
    _Float16 *a_hi = reinterpret_cast<_Float16 *>(smem);   // [M][520]
    _Float16 *a_lo = a_hi + M * H3_PITCH;
    _Float16 *z_hi = a_lo + M * H3_PITCH;                  // [M][24] layer-0 input rows {x, y, yaw, v, w, 0 ...}: one k-step
    _Float16 *z_lo = z_hi + M * H3_ZPITCH;
    float *ypart = reinterpret_cast<float *>(z_lo + M * H3_ZPITCH);  // [NW][M][4]
    float *zscale = ypart + NW * M * 4;                             // [M][4] per-sample {s0, 1 / s1, s1, 0} (H3Scale)
    float *ref_lds = zscale + M * 4;                                // [n_ref][4] when the path fits
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int k0 = ((int)blockIdx.x + (VIZ ? V.block0 : 0)) * M, k = k0 + lane;
    const KParams<float> PL = mlp_stage_path(P, ref_lds, RT == 1 ? H3_REF_LDS_32 : MLP_REF_LDS_MAX);
    const DevState sv = load_state(P, P.st);
    const ObsLanes<float> obs = load_obstacles(P, lane);
    if (!VIZ && k0 + M <= sv.k_start) return;
    const bool in_tile = lane < M;  // (a 32-sample tile: the upper half of wave 0 carries no sample)
    const bool valid = in_tile && k < P.K, live = valid && k >= sv.k_start;
    // VIZ: the workgroup behind the samples' carries the nominal sequence in its lane 0
    const bool eval = VIZ && V.ex != nullptr, eval_row = eval && k < V.en;
    const bool opt_row = VIZ && !eval && k0 >= P.K && lane == 0 && V.opt != nullptr;
    const bool smp_row = VIZ && !eval && valid && V.smp != nullptr;
    const int c = sv.c;
    const unsigned iter = (unsigned)sv.iter;
    MlpLane L{(float)sv.x0[0], (float)sv.x0[1], (float)sv.x0[2], 0.f, c};
    if (eval_row) { L.x = V.ex[3 * k]; L.y = V.ex[3 * k + 1]; L.yaw = V.ex[3 * k + 2]; }
    const int n_steps = eval ? 1 : P.T;
    const bool exploit = (k + P.k_offset) < P.n_exploit;
    f32x16 acc[8 / NW][RT][2];
    if (wid == 0 && in_tile) {  // the padding of the layer-0 rows stays zero
        for (int q = 0; q < H3_ZPITCH; ++q) { z_hi[lane * H3_ZPITCH + q] = (_Float16)0.f; z_lo[lane * H3_ZPITCH + q] = (_Float16)0.f; }
    }
    for (int t = 0; t < n_steps; ++t) {
        float u0 = 0, u1 = 0, v0 = 0, v1 = 0;
        if (wid == 0) {
            if (eval) {
                if (eval_row) { v0 = V.ev[2 * k]; v1 = V.ev[2 * k + 1]; }
            } else if (VIZ) mlp_controls_viz(P, V, k, t, smp_row, opt_row, exploit, v0, v1);
            else mlp_controls(P, iter, k, t, valid, exploit, u0, u1, v0, v1);
            const float z[5] = {L.x, L.y, L.yaw, v0, v1};
            const H3Scale hs = h3_scale(z, Q.in_gain, Q.in_bias);
            if (in_tile) {
                *reinterpret_cast<F4 *>(zscale + 4 * lane) = F4{{hs.s0, hs.inv_s1, hs.s1, 0.f}};
#pragma unroll
                for (int q = 0; q < 5; ++q) split_h3(z[q] * hs.inv_s0, z_hi[lane * H3_ZPITCH + q], z_lo[lane * H3_ZPITCH + q]);
            }
        }
        __syncthreads();
        PH(0);
        gemm_input_h3<NW, RT>(acc, z_hi, z_lo, Q.h3_w_in, wid, lane);
        PH(1);
        store_layer_h3<NW, RT, false, false, 1>(a_hi, a_lo, acc, Q.b_in, wid, lane, nullptr, nullptr, zscale);
        H3Ring ring;
        h3_prime_layer<NW>(Q.h3_w_h[0], wid, lane, ring);
        PH(2);
        __syncthreads();
        PH(3);
        const int l_last = Q.n_hidden - 1;  // (2 or 3 hidden layers: mppi_set_mlp)
        for (int l = 0; l < l_last; ++l) {
            gemm_layer_h3<NW, RT, TERMS>(acc, a_hi, a_lo, Q.h3_w_h[l], wid, lane, ring);
            PH(4);
            __syncthreads();
            PH(5);
            if (l == 0) store_layer_h3<NW, RT, true, false, 2>(a_hi, a_lo, acc, Q.b_h[l], wid, lane, nullptr, nullptr, zscale);
            else store_layer_h3<NW, RT, true>(a_hi, a_lo, acc, Q.b_h[l], wid, lane);
            h3_prime_layer<NW>(Q.h3_w_h[l + 1], wid, lane, ring);
            PH(6);
            __syncthreads();
            PH(7);
        }
        {   // the last hidden layer and out_layer (Linear(512 -> 3), :35) in its epilogue: this wave's share of the 512 inputs
            gemm_layer_h3<NW, RT, TERMS>(acc, a_hi, a_lo, Q.h3_w_h[l_last], wid, lane, ring);
            PH(4);
            float yo[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            store_layer_h3<NW, RT, true, true>(a_hi, a_lo, acc, Q.b_h[l_last], wid, lane, Q.w_out, yo);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {  // the two lane halves hold the two halves of a sample's features
                F4 o;
#pragma unroll
                for (int j = 0; j < 3; ++j) o.v[j] = yo[rt][j] + __shfl_xor(yo[rt][j], 32);
                o.v[3] = 0.f;
                if (lane < 32) *reinterpret_cast<F4 *>(ypart + (wid * M + rt * 32 + lane) * 4) = o;
            }
        }
        PH(8);
        __syncthreads();
        if (wid == 0) {
            float r0 = Q.b_out[0], r1 = Q.b_out[1], r2 = Q.b_out[2];
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const F4 o = *reinterpret_cast<const F4 *>(ypart + (w * M + (in_tile ? lane : 0)) * 4);
                r0 += o.v[0];
                r1 += o.v[1];
                r2 += o.v[2];
            }
            if (VIZ) {
                mlp_euler(P, r0, r1, r2, v0, v1, L);
                float *dst = eval_row ? V.eout + (size_t)k * 3
                             : opt_row ? V.opt + (size_t)t * 3 : smp_row ? V.smp + ((size_t)k * P.T + t) * 3 : nullptr;
                if (dst) { dst[0] = L.x; dst[1] = L.y; dst[2] = L.yaw; }
            } else {
                mlp_advance(PL, obs, c, t, r0, r1, r2, u0, u1, v0, v1, L);
            }
        }
        PH(9);
    }
    if (!VIZ && wid == 0) mlp_record(P, partials, iter, k, c, valid, live, L, lane);
#ifdef MPPI_STAMPS
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && wid == 1)
        for (int i = 0; i < 10; ++i) g_mlp_phase[i] = ph[i];
#endif
}
#undef PH
#ifdef MPPI_STAMPS
extern "C" int mppi_debug_mlp_phases(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_phase), sizeof(unsigned long long) * 10);
}
#endif

// The forms of the f16-split rollout kernel (MPPI_MLP_FORM, for A/B runs): "4x64" one wave per SIMD, "8x64" two waves per
// SIMD, both on one 64-sample tile per workgroup.  (Tried and dropped: 32-sample tiles with two workgroups per CU, so that one
// workgroup's epilogues run under the other's GEMMs -- every weight byte then feeds half as many MFMAs, the weight stream
// from L2 doubles to ~85 B per clock and CU, and the launch went from 6.3 to 8.7 ms.)
enum { H3_FORM_4x64 = 0, H3_FORM_8x64 = 1 };
static int h3_form() {
    static const int f = [] {
        const char *e = getenv("MPPI_MLP_FORM");
        if (e && !strcmp(e, "4x64")) return (int)H3_FORM_4x64;
        if (e && !strcmp(e, "8x64")) return (int)H3_FORM_8x64;
        return (int)H3_FORM_DEFAULT;
    }();
    return f;
}
// MPPI_MLP_TERMS=2: the split product without the a_lo b_hi term (see h3_mma); read at every launch, so a process can compare
static int h3_terms() {
    const char *e = getenv("MPPI_MLP_TERMS");
    return e && atoi(e) == 2 ? 2 : 3;
}
// samples per workgroup (= per softmin record) of the rollout kernel that serves Q
int mlp_tile(const MlpParams &) { return MLP_M; }
int mlp_blocks(int K, int tile) { return (K + tile - 1) / tile; }

static size_t h3_shmem(int nw, int rt) {
    const int m = 32 * rt;
    return sizeof(_Float16) * 2 * (m * H3_PITCH + m * H3_ZPITCH) + sizeof(float) * (nw + 1) * m * 4 +
           sizeof(float) * 4 * (rt == 1 ? H3_REF_LDS_32 : MLP_REF_LDS_MAX);
}

static void launch_mlp_any(const KParams<float> &P, const MlpParams &Q, void *partials, const MlpViz *viz, hipStream_t s) {
    const size_t ref_lds = sizeof(float) * 4 * MLP_REF_LDS_MAX;  // the path (mlp_stage_path)
    const size_t shmem_f32 = sizeof(float) * (MLP_M * MLP_PITCH + MLP_M * 8 + MLP_WAVES * MLP_M * 4) + ref_lds;
    // (the attribute belongs to the device's copy of the code object: one process may drive several GPUs)
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rollout_mlp), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)shmem_f32);
#define H3_ATTR(VIZ_, NW_, RT_)                                                                                          \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rollout_mlp_h3<VIZ_, NW_, RT_>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)h3_shmem(NW_, RT_))
        H3_ATTR(false, 4, 2); H3_ATTR(true, 4, 2); H3_ATTR(false, 8, 2); H3_ATTR(true, 8, 2);
#undef H3_ATTR
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rollout_mlp_h3<false, 8, 2, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)h3_shmem(8, 2));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const MlpViz none{nullptr, nullptr, nullptr, nullptr, 0u, 0, nullptr, nullptr, nullptr, 0};
    const int form = h3_form();
    if (viz) {  // the samples' workgroups (when their trajectories are wanted) and one for the nominal sequence (likewise);
                // always 64-sample tiles
        const dim3 grid(viz->ex ? mlp_blocks(viz->en, MLP_M) : (viz->smp ? mlp_blocks(P.K, MLP_M) : 0) + (viz->opt ? 1 : 0));
        if (form == H3_FORM_8x64) hipLaunchKernelGGL((k_rollout_mlp_h3<true, 8, 2>), grid, dim3(512), h3_shmem(8, 2), s, P, Q, (float *)partials, *viz);
        else hipLaunchKernelGGL((k_rollout_mlp_h3<true, 4, 2>), grid, dim3(256), h3_shmem(4, 2), s, P, Q, (float *)partials, *viz);
    } else if (Q.use_h3) {
        const dim3 grid(mlp_blocks(P.K, mlp_tile(Q)));
        if (form == H3_FORM_8x64 && h3_terms() == 2)
            hipLaunchKernelGGL((k_rollout_mlp_h3<false, 8, 2, 2>), grid, dim3(512), h3_shmem(8, 2), s, P, Q, (float *)partials, none);
        else if (form == H3_FORM_8x64) hipLaunchKernelGGL((k_rollout_mlp_h3<false, 8, 2>), grid, dim3(512), h3_shmem(8, 2), s, P, Q, (float *)partials, none);
        else hipLaunchKernelGGL((k_rollout_mlp_h3<false, 4, 2>), grid, dim3(256), h3_shmem(4, 2), s, P, Q, (float *)partials, none);
    } else {
        hipLaunchKernelGGL(k_rollout_mlp, dim3(mlp_blocks(P.K, MLP_M)), dim3(64 * MLP_WAVES), shmem_f32, s, P, Q, (float *)partials);
    }
}

const char *mlp_kernel_name(const MlpParams &Q) {
    if (!Q.use_h3) return "k_rollout_mlp(";
    const int form = h3_form();
    if (form == H3_FORM_8x64 && h3_terms() == 2) return "k_rollout_mlp_h3<false, 8, 2, 2>";
    return form == H3_FORM_8x64 ? "k_rollout_mlp_h3<false, 8, 2, 3>" : "k_rollout_mlp_h3<false, 4, 2, 3>";
}
void launch_rollout_mlp(const KParams<float> &P, const MlpParams &Q, void *partials, hipStream_t s) {
    launch_mlp_any(P, Q, partials, nullptr, s);
}

// (the f16-split kernel serves the visualisation whichever rollout kernel MPPI_MLP_F32 selects: both weight sets are packed)
void launch_viz_mlp(const KParams<float> &P, const MlpParams &Q, const float *u_before, const float *u_upd, long long iter,
                    float *opt, float *smp, hipStream_t s) {
    if (!opt && !smp) return;
    const MlpViz v{u_before, u_upd, opt, smp, (unsigned)iter, smp ? 0 : mlp_blocks(P.K, MLP_M), nullptr, nullptr, nullptr, 0};
    launch_mlp_any(P, Q, nullptr, &v, s);
}

void launch_eval_mlp(const KParams<float> &P, const MlpParams &Q, const float *x, const float *v, int n, float *out, hipStream_t s) {
    const MlpViz e{nullptr, nullptr, nullptr, nullptr, 0u, 0, x, v, out, n};
    launch_mlp_any(P, Q, nullptr, &e, s);
}

// Host-side packing of a torch Linear weight [n_out = 512][n_in] into fragment order:
// packed[ct (16)][g (n_groups)][lane (64)][s (4)] = W[32 ct + (lane & 31)][8 g + 4 (lane >> 5) + s] (0 beyond n_in)
void pack_linear(const float *w, int n_in, float *packed) {
    const int n_groups = (n_in + 7) / 8;
    for (int ct = 0; ct < MLP_H / 32; ++ct)
        for (int g = 0; g < n_groups; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 4; ++s) {
                    const int n = 32 * ct + (lane & 31), kk = 8 * g + 4 * (lane >> 5) + s;
                    packed[(((size_t)ct * n_groups + g) * 64 + lane) * 4 + s] = kk < n_in ? w[(size_t)n * n_in + kk] : 0.f;
                }
}

// IEEE binary16 of a float, round to nearest even (host; the kernels' v_cvt_f16_f32 does the same)
static unsigned short f32_to_f16_bits(float f) {
    unsigned int x;
    memcpy(&x, &f, 4);
    const unsigned int sign = (x >> 16) & 0x8000u;
    const int exp = (int)((x >> 23) & 0xffu) - 127 + 15;
    unsigned int man = x & 0x7fffffu;
    if (((x >> 23) & 0xffu) == 0xffu) return (unsigned short)(sign | 0x7c00u | (man ? 0x200u : 0u));
    if (exp >= 31) return (unsigned short)(sign | 0x7c00u);
    if (exp <= 0) {  // subnormal or zero
        if (exp < -10) return (unsigned short)sign;
        man |= 0x800000u;
        const int shift = 14 - exp;
        unsigned int h = man >> shift;
        const unsigned int rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) ++h;
        return (unsigned short)(sign | h);
    }
    unsigned int h = ((unsigned int)exp << 10) | (man >> 13);
    const unsigned int rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;  // (a carry into the exponent is the right result)
    return (unsigned short)(sign | h);
}
static float f16_bits_to_f32(unsigned short h) {
    const unsigned int sign = (unsigned int)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    unsigned int x;
    if (exp == 0) {
        if (man == 0) x = sign;
        else {
            int e = -1;
            unsigned int m = man;
            do { ++e; m <<= 1; } while (!(m & 0x400u));
            x = sign | ((unsigned int)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (exp == 31) x = sign | 0x7f800000u | (man << 13);
    else x = sign | ((exp - 15 + 127) << 23) | (man << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// Host-side packing for k_rollout_mlp_h3: W [512][n_in] -> two f16 planes (hi, then lo) in fragment order
// plane[ct (16)][s (n_steps)][lane (64)][j (8)] = W[32 ct + (lane & 31)][16 s + 8 (lane >> 5) + j]  (0 beyond n_in)
void pack_linear_h3(const float *w, int n_in, unsigned short *packed) {
    const int n_steps = (n_in + 15) / 16;
    const size_t plane = (size_t)(MLP_H / 32) * n_steps * 64 * 8;
    for (int ct = 0; ct < MLP_H / 32; ++ct)
        for (int s = 0; s < n_steps; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int n = 32 * ct + (lane & 31), kk = 16 * s + 8 * (lane >> 5) + j;
                    const float v = kk < n_in ? w[(size_t)n * n_in + kk] : 0.f;
                    const unsigned short hi = f32_to_f16_bits(v);
                    const unsigned short lo = f32_to_f16_bits(v - f16_bits_to_f32(hi));
                    const size_t o = (((size_t)ct * n_steps + s) * 64 + lane) * 8 + j;
                    packed[o] = hi;
                    packed[plane + o] = lo;
                }
}

}  // namespace mppi
