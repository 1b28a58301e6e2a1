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
#include "mppi_device.h"

namespace mppi {

constexpr int MLP_M = 64, MLP_H = 512, MLP_PITCH = 516, MLP_WAVES = 4, MLP_GROUPS = MLP_H / 8;
using f32x16 = __attribute__((ext_vector_type(16))) float;
struct alignas(16) F4 { float v[4]; };

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

__global__ __launch_bounds__(64 * MLP_WAVES, 1) void k_rollout_mlp(const KParams<float> P, const MlpParams Q,
                                                                    float *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *act = smem;                       // [64][516]
    float *zbuf = act + MLP_M * MLP_PITCH;   // [64][8]   layer-0 input rows {x, y, yaw, v, w, 0, 0, 0}
    float *ypart = zbuf + MLP_M * 8;         // [4][64][4] partial outputs of the last Linear per wave
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int k0 = blockIdx.x * MLP_M, k = k0 + lane;
    const DevState sv = load_state(P, P.st);
    const ObsLanes<float> obs = load_obstacles(P, lane);
    if (k0 + MLP_M <= sv.k_start) return;  // every sample of the tile is final: its record stands
    const bool valid = k < P.K, live = valid && k >= sv.k_start;
    const int c = sv.c;
    const unsigned iter = (unsigned)sv.iter;
    const float *__restrict__ ref = P.ref;

    // sample state lives in the lanes of wave 0 (one lane = one sample of the tile)
    float x = (float)sv.x0[0], y = (float)sv.x0[1], yaw = (float)sv.x0[2];
    const bool exploit = (k + P.k_offset) < P.n_exploit;
    int p = c;
    float S = 0.f;
    f32x16 acc[2][4];

    for (int t = 0; t < P.T; ++t) {
        float u0 = 0, u1 = 0, v0 = 0, v1 = 0;
        if (wid == 0) {
            float e0 = 0.f, e1 = 0.f;
            if (valid) {
                if (P.use_philox) px::sample(P.seed_lo, P.seed_hi, iter, (unsigned)(k + P.k_offset), t, P.chol, e0, e1, (unsigned)P.noise_stream);
                else {
                    const float2 e = *reinterpret_cast<const float2 *>(P.eps + ((size_t)k * P.T + t) * 2);
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
            F4 z0 = {{x, y, yaw, v0}}, z1 = {{v1, 0.f, 0.f, 0.f}};
            *reinterpret_cast<F4 *>(zbuf + lane * 8) = z0;
            *reinterpret_cast<F4 *>(zbuf + lane * 8 + 4) = z1;
        }
        __syncthreads();
        // input_layer: Linear(5 -> 512), no activation (train/train_diff_mlp.py:32)
        gemm_layer(acc, zbuf, 8, Q.w_in, 1, wid, lane);
        store_layer<false>(act, acc, Q.b_in, wid, lane);
        __syncthreads();
        // hidden_layer[i]: tanh(Linear(512 -> 512)) (:33-34)
        for (int l = 0; l < 3; ++l) {
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
            float sn, cs;
            mf::sincos_(yaw, sn, cs);
            x = x + P.dt * (v0 * cs + r0);  // Euler step of f + MLP (bullet_differential_drive_dnn.py:79-92)
            y = y + P.dt * (v0 * sn + r1);
            yaw = yaw + P.dt * (v1 + r2);
            // waypoint index of this call: per-lane, so the sequential index threads through the sample's own
            // calls in order by construction (mppi_differential_drive.py:228)
            int idx;
            const int from = P.sequential ? p : c;
            {
                float best = dist2(ref, from, x, y);
                int bj = 0;
                for (int j = 1; j < P.window; ++j) {
                    const bool ok = from + j < P.n_ref;
                    const float d = ok ? dist2(ref, min(from + j, P.n_ref - 1), x, y) : INFINITY;
                    if (d < best) { best = d; bj = j; }
                }
                idx = from + bj;
            }
            if (P.sequential) p = idx;
            if (P.accumulate || t == P.T - 1) {
                const bool hit = collided<false>(P, x, y, yaw, obs);
                float st_c = tracking_cost<float, MODEL_DIFF>(P, P.ws, P.wrap_stage, idx, x, y, yaw, 0.f);
                if (hit) st_c += P.penalty;
                const float ctrl = (u0 * P.sinv[0] + u1 * P.sinv[2]) * v0 + (u0 * P.sinv[1] + u1 * P.sinv[3]) * v1;
                const float stage = st_c + P.gamma * ctrl;
                S = P.accumulate ? S + stage : stage;
                if (t == P.T - 1) {
                    int idx_term = idx;
                    if (P.sequential) {  // the terminal call moves the index once more (:244)
                        float best = dist2(ref, p, x, y);
                        int bj = 0;
                        for (int j = 1; j < P.window; ++j) {
                            const bool ok = p + j < P.n_ref;
                            const float d = ok ? dist2(ref, min(p + j, P.n_ref - 1), x, y) : INFINITY;
                            if (d < best) { best = d; bj = j; }
                        }
                        p = p + bj;
                        idx_term = p;
                    }
                    float term = tracking_cost<float, MODEL_DIFF>(P, P.wt, P.wrap_term, idx_term, x, y, yaw, 0.f);
                    if (hit) term += P.penalty;
                    S += term;
                }
            }
        }
        // zbuf / ypart are rewritten only after the next barrier sequence: wave 0 writes zbuf at the top of
        // the next step while the others wait at that step's first barrier
    }

    // ---- this tile's softmin record {rho, eta, eta2, pad, W[T][2]} -------------------------------------
    if (wid == 0) {
        if (live) {
            P.S[k] = S;
            P.pout[k] = p;
            if (P.sequential && p != c) atomicMin(&P.st->first_k, k);
        } else if (valid) {
            S = P.S[k];  // final from an earlier speculation round
        }
        const float Sm = valid ? S : INFINITY;
        const float rho = wv::reduce<wv::OpMin>(Sm);
        const float e = valid ? mf::exp_(-P.beta * (S - rho)) : 0.f;
        const float eta = wv::reduce<wv::OpAdd>(e), eta2 = wv::reduce<wv::OpAdd>(e * e);
        float *out = partials + (size_t)blockIdx.x * record_len(P.T, 4);
        if (lane == 0) {
            out[0] = rho; out[1] = eta; out[2] = eta2;
            float *hd = P.heads + 4 * (size_t)blockIdx.x;  // the compact copy the merge kernels read
            hd[0] = rho; hd[1] = eta; hd[2] = eta2; hd[3] = 0.f;
        }
        for (int t = 0; t < P.T; ++t) {  // second pass over this tile's noise rows (regenerated / re-read)
            float e0 = 0.f, e1 = 0.f;
            if (valid) {
                if (P.use_philox) px::sample(P.seed_lo, P.seed_hi, iter, (unsigned)(k + P.k_offset), t, P.chol, e0, e1, (unsigned)P.noise_stream);
                else {
                    const float2 ee = *reinterpret_cast<const float2 *>(P.eps + ((size_t)k * P.T + t) * 2);
                    e0 = ee.x;
                    e1 = ee.y;
                }
            }
            const float w0 = wv::reduce<wv::OpAdd>(e * e0), w1 = wv::reduce<wv::OpAdd>(e * e1);
            if (lane == 0) { out[4 + 2 * t] = w0; out[4 + 2 * t + 1] = w1; }
        }
    }
}

int mlp_blocks(int K) { return (K + MLP_M - 1) / MLP_M; }

void launch_rollout_mlp(const KParams<float> &P, const MlpParams &Q, void *partials, hipStream_t s) {
    const size_t shmem = sizeof(float) * (MLP_M * MLP_PITCH + MLP_M * 8 + MLP_WAVES * MLP_M * 4);
    // (the attribute belongs to the device's copy of the code object: one process may drive several GPUs)
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rollout_mlp), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)shmem);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(k_rollout_mlp, dim3(mlp_blocks(P.K)), dim3(64 * MLP_WAVES), shmem, s, P, Q, (float *)partials);
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

}  // namespace mppi
