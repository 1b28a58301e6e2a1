// Scalar math used by the rollout: one implementation per arithmetic type.
#pragma once
#include <hip/hip_runtime.h>

namespace mf {

// sin/cos of an fp32 angle: 3-term Cody-Waite reduction by pi/2 + degree-9/8 minimax
// polynomials on [-pi/4, pi/4] (about 20 VALU ops, <= 2 ulp for |x| < 2^15).  Larger
// arguments (a yaw that wound up thousands of turns; the reference never wraps it,
// mppi_differential_drive.py:196) fall back to the library routine.
__device__ __forceinline__ void sincos_poly(float r, int q, float &s, float &c) {
    const float z = r * r;
    float pc = 2.44677067e-5f;
    pc = fmaf(pc, z, -1.38877297e-3f);
    pc = fmaf(pc, z, 4.16666567e-2f);
    pc = fmaf(pc, z, -5.00000000e-1f);
    pc = fmaf(pc, z, 1.0f);
    float ps = 2.86567956e-6f;
    ps = fmaf(ps, z, -1.98559923e-4f);
    ps = fmaf(ps, z, 8.33338592e-3f);
    ps = fmaf(ps, z, -1.66666672e-1f);
    ps = fmaf(ps, r * z, r);
    const float ss = (q & 1) ? pc : ps, cc = (q & 1) ? ps : pc;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ void sincos_(float x, float &s, float &c) {
    if (__builtin_expect(fabsf(x) > 32768.0f, 0)) {
        sincosf(x, &s, &c);
        return;
    }
    const float kf = rintf(x * 6.36619747e-1f);
    float r = fmaf(kf, -1.57079601e+00f, x);
    r = fmaf(kf, -3.13916473e-07f, r);
    r = fmaf(kf, -5.39030253e-15f, r);
    sincos_poly(r, (int)kf, s, c);
}
__device__ __forceinline__ void sincos_(double x, double &s, double &c) { sincos(x, &s, &c); }

// sin/cos(2*pi*u) for u in [0,1]: v_sin_f32 / v_cos_f32 take their argument in revolutions -- two (quarter-rate)
// instructions instead of a quarter-turn reduction and two polynomials (~20); against the f64 restatement of the
// sampler the draw is as close as with the polynomials (max 4.8e-7 on eps either way, tools/sampler_check.py)
__device__ __forceinline__ void sincos_turns(float u, float &s, float &c) {
    s = __builtin_amdgcn_sinf(u);
    c = __builtin_amdgcn_cosf(u);
}

// tan of a steering angle (mppi_race_car.py:192): the bicycle's steer is clamped to +-max_steer_abs (0.523 rad in the
// reference), so the argument lies in [-pi/4, pi/4], where no reduction is needed and the odd minimax polynomial of the
// Cephes tanf (degree 13, <= 2 ulp there) does in 9 instructions what the library routine does in ~40 (reduction,
// rational kernel, division) -- twice per lane and sample in the race-car rollouts.  Anything larger takes the library.
__device__ __forceinline__ float tan_(float x) {
    if (__builtin_expect(fabsf(x) > 0.78539816f, 0)) return tanf(x);
    const float z = x * x;
    float p = 9.38540185543e-3f;
    p = fmaf(p, z, 3.11992232697e-3f);
    p = fmaf(p, z, 2.44301354525e-2f);
    p = fmaf(p, z, 5.34112807005e-2f);
    p = fmaf(p, z, 1.33387994085e-1f);
    p = fmaf(p, z, 3.33331568548e-1f);
    return fmaf(p * z, x, x);
}
__device__ __forceinline__ double tan_(double x) { return tan(x); }
// softmin weights exp(-beta (S - rho)) <= 1: the hardware exp2 path (2 ops, ~1e-6 relative) is ample for fp32 handles
__device__ __forceinline__ float exp_(float x) { return __expf(x); }
__device__ __forceinline__ double exp_(double x) { return exp(x); }

// Python's float `%` with a positive divisor (mppi_race_car.py:141): result in [0, m).
// fp32: a - m*floor(a/m) with one correction step instead of the exact iterative fmodf (a handful of ops instead of
// ~60; it can differ from fmodf by an ulp of a, i.e. ~5e-7 rad on a wrapped yaw -- far inside the fp32 tolerance)
__device__ __forceinline__ float pymod(float a, float m) {
    // (v_rcp_f32 + multiply: `__fdividef` compiles to the ten-instruction IEEE division without fast-math; an
    // off-by-one floor near a multiple of m is what the two correction steps below absorb)
    float r = fmaf(-m, floorf(a * __builtin_amdgcn_rcpf(m)), a);
    r = r < 0.f ? r + m : r;
    return r >= m ? r - m : r;
}
__device__ __forceinline__ double pymod(double a, double m) { const double r = fmod(a, m); return r < 0.0 ? r + m : r; }

template <typename R> __device__ __forceinline__ R clamp(R v, R lim) { return v < -lim ? -lim : (v > lim ? lim : v); }
template <> __device__ __forceinline__ float clamp<float>(float v, float lim) {  // one v_med3_f32
    return __builtin_amdgcn_fmed3f(v, -lim, lim);
}

}  // namespace mf
