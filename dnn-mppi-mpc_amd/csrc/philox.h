// Counter-based control-noise sampler (stage S1).  Stands in for
// np.random.multivariate_normal (controllers/mppi_differential_drive.py:273-283): Philox4x32-10
// keyed by the seed, counter = (global sample k, t >> 1, iteration, stream = agent / noise_stream); words (r0,r1)
// serve even t and (r2,r3) odd t; Box-Muller; 2x2 Cholesky factor.  A sample depends only
// on (seed, iteration, k_global, t), so K can be sharded over ranks without changing the
// draw.  oracle/philox.py restates this in NumPy.
#pragma once
#include <hip/hip_runtime.h>
#include "mathfn.h"

namespace px {

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: integer
        // multiplies are quarter-rate, and the 40 of them were a third of the rollout's VALU time
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float uniform_open(unsigned r) {  // (2*(r>>9)+1) * 2^-24, exact in f32
    return (float)(2u * (r >> 9) + 1u) * 5.9604644775390625e-8f;
}

// one Box-Muller pair from two Philox words, then the 2x2 Cholesky factor chol = {L00, L10, L11}
__device__ __forceinline__ void box_muller(unsigned ra, unsigned rb, const float (&chol)[3], float &e0, float &e1) {
    // hardware log2 / sqrt (1 ulp each): the radius is good to ~3e-7 relative, far inside the sampler's
    // tolerance against its NumPy restatement, at 4 VALU ops instead of ~40
    // (raw v_log_f32 = log2: the argument is in [2^-24, 1], never denormal, so the library's rescaling is dead weight)
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(uniform_open(ra)));
    float s, c;
    mf::sincos_turns(uniform_open(rb), s, c);
    const float z0 = rad * c, z1 = rad * s;
    e0 = chol[0] * z0;
    e1 = fmaf(chol[1], z0, chol[2] * z1);
}

// eps[k_global, t, 0..1] ~ N(0, L L^T), chol = {L00, L10, L11}
// (`stream`: the counter's fourth word -- 0, or the agent of a batched handle / mppi_config.noise_stream)
__device__ __forceinline__ void sample(unsigned seed_lo, unsigned seed_hi, unsigned iter, unsigned k_global, int t,
                                       const float (&chol)[3], float &e0, float &e1, unsigned stream = 0u) {
    unsigned r[4];
    philox4x32_10(k_global, (unsigned)t >> 1, iter, stream, seed_lo, seed_hi, r);
    const unsigned ra = (t & 1) ? r[2] : r[0], rb = (t & 1) ? r[3] : r[1];
    box_muller(ra, rb, chol, e0, e1);
}

}  // namespace px
