// Issue cost (shader cycles per wave-instruction) of the VALU instructions the rollout kernels lean on, one wave per SIMD and
// four (the config-2 rollout launch runs four): a stream of independent instructions between two s_memtime reads.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_cost tools/issue_cost.hip && /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x

#define KERNEL(name, body)                                                                          \
    __global__ __launch_bounds__(1024) void name(unsigned long long *out, unsigned seed) {          \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;                   \
        unsigned b0 = a0 ^ 0x55u, b1 = a1 ^ 0x33u, b2 = a2 ^ 0x77u, b3 = a3 ^ 0x11u;                  \
        unsigned long long w0 = a0, w1 = a1, w2 = a2, w3 = a3;                                       \
        float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;                         \
        typedef float f2v __attribute__((ext_vector_type(2)));                                       \
        f2v p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f1, f0}, p3 = {f3, f2};                               \
        __syncthreads();                                                                             \
        const unsigned long long t0 = __builtin_readcyclecounter();                                  \
        for (int i = 0; i < 64; ++i) { REP16(body) }                                                  \
        const unsigned long long t1 = __builtin_readcyclecounter();                                  \
        unsigned sink = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3 ^ (unsigned)w0 ^ (unsigned)w1 ^ (unsigned)w2 ^ (unsigned)w3 ^ \
                        (unsigned)(w0 >> 32) ^ (unsigned)(w1 >> 32) ^ (unsigned)(w2 >> 32) ^ (unsigned)(w3 >> 32) ^          \
                        __float_as_uint(f0 + f1 + f2 + f3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y);          \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;            \
        if (sink == 0x12345u) out[1023] = sink;                                                      \
    }

// four independent chains per body (4 instructions)
KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %6\n v_xor_b32 %3, %3, %7"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_xor_dpp, asm volatile("v_xor_b32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_xor_b32_dpp %1, %5, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_xor_b32_dpp %2, %6, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_xor_b32_dpp %3, %7, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_mad64, asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, 0\n v_mad_u64_u32 %1, vcc, %5, %8, 0\n"
                             "v_mad_u64_u32 %2, vcc, %6, %8, 0\n v_mad_u64_u32 %3, vcc, %7, %8, 0"
                             : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(a0) : "vcc");)
KERNEL(k_mulhi, asm volatile("v_mul_hi_u32 %0, %4, %8\n v_mul_hi_u32 %1, %5, %8\n v_mul_hi_u32 %2, %6, %8\n v_mul_hi_u32 %3, %7, %8"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(seed));)
KERNEL(k_mullo, asm volatile("v_mul_lo_u32 %0, %4, %8\n v_mul_lo_u32 %1, %5, %8\n v_mul_lo_u32 %2, %6, %8\n v_mul_lo_u32 %3, %7, %8"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(seed));)
KERNEL(k_mul24, asm volatile("v_mul_u32_u24 %0, %4, %8\n v_mul_u32_u24 %1, %5, %8\n v_mul_u32_u24 %2, %6, %8\n v_mul_u32_u24 %3, %7, %8"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(seed));)
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                           : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1));)
KERNEL(k_pk_fma, asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(w0), "v"(w1));)
KERNEL(k_pk_mul, asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(w0));)
KERNEL(k_exp, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3"
                           : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_sin, asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3"
                           : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_mov_dpp, asm volatile("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 row_shr:1 row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_add_dpp, asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                               "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_add_dpp_dep, asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                                   "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                                   : "+v"(f0));)
KERNEL(k_readlane, asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9"
                                : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %7, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
KERNEL(k_cnd_vcc_set, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %7, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
KERNEL(k_cnd_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %5, s[20:21]\n v_cndmask_b32_e64 %2, %2, %6, s[20:21]\n v_cndmask_b32_e64 %3, %3, %7, s[20:21]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s20", "s21");)
KERNEL(k_cnd_newdst, asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %6, vcc\n v_cndmask_b32 %2, %6, %7, vcc\n v_cndmask_b32 %3, %7, %4, vcc"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
KERNEL(k_cmp_vcc, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0"
                               : : "v"(f0), "v"(f1), "v"(f2), "v"(f3) : "vcc");)
KERNEL(k_cmp_e64, asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cmp_lt_f32_e64 s[22:23], %1, %2\n v_cmp_lt_f32_e64 s[24:25], %2, %3\n v_cmp_lt_f32_e64 s[26:27], %3, %0"
                               : : "v"(f0), "v"(f1), "v"(f2), "v"(f3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
KERNEL(k_cmp_cnd, asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %5\n v_cndmask_b32 %1, %1, %5, vcc"
                               : "+v"(f0), "+v"(f1) : "v"(f2), "v"(f3), "v"(b0), "v"(b1) : "vcc");)
KERNEL(k_med3, asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5"
                           : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1));)
KERNEL(k_max, asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4"
                           : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0));)
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                           : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0));)
KERNEL(k_mul_sgpr, asm volatile("v_mul_f32 %0, s20, %0\n v_mul_f32 %1, s21, %1\n v_mul_f32 %2, s22, %2\n v_mul_f32 %3, s23, %3"
                           : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : : "s20", "s21", "s22", "s23");)
KERNEL(k_mul_lit, asm volatile("v_mul_f32 %0, 0x3f9d70a4, %0\n v_mul_f32 %1, 0x3f9d70a4, %1\n v_mul_f32 %2, 0x3f9d70a4, %2\n v_mul_f32 %3, 0x3f9d70a4, %3"
                           : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_addu, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_lshladd, asm volatile("v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7"
                           : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_cvt, asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %6\n v_cvt_f32_u32 %3, %7"
                           : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_log, asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3"
                           : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                           : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_sqrt, asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3"
                           : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_fma64, asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4"
                           : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(p0));)
KERNEL(k_salu, asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1"
                           : : : "s20", "s21", "s22", "s23", "scc");)
KERNEL(k_salu_valu, asm volatile("s_add_u32 s20, s20, 1\n v_xor_b32 %0, %0, %4\n s_add_u32 s21, s21, 1\n v_xor_b32 %1, %1, %5"
                           : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3), "v"(b0), "v"(b1) : "s20", "s21", "scc");)
KERNEL(k_readfirst, asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3"
                                : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");)
KERNEL(k_bperm, asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_swap, asm volatile("v_swap_b32 %0, %1\n v_swap_b32 %2, %3\n v_swap_b32 %0, %2\n v_swap_b32 %1, %3"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)

KERNEL(k2_add, asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_sub, asm volatile("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %5\n v_sub_f32 %2, %2, %6\n v_sub_f32 %3, %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_fmac, asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %5, %6\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %7, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_fmaak, asm volatile("v_fmaak_f32 %0, %0, %4, 0x3d2aaaa8\n v_fmaak_f32 %1, %1, %5, 0x3d2aaaa8\n v_fmaak_f32 %2, %2, %6, 0x3d2aaaa8\n v_fmaak_f32 %3, %3, %7, 0x3d2aaaa8" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_fmamk, asm volatile("v_fmamk_f32 %0, %0, 0x3d2aaaa8, %4\n v_fmamk_f32 %1, %1, 0x3d2aaaa8, %5\n v_fmamk_f32 %2, %2, 0x3d2aaaa8, %6\n v_fmamk_f32 %3, %3, 0x3d2aaaa8, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_fmas, asm volatile("v_fma_f32 %0, %0, s20, %4\n v_fma_f32 %1, %1, s20, %5\n v_fma_f32 %2, %2, s20, %6\n v_fma_f32 %3, %3, s20, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s20");)
KERNEL(k2_adds, asm volatile("v_add_f32 %0, s20, %0\n v_add_f32 %1, s20, %1\n v_add_f32 %2, s20, %2\n v_add_f32 %3, s20, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s20");)
KERNEL(k2_min, asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %5\n v_min_f32 %2, %2, %6\n v_min_f32 %3, %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_max64, asm volatile("v_max_f32_e64 %0, %0, %4\n v_max_f32_e64 %1, %1, %5\n v_max_f32_e64 %2, %2, %6\n v_max_f32_e64 %3, %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_and, asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %6\n v_and_b32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_or, asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %5\n v_or_b32 %2, %2, %6\n v_or_b32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_xorl, asm volatile("v_xor_b32 %0, 0x9e3779b9, %0\n v_xor_b32 %1, 0x9e3779b9, %1\n v_xor_b32 %2, 0x9e3779b9, %2\n v_xor_b32 %3, 0x9e3779b9, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_xors, asm volatile("v_xor_b32 %0, s20, %0\n v_xor_b32 %1, s20, %1\n v_xor_b32 %2, s20, %2\n v_xor_b32 %3, s20, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s20");)
KERNEL(k2_shl, asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_shr, asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_andor, asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %5, %6\n v_and_or_b32 %2, %2, %6, %7\n v_and_or_b32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_bfe, asm volatile("v_bfe_u32 %0, %0, 3, 5\n v_bfe_u32 %1, %1, 3, 5\n v_bfe_u32 %2, %2, 3, 5\n v_bfe_u32 %3, %3, 3, 5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_align, asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %5, 7\n v_alignbit_b32 %2, %2, %6, 7\n v_alignbit_b32 %3, %3, %7, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_cvti, asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_rnd, asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_movs, asm volatile("v_mov_b32 %0, s20\n v_mov_b32 %1, s20\n v_mov_b32 %2, s20\n v_mov_b32 %3, s20" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s20");)
KERNEL(k2_movl, asm volatile("v_mov_b32 %0, 0x3d2aaaa8\n v_mov_b32 %1, 0x3d2aaaa8\n v_mov_b32 %2, 0x3d2aaaa8\n v_mov_b32 %3, 0x3d2aaaa8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_mov0, asm volatile("v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n v_mov_b32 %2, 0\n v_mov_b32 %3, 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_subu, asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %5\n v_sub_u32 %2, %2, %6\n v_sub_u32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_mini, asm volatile("v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %5\n v_min_i32 %2, %2, %6\n v_min_i32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_min3, asm volatile("v_min3_i32 %0, %0, %4, %5\n v_min3_i32 %1, %1, %5, %6\n v_min3_i32 %2, %2, %6, %7\n v_min3_i32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_add3, asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %6\n v_add3_u32 %2, %2, %6, %7\n v_add3_u32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_mulneg, asm volatile("v_mul_f32_e64 %0, -%0, %4\n v_mul_f32_e64 %1, -%1, %5\n v_mul_f32_e64 %2, -%2, %6\n v_mul_f32_e64 %3, -%3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_fmaabs, asm volatile("v_fma_f32 %0, |%0|, %4, %5\n v_fma_f32 %1, |%1|, %5, %6\n v_fma_f32 %2, |%2|, %6, %7\n v_fma_f32 %3, |%3|, %7, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_pkadd, asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %6\n v_pk_add_f32 %3, %3, %7" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(w0), "v"(w1), "v"(w2), "v"(w3));)
KERNEL(k2_cvtd, asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(w0), "v"(w1), "v"(w2), "v"(w3));)
KERNEL(k2_addd, asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %5\n v_add_f64 %2, %2, %6\n v_add_f64 %3, %3, %7" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3));)
KERNEL(k2_muld, asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %5\n v_mul_f64 %2, %2, %6\n v_mul_f64 %3, %3, %7" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3));)
KERNEL(k2_dsr, asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_dsw, asm volatile("ds_write_b32 %4, %0\n ds_write_b32 %5, %1\n ds_write_b32 %6, %2\n ds_write_b32 %7, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_cmpcnd64, asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cmp_lt_f32_e64 s[20:21], %1, %5\n v_cndmask_b32_e64 %1, %1, %5, s[20:21]\n v_cmp_lt_f32_e64 s[20:21], %2, %6\n v_cndmask_b32_e64 %2, %2, %6, s[20:21]\n v_cmp_lt_f32_e64 s[20:21], %3, %7\n v_cndmask_b32_e64 %3, %3, %7, s[20:21]" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s20", "s21");)
KERNEL(k2_perm, asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %5, %6\n v_perm_b32 %2, %2, %6, %7\n v_perm_b32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_mbcnt, asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0\n v_mbcnt_lo_u32_b32 %1, -1, %1\n v_mbcnt_lo_u32_b32 %2, -1, %2\n v_mbcnt_lo_u32_b32 %3, -1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_ldexp, asm volatile("v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %5\n v_ldexp_f32 %2, %2, %6\n v_ldexp_f32 %3, %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_frexp, asm volatile("v_frexp_mant_f32 %0, %0\n v_frexp_mant_f32 %1, %1\n v_frexp_mant_f32 %2, %2\n v_frexp_mant_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_fract, asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_cos, asm volatile("v_cos_f32 %0, %0\n v_cos_f32 %1, %1\n v_cos_f32 %2, %2\n v_cos_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k2_rsq, asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)

// v_cndmask_b32 reading a mask that was written (a) by a VALU compare long ago, (b) by the SALU long ago, (c) by the SALU just before
#define KERNEL_PRE(name, pre, body)                                                                  \
    __global__ __launch_bounds__(1024) void name(unsigned long long *out, unsigned seed) {          \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;                   \
        unsigned b0 = a0 ^ 0x55u, b1 = a1 ^ 0x33u, b2 = a2 ^ 0x77u, b3 = a3 ^ 0x11u;                  \
        __syncthreads();                                                                             \
        pre;                                                                                         \
        const unsigned long long t0 = __builtin_readcyclecounter();                                  \
        for (int i = 0; i < 64; ++i) { REP16(body) }                                                  \
        const unsigned long long t1 = __builtin_readcyclecounter();                                  \
        unsigned sink = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3;                                        \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;            \
        if (sink == 0x12345u) out[1023] = sink;                                                      \
    }
#define CND4_VCC asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %7, vcc" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");
#define CND4_S asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %5, s[20:21]\n v_cndmask_b32_e64 %2, %2, %6, s[20:21]\n v_cndmask_b32_e64 %3, %3, %7, s[20:21]" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s20", "s21");
KERNEL_PRE(k3_vcc_valu_old, asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(b1) : "vcc"), CND4_VCC)
KERNEL_PRE(k3_vcc_salu_old, asm volatile("s_mov_b64 vcc, 0x5555" : : : "vcc"), CND4_VCC)
KERNEL_PRE(k3_vcc_salu_now, , asm volatile("s_not_b64 vcc, vcc" : : : "vcc", "scc"); CND4_VCC)
KERNEL_PRE(k3_vcc_valu_now, , asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a0), "v"(b1) : "vcc"); CND4_VCC)
KERNEL_PRE(k3_s_valu_old, asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1" : : "v"(a0), "v"(b1) : "s20", "s21"), CND4_S)
KERNEL_PRE(k3_s_salu_now, , asm volatile("s_not_b64 s[20:21], s[20:21]" : : : "s20", "s21", "scc"); CND4_S)
KERNEL_PRE(k3_vcc_mixed, asm volatile("s_mov_b64 vcc, 0x5555" : : : "vcc"), asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %6\n v_xor_b32 %3, %3, %7" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)

KERNEL_PRE(k4_cmp_2cnd, , asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_u32 vcc, %2, %6\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %7, vcc" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
KERNEL_PRE(k4_cmp_2cnd_e64, , asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %4\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %5, s[20:21]\n v_cmp_lt_u32_e64 s[22:23], %2, %6\n v_cndmask_b32_e64 %2, %2, %6, s[22:23]\n v_cndmask_b32_e64 %3, %3, %7, s[22:23]" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s20", "s21", "s22", "s23");)
KERNEL_PRE(k4_cmp_cnd_x_cnd, , asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_xor_b32 %2, %2, %6\n v_cndmask_b32 %1, %1, %5, vcc\n v_xor_b32 %3, %3, %7\n v_xor_b32 %2, %2, %7" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
KERNEL_PRE(k4_min_cmp_cnd, , asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_min_u32 %0, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_u32 vcc, %2, %6\n v_min_u32 %2, %2, %6\n v_cndmask_b32 %3, %3, %7, vcc" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)

template <typename K> static void run(const char *name, K kern, int per_body) {
    unsigned long long *d, h[4096];
    (void)hipMalloc(&d, sizeof(h));
    for (int waves : {4, 16}) {  // one wave per SIMD, four per SIMD (one workgroup on one CU)
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipMemset(d, 0, sizeof(h));
            hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, d, 12345u);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0;
            for (int w = 0; w < waves; ++w) m = h[w] > m ? (double)h[w] : m;
            best = m < best ? m : best;
        }
        // s_memtime counts at 100 MHz on this part (10 ns ticks)?  report raw ticks too
        printf("%-14s %2d waves/CU: %9.0f ticks for %d instr/wave -> %.3f ticks/instr, x waves/SIMD %d -> %.3f per SIMD-instr\n", name, waves,
               best, 64 * 16 * per_body, best / (64.0 * 16 * per_body), waves / 4, best / (64.0 * 16 * per_body) / (waves / 4));
    }
    (void)hipFree(d);
}

int main() {
    run("xor", k_xor, 4);
    run("xor_dpp", k_xor_dpp, 4);
    run("mad_u64_u32", k_mad64, 4);
    run("mul_hi_u32", k_mulhi, 4);
    run("mul_lo_u32", k_mullo, 4);
    run("mul_u32_u24", k_mul24, 4);
    run("fma_f32", k_fma, 4);
    run("pk_fma_f32", k_pk_fma, 4);
    run("pk_mul_f32", k_pk_mul, 4);
    run("exp_f32", k_exp, 4);
    run("sin_f32", k_sin, 4);
    run("mov_dpp", k_mov_dpp, 4);
    run("add_dpp", k_add_dpp, 4);
    run("add_dpp dep", k_add_dpp_dep, 4);
    run("readlane", k_readlane, 4);
    run("cndmask", k_cndmask, 4);
    run("cnd vcc(set)", k_cnd_vcc_set, 4);
    run("cnd e64 sgpr", k_cnd_e64, 4);
    run("cnd new dst", k_cnd_newdst, 4);
    run("cmp vcc", k_cmp_vcc, 4);
    run("cmp e64", k_cmp_e64, 4);
    run("cmp+cnd", k_cmp_cnd, 4);
    run("med3_f32", k_med3, 4);
    run("max_f32", k_max, 4);
    run("mul_f32", k_mul, 4);
    run("mul_f32 sgpr", k_mul_sgpr, 4);
    run("mul_f32 lit", k_mul_lit, 4);
    run("add_u32", k_addu, 4);
    run("lshl_add_u32", k_lshladd, 4);
    run("mov_b32", k_mov, 4);
    run("cvt_f32_u32", k_cvt, 4);
    run("log_f32", k_log, 4);
    run("rcp_f32", k_rcp, 4);
    run("sqrt_f32", k_sqrt, 4);
    run("fma_f64", k_fma64, 4);
    run("s_add_u32", k_salu, 4);
    run("salu+valu", k_salu_valu, 4);
    run("readfirstlane", k_readfirst, 4);
    run("ds_bpermute", k_bperm, 4);
    run("swap_b32", k_swap, 4);
    run("add_f32", k2_add, 4);
    run("sub_f32", k2_sub, 4);
    run("fmac_f32", k2_fmac, 4);
    run("fmaak lit", k2_fmaak, 4);
    run("fmamk lit", k2_fmamk, 4);
    run("fma sgpr", k2_fmas, 4);
    run("add_f32 sgpr", k2_adds, 4);
    run("min_f32", k2_min, 4);
    run("max_f32 e64", k2_max64, 4);
    run("and_b32", k2_and, 4);
    run("or_b32", k2_or, 4);
    run("xor lit", k2_xorl, 4);
    run("xor sgpr", k2_xors, 4);
    run("lshlrev", k2_shl, 4);
    run("lshrrev", k2_shr, 4);
    run("and_or", k2_andor, 4);
    run("bfe_u32", k2_bfe, 4);
    run("alignbit", k2_align, 4);
    run("cvt_i32_f32", k2_cvti, 4);
    run("rndne", k2_rnd, 4);
    run("mov sgpr", k2_movs, 4);
    run("mov lit", k2_movl, 4);
    run("mov inline0", k2_mov0, 4);
    run("sub_u32", k2_subu, 4);
    run("min_i32", k2_mini, 4);
    run("min3_i32", k2_min3, 4);
    run("add3_u32", k2_add3, 4);
    run("mul_f32 e64 neg", k2_mulneg, 4);
    run("fma_f32 abs", k2_fmaabs, 4);
    run("pk_add_f32", k2_pkadd, 4);
    run("cvt_f32_f64", k2_cvtd, 4);
    run("add_f64", k2_addd, 4);
    run("mul_f64", k2_muld, 4);
    run("ds_read_b32", k2_dsr, 4);
    run("ds_write_b32", k2_dsw, 4);
    run("v_cmp+cnd e64", k2_cmpcnd64, 8);
    run("perm_b32", k2_perm, 4);
    run("mbcnt", k2_mbcnt, 4);
    run("exp f16?", k2_ldexp, 4);
    run("frexp_mant", k2_frexp, 4);
    run("fract", k2_fract, 4);
    run("cos_f32", k2_cos, 4);
    run("rsq_f32", k2_rsq, 4);
    run("cnd vcc valu-old", k3_vcc_valu_old, 4);
    run("cnd vcc salu-old", k3_vcc_salu_old, 4);
    run("cnd vcc salu-now(5)", k3_vcc_salu_now, 5);
    run("cnd vcc valu-now(5)", k3_vcc_valu_now, 5);
    run("cnd s valu-old", k3_s_valu_old, 4);
    run("cnd s salu-now(5)", k3_s_salu_now, 5);
    run("1cnd+3xor vcc", k3_vcc_mixed, 4);
    run("cmp+2cnd vcc (6)", k4_cmp_2cnd, 6);
    run("cmp+2cnd e64 (6)", k4_cmp_2cnd_e64, 6);
    run("cmp cnd x cnd x x (6)", k4_cmp_cnd_x_cnd, 6);
    run("cmp min cnd (6)", k4_min_cmp_cnd, 6);
    return 0;
}
