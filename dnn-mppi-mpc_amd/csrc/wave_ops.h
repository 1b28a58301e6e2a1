// Wave64 cross-lane primitives for gfx950 (CDNA4).  One wavefront = 64 lanes = 4 DPP rows of 16.
// Scans/reductions are DPP row shifts + row broadcasts (6 VALU ops, no LDS traffic).
#pragma once
#include <hip/hip_runtime.h>

namespace wv {

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

__device__ __forceinline__ int lane_id() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Returns the DPP-moved `x`; lanes the move does not write (masked rows/banks, shifted-in lanes) get `fill`.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp(float x, float fill) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill),
                                                                  __builtin_bit_cast(int, x), CTRL, ROW_MASK,
                                                                  BANK_MASK, false));
}
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int dpp(int x, int fill) {
    return __builtin_amdgcn_update_dpp(fill, x, CTRL, ROW_MASK, BANK_MASK, false);
}
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ double dpp(double x, double fill) {
    const long long xb = __builtin_bit_cast(long long, x), fb = __builtin_bit_cast(long long, fill);
    const int lo = __builtin_amdgcn_update_dpp((int)fb, (int)xb, CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(fb >> 32), (int)(xb >> 32), CTRL, ROW_MASK, BANK_MASK, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

struct OpAdd {
    template <typename R> static __device__ __forceinline__ R apply(R a, R b) { return a + b; }
    template <typename R> static __device__ __forceinline__ R identity() { return R(0); }
};
struct OpMin {  // fmin: one v_min with the DPP move folded in (a compare + select would be three instructions)
    template <typename R> static __device__ __forceinline__ R apply(R a, R b) { return fmin(a, b); }
    template <typename R> static __device__ __forceinline__ R identity() { return R(INFINITY); }
};
struct OpMinInt {
    static __device__ __forceinline__ int apply(int a, int b) { return b < a ? b : a; }
    template <typename R> static __device__ __forceinline__ int identity() { return 0x7fffffff; }
};

struct OpMaxInt {
    static __device__ __forceinline__ int apply(int a, int b) { return b > a ? b : a; }
    template <typename R> static __device__ __forceinline__ int identity() { return (int)0x80000000; }
};

// Inclusive scan over the 64 lanes (lane i <- op(x[0..i])).
template <typename Op, typename R> __device__ __forceinline__ R scan_incl(R x) {
    const R id = Op::template identity<R>();
    x = Op::apply(x, dpp<DPP_ROW_SHR1>(x, id));
    x = Op::apply(x, dpp<DPP_ROW_SHR2>(x, id));
    x = Op::apply(x, dpp<DPP_ROW_SHR4>(x, id));
    x = Op::apply(x, dpp<DPP_ROW_SHR8>(x, id));
    x = Op::apply(x, dpp<DPP_ROW_BCAST15, 0xA>(x, id));  // lane 15/47 -> rows 1/3
    x = Op::apply(x, dpp<DPP_ROW_BCAST31, 0xC>(x, id));  // lane 31 -> rows 2,3
    return x;
}

// The same scan over each 32-lane half of the wave separately (two segments: lanes 0-31, 32-63).
template <typename Op, typename R> __device__ __forceinline__ R scan_incl_half(R x) {
    const R id = Op::template identity<R>();
    x = Op::apply(x, dpp<DPP_ROW_SHR1>(x, id));
    x = Op::apply(x, dpp<DPP_ROW_SHR2>(x, id));
    x = Op::apply(x, dpp<DPP_ROW_SHR4>(x, id));
    x = Op::apply(x, dpp<DPP_ROW_SHR8>(x, id));
    x = Op::apply(x, dpp<DPP_ROW_BCAST15, 0xA>(x, id));  // lane 15 -> row 1, lane 47 -> row 3
    return x;
}

// op over the 16 lanes of each DPP row (lane 15 / 31 / 47 / 63 of a row ends up with the row's result): 4 VALU ops
template <typename Op, typename R> __device__ __forceinline__ R scan_incl_row(R x) {
    const R id = Op::template identity<R>();
    x = Op::apply(x, dpp<DPP_ROW_SHR1>(x, id));
    x = Op::apply(x, dpp<DPP_ROW_SHR2>(x, id));
    x = Op::apply(x, dpp<DPP_ROW_SHR4>(x, id));
    x = Op::apply(x, dpp<DPP_ROW_SHR8>(x, id));
    return x;
}

// SEG segments per wave (1: the whole wave, 2: its 32-lane halves)
template <typename Op, int SEG, typename R> __device__ __forceinline__ R scan_incl_seg(R x) {
    return SEG == 2 ? scan_incl_half<Op>(x) : scan_incl<Op>(x);
}

// lane i <- x[i-1], lane 0 <- carry.
template <typename R> __device__ __forceinline__ R shift_up1(R x, R carry) { return dpp<DPP_WAVE_SHR1>(x, carry); }

// lane i <- x[i-1] within each 32-lane half; the first lane of each half <- carry
template <typename R> __device__ __forceinline__ R shift_up1_half(R x, R carry) {
    const R sh = dpp<DPP_WAVE_SHR1>(x, carry);
    return (lane_id() & 31) == 0 ? carry : sh;
}

template <int SEG, typename R> __device__ __forceinline__ R shift_up1_seg(R x, R carry) {
    return SEG == 2 ? shift_up1_half(x, carry) : shift_up1(x, carry);
}

__device__ __forceinline__ float read_lane(float x, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
__device__ __forceinline__ int read_lane(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }
__device__ __forceinline__ double read_lane(double x, int lane) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// acc + v[first] + v[first + 1] + ... (n lanes) added strictly in that order: fp addition does not associate and some
// callers must reproduce a sequential `+=` (see Rollout::chunk).  The lane reads are batched eight at a time so that
// only the chain of additions is serial (one read, select and add per trip cost ~120 clocks per term).
template <typename R> __device__ __forceinline__ R ordered_sum(R acc, R v, int first, int n) {
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        R a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = read_lane(v, first + i + j);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += a[j];
    }
    for (; i < n; ++i) acc += read_lane(v, first + i);
    return acc;
}
template <typename Op, typename R> __device__ __forceinline__ R reduce(R x) {
    return read_lane(scan_incl<Op>(x), 63);
}

__device__ __forceinline__ float shfl_xor(float x, int m) { return __shfl_xor(x, m); }
__device__ __forceinline__ double shfl_xor(double x, int m) { return __shfl_xor(x, m); }

// (d, j) -> the pair with the smallest d over the wave; ties go to the smaller j
// (np.argmin / list.index semantics: first minimum).  Every lane gets the result.
// Two DPP reductions (min of d, then min of j among the lanes that hold it): 12 VALU ops, no LDS traffic.
template <typename R> __device__ __forceinline__ void argmin_first(R &d, int &j) {
    const R m = reduce<OpMin>(d);
    j = reduce<OpMinInt>(d == m ? j : 0x7fffffff);
    d = m;
}

}  // namespace wv
