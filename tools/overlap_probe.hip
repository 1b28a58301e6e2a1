// Experiment (not product): what would an iteration cost if its two launches overlapped their kernel boundaries?
//
// The closed loop is rollout (256 workgroups) -> finalize (one workgroup) -> rollout -> ...; each arrow is a kernel
// boundary of ~2 us (end-of-kernel writeback, dispatch, kernel-argument fetch, wave launch, cold first loads), and the
// two boundaries are half of config 2's 9 us hold-phase iteration.  Here two stand-in kernels of
// the same shape and roughly the same work run
//   mode 0: as the engine runs them -- one stream, stream order is the dependency;
//   mode 1: on TWO streams, every launch enqueued up front, the dependency carried by device-scope flags: R(i+1) is
//           dispatched while F(i) still runs, does the part of its work that needs nothing from F(i) (the noise draw),
//           then waits for F(i)'s flag; F(i+1) waits for a counter the 256 workgroups of R(i+1) raise;
//   mode 2: mode 1 with one counter per XCD-sized group of workgroups (32) instead of one address for all 256.
// Every wait is bounded (wall clock): a lost flag ends the run with an error instead of hanging the device.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o tools/_bin/overlap_probe && tools/_bin/overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr int N_WG = 256, REC = 104;  // records of 104 floats = 416 bytes, as config 2's
constexpr unsigned long long TIMEOUT_TICKS = 5000000ull;  // 50 ms at 100 MHz

struct Shared {
    int f_done;        // iterations the finalize stand-in has completed
    int abort_flag;    // a wait timed out
    int pad[14];
    int r_count[16];   // workgroups of the rollout stand-in that have finished (cumulative), one counter per group
    float state[16];
};

__device__ __forceinline__ bool wait_ge(const int *p, int target, Shared *sh) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (wall_clock64() - t0 > TIMEOUT_TICKS || __hip_atomic_load(&sh->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(&sh->abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

__device__ __forceinline__ float busy(float x, int n) {  // n dependent-ish FMAs per lane
    float a = x, b = 1.0001f;
#pragma unroll 8
    for (int i = 0; i < n; ++i) { a = fmaf(a, b, 0.5f); b = fmaf(b, 0.9999f, 1e-4f); }
    return a + b;
}

// rollout stand-in: 16 waves per workgroup, ~110 instructions that need nothing (the draw), then the state, ~170 more, a record
__global__ __launch_bounds__(1024) void k_r(Shared *sh, float *recs, int iter, int mode, int groups) {
    __shared__ float red[16];
    const int tid = threadIdx.x, wid = tid >> 6;
    float v = busy((float)(tid + blockIdx.x), 55);
    if (mode) {
        __shared__ int ok;
        if (tid == 0) ok = wait_ge(&sh->f_done, iter, sh) ? 1 : 0;
        __syncthreads();
        if (!ok) return;
    }
    const float s = sh->state[tid & 15];
    v = busy(v + s, 85);
    if ((tid & 63) == 0) red[wid] = v;
    __syncthreads();
    float *out = recs + (size_t)blockIdx.x * REC;
    if (tid < REC) out[tid] = red[tid & 15] + v;
    if (mode) {
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(&sh->r_count[groups > 1 ? (blockIdx.x * groups) / N_WG : 0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// finalize stand-in: 4 waves, reads the 256 records (106 KB), ~400 instructions, writes the state
__global__ __launch_bounds__(256) void k_f(Shared *sh, const float *recs, int iter, int mode, int groups) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    if (mode) {
        if (tid < groups) wait_ge(&sh->r_count[tid], (N_WG / groups) * (iter + 1), sh);  // (a timeout raises abort_flag)
        __syncthreads();
        if (__hip_atomic_load(&sh->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // (every load issued before the first use, as k_finalize does: one memory round trip for the 106 KB)
    const float4 *r4 = reinterpret_cast<const float4 *>(recs);
    float4 r[26];
#pragma unroll
    for (int j = 0; j < 26; ++j) r[j] = r4[tid + 256 * j];
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 26; ++j) acc += (r[j].x + r[j].y) + (r[j].z + r[j].w);
    acc = busy(acc, 120);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid < 16) sh->state[tid] = (red[0] + red[1] + red[2] + red[3]) * 1e-9f + (float)tid;
    if (mode) {
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(&sh->f_done, iter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char **argv) {
    const int n_iter = argc > 1 ? atoi(argv[1]) : 2000;
    Shared *sh;
    float *recs;
    CHECK(hipMalloc((void **)&sh, sizeof(Shared)));
    CHECK(hipMalloc((void **)&recs, sizeof(float) * N_WG * REC));
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        const int groups = mode == 2 ? 8 : 1;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(sh, 0, sizeof(Shared)));
            CHECK(hipMemset(recs, 0, sizeof(float) * N_WG * REC));
            CHECK(hipDeviceSynchronize());
            hipStream_t s_r = mode ? sb : sa;
            CHECK(hipEventRecord(e0, sa));
            if (mode) CHECK(hipStreamWaitEvent(sb, e0, 0));
            // (mode 1/2: at most a few hundred launches are enqueued ahead; the queues hold them)
            for (int i = 0; i < n_iter; ++i) {
                hipLaunchKernelGGL(k_r, dim3(N_WG), dim3(1024), 0, s_r, sh, recs, i, mode, groups);
                hipLaunchKernelGGL(k_f, dim3(1), dim3(256), 0, sa, sh, recs, i, mode, groups);
            }
            CHECK(hipEventRecord(e1, sa));
            CHECK(hipStreamSynchronize(sa));
            CHECK(hipStreamSynchronize(sb));
            CHECK(hipGetLastError());
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            Shared host;
            CHECK(hipMemcpy(&host, sh, sizeof(host), hipMemcpyDeviceToHost));
            printf("{\"mode\": %d, \"counters\": %d, \"rep\": %d, \"iterations\": %d, \"us_per_iteration\": %.3f, \"f_done\": %d, \"aborted\": %d}\n",
                   mode, groups, rep, n_iter, 1e3 * ms / n_iter, host.f_done, host.abort_flag);
            if (host.abort_flag) break;
        }
    }
    return 0;
}
