// Experiment (not product): does it matter whether a kernel's ~650 bytes of parameters arrive by value (kernel-argument
// segment, rewritten by the host for every launch) or through a pointer to a block that stays in device memory?
// Two stand-in kernels per iteration as in tools/overlap_probe.hip (256 x 1024 threads, then 1 x 256), stream order.
//   mode 0: 16 bytes of arguments (pointers only); mode 1: + a 640-byte struct by value, every word of it used;
//   mode 2: + a pointer to the same struct resident in device memory, every word of it used.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e6 * t.tv_sec + 1e-3 * t.tv_nsec; }
static double g_host_us;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Big { unsigned v[160]; };
__device__ __forceinline__ float busy(float x, int n) {
    float a = x, b = 1.0001f;
#pragma unroll 8
    for (int i = 0; i < n; ++i) { a = fmaf(a, b, 0.5f); b = fmaf(b, 0.9999f, 1e-4f); }
    return a + b;
}
__device__ __forceinline__ float sum_all(const Big &p) {  // (scalar unit: 160 one-cycle instructions)
    unsigned s = 0u;
#pragma unroll
    for (int i = 0; i < 160; ++i) s ^= p.v[i];
    return (float)(s & 1023u);
}
template <int MODE> __global__ __launch_bounds__(1024) void k_r(float *state, float *recs, Big byval, const Big *resident) {
    float v = busy((float)(threadIdx.x + blockIdx.x), 40) + state[threadIdx.x & 15];
    if (MODE == 1) v += sum_all(byval);
    if (MODE == 2) v += sum_all(*resident);
    v = busy(v, 60);
    if (threadIdx.x < 104) recs[(size_t)blockIdx.x * 104 + threadIdx.x] = v;
}
template <int MODE> __global__ __launch_bounds__(256) void k_f(float *state, const float *recs, Big byval, const Big *resident) {
    const float4 *r4 = reinterpret_cast<const float4 *>(recs);
    float4 r[26];
#pragma unroll
    for (int j = 0; j < 26; ++j) r[j] = r4[threadIdx.x + 256 * j];
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 26; ++j) acc += (r[j].x + r[j].y) + (r[j].z + r[j].w);
    if (MODE == 1) acc += sum_all(byval);
    if (MODE == 2) acc += sum_all(*resident);
    acc = busy(acc, 120);
    if (threadIdx.x < 16) state[threadIdx.x] = acc * 1e-9f + (float)threadIdx.x;
}
template <int MODE> static float run(float *state, float *recs, const Big &host, const Big *dev, int n, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    CHECK(hipEventRecord(e0, s));
    const double h0 = now_us();
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(k_r<MODE>, dim3(256), dim3(1024), 0, s, state, recs, host, dev);
        hipLaunchKernelGGL(k_f<MODE>, dim3(1), dim3(256), 0, s, state, recs, host, dev);
    }
    g_host_us = (now_us() - h0) / n;
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / n;
}
// the same through a captured graph of 50 iterations, replayed
template <int MODE> static float run_graph(float *state, float *recs, const Big &host, const Big *dev, int n, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 50; ++i) {
        hipLaunchKernelGGL(k_r<MODE>, dim3(256), dim3(1024), 0, s, state, recs, host, dev);
        hipLaunchKernelGGL(k_f<MODE>, dim3(1), dim3(256), 0, s, state, recs, host, dev);
    }
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < n / 50; ++i) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    return 1e3f * ms / (50 * (n / 50));
}
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 3000;
    float *state, *recs;
    Big host, *dev;
    for (int i = 0; i < 160; ++i) host.v[i] = 2654435761u * (unsigned)i;
    CHECK(hipMalloc((void **)&state, 64));
    CHECK(hipMalloc((void **)&recs, sizeof(float) * 256 * 104));
    CHECK(hipMalloc((void **)&dev, sizeof(Big)));
    CHECK(hipMemcpy(dev, &host, sizeof(Big), hipMemcpyHostToDevice));
    CHECK(hipMemset(state, 0, 64));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        const float a = run<0>(state, recs, host, dev, n, s, e0, e1); const double ha = g_host_us;
        const float b = run<1>(state, recs, host, dev, n, s, e0, e1); const double hb = g_host_us;
        const float c = run<2>(state, recs, host, dev, n, s, e0, e1); const double hc = g_host_us;
        printf("{\"rep\": %d, \"host_enqueue_us_per_iteration\": [%.3f, %.3f, %.3f], \"graph_us_per_iteration\": [%.3f, %.3f, %.3f]}\n", rep, ha, hb, hc,
               run_graph<0>(state, recs, host, dev, n, s, e0, e1), run_graph<1>(state, recs, host, dev, n, s, e0, e1), run_graph<2>(state, recs, host, dev, n, s, e0, e1));
        printf("{\"rep\": %d, \"iterations\": %d, \"us_per_iteration\": {\"pointers_only\": %.3f, \"640B_by_value\": %.3f, \"640B_resident_in_device_memory\": %.3f}}\n", rep, n, a, b, c);
    }
    return 0;
}
