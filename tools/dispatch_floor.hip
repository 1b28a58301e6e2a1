// Diagnostic (not product): what does a kernel boundary cost on this GPU?  Back-to-back dependent launches on one
// stream of (a) an empty 1-workgroup kernel, (b) an empty 256 x 1024-thread kernel, (c) the two alternating -- the launch
// shape of an MPPI iteration (full-chip rollout, one-workgroup finalize) with nothing inside.  Prints microseconds per
// launch.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_floor.hip -o gpurun_out/dispatch_floor && gpurun_out/dispatch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void k_empty(int *p) {
    if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1;
}
__global__ void k_touch(float *p, int n) {  // every workgroup writes one cache line, like a record
    if (threadIdx.x < 32) p[(blockIdx.x * 32 + threadIdx.x) % n] = 1.0f;
}

template <typename F> static double per_launch_us(F &&body, int n) {
    body(200);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    body(n);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
    float *buf;
    hipMalloc(&buf, 1 << 20);
    const int N = 20000;
    const double a = per_launch_us([&](int n) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, 0, nullptr); }, N);
    const double b = per_launch_us([&](int n) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, 0, nullptr); }, N);
    const double c = per_launch_us([&](int n) {
        for (int i = 0; i < n; i += 2) {
            hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, 0, nullptr);
            hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, 0, nullptr);
        }
    }, N);
    const double d = per_launch_us([&](int n) {
        for (int i = 0; i < n; i += 2) {
            hipLaunchKernelGGL(k_touch, dim3(256), dim3(1024), 0, 0, buf, 1 << 18);
            hipLaunchKernelGGL(k_touch, dim3(1), dim3(256), 0, 0, buf, 1 << 18);
        }
    }, N);
    // the same alternating pair replayed from a HIP graph: no host call per launch, the GPU-side floor
    hipStream_t st;
    hipStreamCreate(&st);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 1000; i += 2) {
        hipLaunchKernelGGL(k_touch, dim3(256), dim3(1024), 0, st, buf, 1 << 18);
        hipLaunchKernelGGL(k_touch, dim3(1), dim3(256), 0, st, buf, 1 << 18);
    }
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    const double e = per_launch_us([&](int n) { for (int i = 0; i < n / 1000 + 1; ++i) hipGraphLaunch(ge, st); }, N) * N /
                     ((N / 1000 + 1) * 1000.0);
    printf("{\"graph_replay_alternating_pair_us\": %.3f}\n", 2 * e);
    printf("{\"empty_1x256_us\": %.3f, \"empty_256x1024_us\": %.3f, \"alternating_empty_pair_us\": %.3f, "
           "\"alternating_pair_writing_a_line_per_workgroup_us\": %.3f}\n", a, b, 2 * c, 2 * d);
    return 0;
}
