// Diagnostic (not product): what one kernel launch costs the HOST, by API -- the closed loop enqueues two per iteration and
// its GPU side runs 4.4 us per kernel, so the host's ~2.8 us per launch leave little slack.  Enqueue-only timing of bursts
// of 400 launches into an idle stream (below the queue depth, so nothing blocks), a 600-byte by-value argument like KParams.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_cost.hip -o tools/_bin/launch_cost && tools/_bin/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct Fat { int v[150]; };
__global__ void k_fat(const int *p, Fat f, float *out) {
    if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = (float)f.v[7];
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    Fat f{};
    const int *p = nullptr;
    float *o = nullptr;
    const int N = 400, REP = 20;
    auto burst = [&](auto &&one) {
        double best = 1e30;
        for (int r = 0; r < REP; ++r) {
            (void)hipStreamSynchronize(s);
            const double t0 = now_us();
            for (int i = 0; i < N; ++i) one();
            const double dt = (now_us() - t0) / N;
            best = dt < best ? dt : best;
        }
        (void)hipStreamSynchronize(s);
        return best;
    };
    const double a = burst([&] { hipLaunchKernelGGL(k_fat, dim3(256), dim3(1024), 0, s, p, f, o); });
    void *args[3] = {(void *)&p, (void *)&f, (void *)&o};
    const double b = burst([&] { (void)hipLaunchKernel((const void *)k_fat, dim3(256), dim3(1024), args, 0, s); });
    hipFunction_t fn = nullptr;
    double c = -1.0;
    if (hipGetFuncBySymbol(&fn, (const void *)k_fat) == hipSuccess && fn) {
        struct alignas(8) Packed { const int *p; Fat f; float *o; } pk{p, f, o};
        size_t sz = sizeof(pk);
        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pk, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        c = burst([&] { (void)hipModuleLaunchKernel(fn, 256, 1, 1, 1024, 1, 1, 0, s, nullptr, extra); });
    }
    printf("{\"hipLaunchKernelGGL_us\": %.3f, \"hipLaunchKernel_us\": %.3f, \"hipModuleLaunchKernel_extra_us\": %.3f, \"last_error\": \"%s\"}\n",
           a, b, c, hipGetErrorString(hipGetLastError()));
    return 0;
}
