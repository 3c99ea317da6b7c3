// Host cost of one kernel launch with a ~700-byte by-value argument (KArgs is that size), back to back on one stream:
//   (a) hipLaunchKernelGGL (what gemx_step does), (b) hipModuleLaunchKernel on a hipFunction_t resolved once, arguments as ONE buffer
//   (HIP_LAUNCH_PARAM_BUFFER_POINTER), (c) the same launches captured once into a graph and replayed.
// hipcc --offload-arch=gfx950 -O2 tools/microbench_launch.hip -o tools/microbench_launch && tools/microbench_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float p[170]; float *out; int n; };
__global__ void k(const Big a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.out[0] = a.p[3] + a.n; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    Big a{};
    hipMalloc(&a.out, 4);
    a.n = 1;
    hipStream_t st;
    hipStreamCreate(&st);
    const int N = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipStreamSynchronize(st);
        double t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, st, a);
        double t1 = now();
        hipStreamSynchronize(st);
        double t2 = now();
        printf("hipLaunchKernelGGL: %.2f us per launch to enqueue, %.2f us per launch until done\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
    }
    hipFunction_t f;
    if (hipGetFuncBySymbol(&f, (const void *)k) != hipSuccess) { printf("hipGetFuncBySymbol failed\n"); return 1; }
    for (int rep = 0; rep < 2; ++rep) {
        hipStreamSynchronize(st);
        double t0 = now();
        for (int i = 0; i < N; ++i) {
            size_t sz = sizeof(Big);
            void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            hipModuleLaunchKernel(f, 256, 1, 1, 64, 1, 1, 0, st, nullptr, cfg);
        }
        double t1 = now();
        hipStreamSynchronize(st);
        double t2 = now();
        printf("hipModuleLaunchKernel (buffer): %.2f us per launch to enqueue, %.2f us per launch until done\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
    }
    return 0;
}
