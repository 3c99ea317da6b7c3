// Store issue rate of one CU on gfx950: each wave writes 1-KiB contiguous spans (64 lanes x 16 B) back to back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(float4 *out, unsigned long long *t, int iters, size_t stride_vec) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    float4 v = {1.f * lane, 2.f, 3.f, 4.f};
    float4 *p = out + ((size_t)blockIdx.x * nw + wave) * 64 + lane;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) p[(size_t)(it * 16 + j) * stride_vec] = v;
    }
    unsigned long long t1 = clock64();
    if (lane == 0) t[blockIdx.x * nw + wave] = t1 - t0;
}
int main() {
    const size_t stride_vec = 256 * 8 * 64;  // one "row" of all workgroups' spans: 2 MiB
    const int iters = 64;
    float4 *out; unsigned long long *t;
    hipMalloc(&out, stride_vec * 16 * iters * 16 + (1 << 20));
    hipMalloc(&t, 256 * 8 * 8);
    for (int grid : {16, 64, 256}) for (int nw : {1, 2, 3, 6, 8}) {
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), 0, 0, out, t, iters, stride_vec);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid * nw);
        hipMemcpy(h.data(), t, grid * nw * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto x : h) s += (double)x;
        double cyc = s / (grid * nw) / (iters * 16.0);
        printf("grid=%3d waves/WG=%d: %.1f cycles per 1-KiB store per wave -> %.1f B/cycle per CU, %.2f TB/s chip-wide at 2.3 GHz\n", grid, nw, cyc,
               1024.0 * nw / cyc, 1024.0 * nw / cyc * grid * 2.3e9 / 1e12);
    }
    return 0;
}
