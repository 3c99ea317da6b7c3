// The rollout's WRITE PATTERN without any compute (round 4): workgroup w writes, for k = 0 .. K-1, its contiguous 64-env span of observation
// row k -- ROWB = 64 * S_out * 4 bytes at  base + k * (N * S_out * 4) + w * ROWB  -- with non-temporal 16-byte stores, exactly the addresses
// advance_pipe_kernel's output waves produce for obs[K][N][S_out].  Question: is the drop from 0.83 of the HBM peak at 16384 envs to ~0.65 at
// 32768 ... 131072 envs (whatever the kernel's shape: profiles/r04k_d6_ab.txt) the write path's answer to the PATTERN (row pitch, number
// of workgroups, rounds), or the kernel's doing?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_rowpitch.hip -o tools/microbench_rowpitch && tools/microbench_rowpitch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rows(vf4 *out, int K, long long pitch_v, int rowv, int D, int pace_cycles) {
    extern __shared__ unsigned char pad[];  // (sets the residency: LDS bytes per workgroup)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const vf4 v = {1.f * lane, 2.f, 3.f, (float)blockIdx.x};
    // like the kernel: blocks of D steps, wave `wave` of WAVES takes rows wave, wave + WAVES, ... of the block
    // pace_cycles > 0: the workgroup starts block k0 / D no earlier than (k0 / D) * D * pace_cycles shader cycles after its start -- a
    // rate limit of one row per pace_cycles per workgroup, the way the real kernel's integrator paces its output waves at 16384 envs
    const long long t_start = clock64();
    for (int k0 = 0; k0 < K; k0 += D) {
        if (pace_cycles > 0) {
            const long long due = t_start + (long long)k0 * pace_cycles;
            while ((long long)clock64() < due) __builtin_amdgcn_s_sleep(2);
        }
        for (int s = wave; s < D && k0 + s < K; s += WAVES) {
            vf4 *row = out + (long long)(k0 + s) * pitch_v + (long long)blockIdx.x * rowv;
            for (int j = lane; j < rowv; j += 64) __builtin_nontemporal_store(v, row + j);
        }
        __syncthreads();
    }
    if (pad[0] == 77 && out == nullptr) out[0] = v;
}

int main() {
    const int S_OUT = 14, K = 1000;
    const int rowv = 64 * S_OUT * 4 / 16;  // 224 sixteen-byte units per 64-env row span
    size_t maxbytes = (size_t)131072 * S_OUT * 4 * K + (1 << 20);
    vf4 *out;
    if (hipMalloc(&out, maxbytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("| envs | workgroups | waves/WG | D | LDS KiB/WG (residency) | pace (cycles per row and WG) | us per launch | GB/s | of 8 TB/s |\n|---|---|---|---|---|---|---|---|---|\n");
    for (int N : {16384, 32768, 65536, 131072}) {
        const long long pitch_v = (long long)N * S_OUT * 4 / 16;
        const int grid = N / 64;
        struct Cfg { int waves, D, lds_kib, pace; };
        // pace: shader cycles per row and workgroup; demand = 3584 B x resident workgroups / pace.  One workgroup per CU (100 KiB LDS) at
        // 2.3 GHz: 300 cycles = 7.0 TB/s, 330 = 6.4, 360 = 5.9; four per CU (36 KiB): x 4
        for (Cfg c : {Cfg{3, 12, 100, 0}, Cfg{3, 12, 100, 300}, Cfg{3, 12, 100, 320}, Cfg{3, 12, 100, 340}, Cfg{3, 12, 100, 360}, Cfg{3, 12, 100, 400},
                      Cfg{2, 4, 36, 0}, Cfg{2, 4, 36, 1200}, Cfg{2, 4, 36, 1280}, Cfg{2, 4, 36, 1360}, Cfg{2, 4, 36, 1440}, Cfg{2, 4, 36, 1600}}) {
            auto launch = [&]() {
                const size_t lds = (size_t)c.lds_kib * 1024;
                if (c.waves == 2) { hipFuncSetAttribute((const void *)rows<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(rows<2>, dim3(grid), dim3(128), lds, 0, out, K, pitch_v, rowv, c.D, c.pace); }
                else if (c.waves == 3) { hipFuncSetAttribute((const void *)rows<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(rows<3>, dim3(grid), dim3(192), lds, 0, out, K, pitch_v, rowv, c.D, c.pace); }
                else if (c.waves == 4) { hipFuncSetAttribute((const void *)rows<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(rows<4>, dim3(grid), dim3(256), lds, 0, out, K, pitch_v, rowv, c.D, c.pace); }
                else { hipFuncSetAttribute((const void *)rows<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(rows<6>, dim3(grid), dim3(384), lds, 0, out, K, pitch_v, rowv, c.D, c.pace); }
            };
            for (int r = 0; r < 20; ++r) launch();
            hipDeviceSynchronize();
            std::vector<float> ts;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                for (int r = 0; r < 10; ++r) launch();
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                ts.push_back(ms / 10);
            }
            std::sort(ts.begin(), ts.end());
            const double bytes = (double)N * S_OUT * 4 * K, gbs = bytes / (ts[1] * 1e-3) / 1e9;
            printf("| %d | %d | %d | %d | %d | %d | %.1f | %.0f | %.3f |\n", N, grid, c.waves, c.D, c.lds_kib, c.pace, ts[1] * 1e3, gbs, gbs / 8000.0);
            fflush(stdout);
        }
    }
    return 0;
}
