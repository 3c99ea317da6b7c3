// Does the LAYOUT of the action tensor matter once it no longer fits the 256 MB Infinity Cache?  (Round 5: the continuous-action rows at 65536
// envs run 0.79-0.86 of the roofline while the action tensor is cache resident -- K = 250 -- and 0.64-0.69 at K = 1000, 786 MB.)
// The rollout's memory traffic without any compute, 65536 envs = 1024 workgroups of four waves, D = 2 steps per barrier, like BASELINE config 4:
//   two waves write their row of the block (3584 B, non-temporal 16-byte stores at the [K][N][14] addresses);
//   one wave reads the block's actions (A = 12 B per env and step) in one of four ways:
//     0  none
//     1  the reference's [K][N][A]: 768 B per step at a stride of N * 12 B
//     2  group-major [N/64][K][64][A]: the workgroup's own stream, 768 B per step, contiguous from step to step
//     3  group-major, 16 steps (12 KB) at once every eighth block
//     4  (round 6) the reference layout with HALF-precision values: 384 B per step -- what a narrow action tensor (gemx_rollout_half) could buy at best
// pace_ns > 0: block b starts no earlier than b * pace_ns after the workgroup's start (the rate limiter).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_readmix.hip -o tools/microbench_readmix && tools/microbench_readmix
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long wall100() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz

__global__ __launch_bounds__(256) void mix(vf4 *out, const vf4 *act, vf4 *sink, int K, int N, int mode, unsigned pace_ticks) {
    __shared__ vf4 lds[16 * 48];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, w = blockIdx.x;
    const long long pitch_v = (long long)N * 14 * 4 / 16;  // observation row pitch in 16-byte units
    const int rowv = 224;                                  // 64 envs x 14 x 4 B
    const long long act_row_v = (long long)N * 12 / 16;    // [K][N][A]: one step's row in 16-byte units
    const vf4 v = {1.f * lane, 2.f, 3.f, (float)w};
    vf4 acc = {0.f, 0.f, 0.f, 0.f};
    vf4 q[4][2] = {};
    const unsigned long long t0 = wall100();
    #pragma unroll 4
    for (int k0 = 0, b = 0; k0 < K; k0 += 2, ++b) {
        if (wave == 0 && pace_ticks) {
            const unsigned long long due = t0 + (unsigned long long)b * pace_ticks;
            while (wall100() < due) __builtin_amdgcn_s_sleep(4);
        }
        if (wave == 1 || wave == 2) {
            const int s = wave - 1;
            if (k0 + s < K) {
                vf4 *row = out + (long long)(k0 + s) * pitch_v + (long long)w * rowv;
                for (int j = lane; j < rowv; j += 64) __builtin_nontemporal_store(v, row + j);
            }
        } else if (wave == 3) {
            // four blocks ahead through a register queue: the loads' latency is never exposed (the real kernel's loader stages one block
            // ahead into LDS and has the integrator's block to hide behind)
            const int kb = k0 + 8;  // the block fetched now
            vf4 f0 = {0.f, 0.f, 0.f, 0.f}, f1 = f0;
            if (mode == 1) {
                if (lane < 48 && kb < K) f0 = act[(long long)kb * act_row_v + (long long)w * 48 + lane];
                if (lane < 48 && kb + 1 < K) f1 = act[(long long)(kb + 1) * act_row_v + (long long)w * 48 + lane];
            } else if (mode == 2) {
                if (lane < 48 && kb < K) f0 = act[((long long)w * K + kb) * 48 + lane];
                if (lane < 48 && kb + 1 < K) f1 = act[((long long)w * K + kb + 1) * 48 + lane];
            } else if (mode == 5) {  // round 6: mode 1 with NON-TEMPORAL loads (the action stream is read once)
                if (lane < 48 && kb < K) f0 = __builtin_nontemporal_load(&act[(long long)kb * act_row_v + (long long)w * 48 + lane]);
                if (lane < 48 && kb + 1 < K) f1 = __builtin_nontemporal_load(&act[(long long)(kb + 1) * act_row_v + (long long)w * 48 + lane]);
            } else if (mode == 6) {  // round 6: mode 1, but BOTH steps' rows fetched by one instruction's 64 lanes where they fit (2 x 48 = 96 units: two instructions of 48 -> 64 + 32)
                const long long u = (long long)lane;
                if (kb + 1 < K) {
                    const long long r0 = (long long)kb * act_row_v + (long long)w * 48, r1 = (long long)(kb + 1) * act_row_v + (long long)w * 48;
                    f0 = act[u < 48 ? r0 + u : r1 + (u - 48)];
                    if (lane < 32) f1 = act[r1 + 16 + lane];
                }
            } else if (mode == 4) {  // round 6: a NARROW action tensor, [K][N][A] halves -- 384 B per workgroup and step (24 sixteen-byte units)
                const long long hrow_v = (long long)N * 6 / 16;
                if (lane < 24 && kb < K) f0 = act[(long long)kb * hrow_v + (long long)w * 24 + lane];
                if (lane < 24 && kb + 1 < K) f1 = act[(long long)(kb + 1) * hrow_v + (long long)w * 24 + lane];
            } else if (mode == 3) {
                if ((b & 7) == 0)
                    for (int j = lane; j < 16 * 48 && kb + j / 48 < K; j += 64) f0 += act[((long long)w * K + kb) * 48 + j];
            }
            acc += q[b & 3][0] + q[b & 3][1];
            q[b & 3][0] = f0; q[b & 3][1] = f1;
            lds[lane] = acc;
        }
        __syncthreads();
    }
    if (wave == 3 && acc.x == 12345.f) sink[w] = acc;
}

int main() {
    const int N = 65536, K = 1000;
    vf4 *out, *act, *sink;
    const size_t ob = (size_t)N * 14 * 4 * K, ab = (size_t)N * 12 * K;
    if (hipMalloc(&out, ob) != hipSuccess || hipMalloc(&act, ab) != hipSuccess || hipMalloc(&sink, 1 << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(act, 0, ab);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("| action reads | pace (ns per 2-step block) | us per launch | bytes moved | GB/s | of 8 TB/s |\n|---|---|---|---|---|---|\n");
    const char *names[7] = {"none", "[K][N][A] (reference)", "group-major, per step", "group-major, 12 KB every 8 blocks", "[K][N][A] as HALVES (6 B per env-step)",
                            "[K][N][A], non-temporal loads", "[K][N][A], two rows per 64-lane instruction"};
    for (unsigned pace : {0u, 150u, 162u, 175u}) {
        for (int mode = 0; mode < 7; ++mode) {
            auto launch = [&]() { hipLaunchKernelGGL(mix, dim3(N / 64), dim3(256), 0, 0, out, act, sink, K, N, mode, pace); };
            for (int r = 0; r < 10; ++r) launch();
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
            const double bytes = (double)ob + (mode == 4 ? (double)ab / 2 : (mode ? (double)ab : 0.0)), gbs = bytes / (ts[1] * 1e-3) / 1e9;
            printf("| %s | %u | %.1f | %.0f MB | %.0f | %.3f |\n", names[mode], pace * 10, ts[1] * 1e3, bytes / 1e6, gbs, gbs / 8000.0);
            fflush(stdout);
        }
    }
    return 0;
}
