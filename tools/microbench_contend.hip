// How much does activity on the OTHER three SIMDs of a CU slow down one wave's VALU stream?  (gfx950 scratch tool)
// Wave 0 of each 256-thread workgroup runs a dependent chain of 8-byte VALU instructions and times itself; waves 1..3 run OTHER.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int OTHER> __global__ void k(unsigned long long *out, float *sink, float seed, int iters) {
    extern __shared__ float lds[];
    float a = seed + threadIdx.x, m = 1.0001f, q = 0.5f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        unsigned long long t0 = clock64();
        for (int it = 0; it < iters; ++it) asm volatile(REP64("v_fma_f32 %0, %1, %0, %2\n") : "+v"(a) : "v"(m), "v"(q));
        unsigned long long t1 = clock64();
        if (lane == 0) out[blockIdx.x] = t1 - t0;
    } else {
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 w = {a, a, a, a};
        float *p32 = lds + wave * 4096 + lane;            // conflict-free dword
        float *p128 = lds + wave * 4096 + lane * 12;      // 48-byte lane stride: 4-way conflicts on b128
        for (int it = 0; it < iters * 2; ++it) {
            if (OTHER == 1) asm volatile(REP64("v_fma_f32 %0, %1, %0, %2\n") : "+v"(a) : "v"(m), "v"(q));
            else if (OTHER == 2) { asm volatile(REP16("ds_write_b32 %0, %1\n") :: "v"((unsigned)(size_t)p32 & 0xffffu), "v"(a) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            else if (OTHER == 3) { asm volatile(REP16("ds_write_b128 %0, %1\n") :: "v"((unsigned)(size_t)p128 & 0xffffu), "v"(w) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            else if (OTHER == 4) { v4 r; asm volatile(REP16("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r) : "v"((unsigned)(size_t)p128 & 0xffffu) : "memory"); a += r.x; }
            else if (OTHER == 5) { for (int j = 0; j < 4; ++j) sink[(size_t)(blockIdx.x * 256 + threadIdx.x) + (size_t)((it * 4 + j) & 1023) * 65536] = a; }
            else if (OTHER == 6) { unsigned s = it; asm volatile(REP64("s_add_u32 %0, %0, 1\n") : "+s"(s)); a += s; }
            else if (OTHER == 7) {  // a different VALU code stream mixed with LDS traffic, like the output waves
                asm volatile(REP16("v_fma_f32 %0, %1, %0, %2\n v_mul_f32 %0, %1, %0\n ds_write_b32 %3, %0\n v_fma_f32 %0, %1, %0, %2\n") : "+v"(a) : "v"(m), "v"(q), "v"((unsigned)(size_t)p32 & 0xffffu) : "memory");
            }
        }
        if (a == 12345.678f) sink[0] = a;
    }
}
template <int OTHER> void run(const char *name, unsigned long long *dev, float *sink) {
    const int iters = 200, grid = 256;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<OTHER>, dim3(grid), dim3(256), 65536, 0, dev, sink, 1.0f, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> hst(grid);
    (void)hipMemcpy(hst.data(), dev, grid * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : hst) s += (double)v;
    printf("others: %-44s wave0 %.2f cycles/instr\n", name, s / grid / (iters * 64.0));
}
int main() {
    unsigned long long *dev; float *sink;
    (void)hipMalloc(&dev, 256 * 8);
    (void)hipMalloc(&sink, (size_t)1024 * 65536 * 4 + 65536 * 4);
    run<0>("idle (exit at once)", dev, sink);
    run<1>("same VALU chain (own code copy)", dev, sink);
    run<2>("ds_write_b32 conflict-free", dev, sink);
    run<3>("ds_write_b128, 48-B lane stride", dev, sink);
    run<4>("ds_read_b128, 48-B lane stride", dev, sink);
    run<5>("global stores (dword, coalesced)", dev, sink);
    run<6>("SALU chain", dev, sink);
    run<7>("VALU + ds_write_b32 mix", dev, sink);
    return 0;
}
