// Single-wave VALU issue-rate microbenchmark for gfx950 (scratch tool behind DESIGN.md's latency notes).
// Each kernel runs REP x 64 instructions of one pattern between two s_memtime reads; one wave per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int MODE> __global__ void k(unsigned long long *out, float seed, int iters) {
    float a = seed + threadIdx.x, b = seed * 2, c = seed * 3, d = seed * 4, e = seed * 5, f = seed * 6, g = seed * 7, h = seed * 8;
    float m = 1.0001f, q = 0.5f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 pa = {a, b}, pb = {c, d}, pc = {e, f}, pd = {g, h}, pm = {m, m}, pq = {q, q};
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // dependent v_fmac (VOP2, 4 bytes)
            asm volatile(REP64("v_fmac_f32 %0, %1, %0\n") : "+v"(a) : "v"(m));
        } else if (MODE == 1) {  // two independent chains
            asm volatile(REP16("v_fmac_f32 %0, %2, %0\n v_fmac_f32 %1, %2, %1\n v_fmac_f32 %0, %2, %0\n v_fmac_f32 %1, %2, %1\n") : "+v"(a), "+v"(b) : "v"(m));
        } else if (MODE == 2) {  // four independent chains
            asm volatile(REP16("v_fmac_f32 %0, %4, %0\n v_fmac_f32 %1, %4, %1\n v_fmac_f32 %2, %4, %2\n v_fmac_f32 %3, %4, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
        } else if (MODE == 3) {  // dependent v_fma (VOP3, 8 bytes)
            asm volatile(REP64("v_fma_f32 %0, %1, %0, %2\n") : "+v"(a) : "v"(m), "v"(q));
        } else if (MODE == 4) {  // four independent v_fma (VOP3)
            asm volatile(REP16("v_fma_f32 %0, %4, %0, %5\n v_fma_f32 %1, %4, %1, %5\n v_fma_f32 %2, %4, %2, %5\n v_fma_f32 %3, %4, %3, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
        } else if (MODE == 5) {  // dependent v_pk_fma_f32
            asm volatile(REP64("v_pk_fma_f32 %0, %1, %0, %2\n") : "+v"(pa) : "v"(pm), "v"(pq));
        } else if (MODE == 6) {  // four independent v_pk_fma_f32
            asm volatile(REP16("v_pk_fma_f32 %0, %4, %0, %5\n v_pk_fma_f32 %1, %4, %1, %5\n v_pk_fma_f32 %2, %4, %2, %5\n v_pk_fma_f32 %3, %4, %3, %5\n") : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pm), "v"(pq));
        } else if (MODE == 7) {  // dependent v_mul with SGPR operand
            asm volatile(REP64("v_mul_f32 %0, %1, %0\n") : "+v"(a) : "s"(m));
        } else if (MODE == 8) {  // dependent pairs: cmp -> cndmask
            asm volatile(REP16("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a) : "v"(m) : "vcc");
        } else if (MODE == 9) {  // v_fmaak (literal constant, 8 bytes) dependent
            asm volatile(REP64("v_fmaak_f32 %0, %1, %0, 0x3f000000\n") : "+v"(a) : "v"(m));
        } else if (MODE == 10) {  // dependent chain alternating pk_fma -> fma reading one half
            asm volatile(REP16("v_pk_fma_f32 %0, %1, %0, %2\n v_pk_mul_f32 %0, %1, %0\n v_pk_fma_f32 %0, %1, %0, %2\n v_pk_add_f32 %0, %1, %0\n") : "+v"(pa) : "v"(pm), "v"(pq));
        } else if (MODE == 11) {  // two independent pk chains
            asm volatile(REP16("v_pk_fma_f32 %0, %2, %0, %3\n v_pk_fma_f32 %1, %2, %1, %3\n v_pk_fma_f32 %0, %2, %0, %3\n v_pk_fma_f32 %1, %2, %1, %3\n") : "+v"(pa), "+v"(pb) : "v"(pm), "v"(pq));
        } else if (MODE == 12) {  // pk_fma interleaved with an independent scalar-fp32 chain
            asm volatile(REP16("v_pk_fma_f32 %0, %2, %0, %3\n v_fmac_f32 %1, %4, %1\n v_pk_fma_f32 %0, %2, %0, %3\n v_fmac_f32 %1, %4, %1\n") : "+v"(pa), "+v"(a) : "v"(pm), "v"(pq), "v"(m));
        }
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a + b + c + d + pa.x + pa.y + pb.x + pc.x + pd.x == 12345.678f) out[0] = 0;
}

template <int MODE> void run(const char *name, unsigned long long *dev, int threads) {
    const int iters = 200, grid = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, dev, 1.0f, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, dev, 1.0f, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hst(grid);
    hipMemcpy(hst.data(), dev, grid * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : hst) s += (double)v;
    printf("%-44s threads=%3d  %.2f cycles/instr\n", name, threads, s / grid / (iters * 64.0));
}

int main() {
    unsigned long long *dev;
    hipMalloc(&dev, 256 * 16 * 8);
    for (int threads : {64, 256, 512}) {
        run<0>("dependent v_fmac (VOP2)", dev, threads);
        run<1>("2 independent v_fmac chains", dev, threads);
        run<2>("4 independent v_fmac chains", dev, threads);
        run<3>("dependent v_fma (VOP3)", dev, threads);
        run<4>("4 independent v_fma (VOP3)", dev, threads);
        run<5>("dependent v_pk_fma_f32", dev, threads);
        run<11>("2 independent v_pk_fma_f32", dev, threads);
        run<6>("4 independent v_pk_fma_f32", dev, threads);
        run<7>("dependent v_mul with SGPR", dev, threads);
        run<8>("v_cmp -> v_cndmask chain", dev, threads);
        run<9>("dependent v_fmaak (literal)", dev, threads);
        run<10>("dependent pk fma/mul/fma/add", dev, threads);
        run<12>("pk_fma chain + independent fmac chain", dev, threads);
    }
    return 0;
}
