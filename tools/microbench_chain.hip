// Cycles per step of candidate forms of dc_stream_kernel's integrator recurrence (one wave per SIMD, as in the kernel): the step
// x <- Phi x + in_k followed by the reset "x <- init where |x| >= T".  Scratch tool behind DESIGN.md 4.4a.
//   hipcc --offload-arch=gfx950 -O3 -o microbench_chain tools/microbench_chain.hip && ./microbench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define STEP1(IN, OUT) "v_fma_f32 %0, %1, %0, " IN "\n v_fma_f32 %5, -|%0|, %3, %4 clamp\n v_mov_b32 " OUT ", %0\n v_mul_f32_e32 %0, %5, %0\n"
#define GROUP4(I0, I1, I2, I3, RD, WR) RD "s_waitcnt lgkmcnt(1)\n" STEP1(I0, "v48") STEP1(I1, "v49") STEP1(I2, "v50") STEP1(I3, "v51") WR
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "memory"
template <int MODE> __global__ void k(unsigned long long *out, float seed, int iters, float T) {
    float x = seed * 0.001f * threadIdx.x, phi = 0.97f, b = 0.01f, init = 0.0f, t = 0.f, keep = 0.f;
    const float BIG = 1.2676506e30f, TBIG = T * BIG;  // 2^100
    __shared__ float lds[4096];
    lds[threadIdx.x] = seed;
    const unsigned ldsaddr = (unsigned)(size_t)lds + threadIdx.x * 16;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // the step alone: dependent v_fma (VOP3)
            asm volatile(REP16("v_fma_f32 %0, %1, %0, %2\n") : "+v"(x) : "s"(phi), "v"(b));
        } else if (MODE == 1) {  // step + v_cmp (VCC) + s_nop 1 + v_cndmask: what the compiler emits today
            asm volatile(REP16("v_fma_f32 %0, %1, %0, %2\n v_cmp_ge_f32_e64 vcc, |%0|, %3\n s_nop 1\n v_cndmask_b32_e32 %0, %0, %4, vcc\n") : "+v"(x) : "s"(phi), "v"(b), "v"(T), "v"(init) : "vcc");
        } else if (MODE == 2) {  // the same without the s_nop (is the hazard interlocked? timing only)
            asm volatile(REP16("v_fma_f32 %0, %1, %0, %2\n v_cmp_ge_f32_e64 vcc, |%0|, %3\n v_cndmask_b32_e32 %0, %0, %4, vcc\n") : "+v"(x) : "s"(phi), "v"(b), "v"(T), "v"(init) : "vcc");
        } else if (MODE == 3) {  // no SGPR in the chain: keep = sat((T - |x|) 2^100) in {0, 1}; x <- keep x   (init == 0)
            asm volatile(REP16("v_fma_f32 %0, %1, %0, %2\n v_fma_f32 %5, -|%0|, %3, %4 clamp\n v_mul_f32_e32 %0, %5, %0\n") : "+v"(x) : "s"(phi), "v"(b), "v"(BIG), "v"(TBIG), "v"(keep));
        } else if (MODE == 4) {  // v_cmpx + v_mov under the new exec + restore
            asm volatile(REP16("v_fma_f32 %0, %1, %0, %2\n v_cmpx_ge_f32_e64 s[22:23], |%0|, %3\n v_mov_b32_e32 %0, %4\n s_mov_b64 exec, -1\n") : "+v"(x) : "s"(phi), "v"(b), "v"(T), "v"(init) : "exec", "s22", "s23");
        } else if (MODE == 5) {  // round 2's step: two FMAs (Euler), normalise, compare, select
            asm volatile(REP16("v_fma_f32 %5, %1, %0, %2\n v_fmac_f32_e32 %0, %1, %5\n v_mul_f32_e32 %5, %1, %0\n v_cmp_gt_f32_e64 vcc, |%5|, %3\n s_nop 1\n v_cndmask_b32_e32 %0, %0, %4, vcc\n") : "+v"(x) : "s"(phi), "v"(b), "v"(T), "v"(init), "v"(t) : "vcc");
        } else if (MODE == 6) {  // MODE 3 with the keep computation as v_med3-free min/max form: x <- x * (|x| < T) via v_cmp_class-free sub+clamp, then mul; plus an LDS write/read per 4 steps
            asm volatile(REP4("v_fma_f32 %0, %1, %0, %2\n v_fma_f32 %5, -|%0|, %3, %4 clamp\n v_mul_f32_e32 %0, %5, %0\n v_fma_f32 %0, %1, %0, %2\n v_fma_f32 %5, -|%0|, %3, %4 clamp\n v_mul_f32_e32 %0, %5, %0\n v_fma_f32 %0, %1, %0, %2\n v_fma_f32 %5, -|%0|, %3, %4 clamp\n v_mul_f32_e32 %0, %5, %0\n v_fma_f32 %0, %1, %0, %2\n v_fma_f32 %5, -|%0|, %3, %4 clamp\n v_mul_f32_e32 %0, %5, %0\n s_nop 0\n s_nop 0\n") : "+v"(x) : "s"(phi), "v"(b), "v"(BIG), "v"(TBIG), "v"(keep));
        } else if (MODE == 7) {  // compare into an SGPR pair other than VCC, select e64
            asm volatile(REP16("v_fma_f32 %0, %1, %0, %2\n v_cmp_ge_f32_e64 s[20:21], |%0|, %3\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n") : "+v"(x) : "s"(phi), "v"(b), "v"(T), "v"(init) : "s20", "s21");
        } else if (MODE == 9) {  // the kernel's group of four steps: inputs by ds_read_b128 (a group ahead), the four states out by ds_write_b128
            asm volatile(REP4(GROUP4("v44", "v45", "v46", "v47", "ds_read_b128 v[40:43], %6\n", "ds_write_b128 %6, v[48:51] offset:8192\n")
                              GROUP4("v40", "v41", "v42", "v43", "ds_read_b128 v[44:47], %6 offset:1024\n", "ds_write_b128 %6, v[48:51] offset:9216\n"))
                         : "+v"(x) : "s"(phi), "v"(b), "v"(BIG), "v"(TBIG), "v"(keep), "v"(ldsaddr) : CLOB);
        } else if (MODE == 10) {  // a checkpoint only: ds_write_b32
            asm volatile(REP4(GROUP4("v44", "v45", "v46", "v47", "ds_read_b128 v[40:43], %6\n", "ds_write_b32 %6, v51 offset:8192\n")
                              GROUP4("v40", "v41", "v42", "v43", "ds_read_b128 v[44:47], %6 offset:1024\n", "ds_write_b32 %6, v51 offset:9216\n"))
                         : "+v"(x) : "s"(phi), "v"(b), "v"(BIG), "v"(TBIG), "v"(keep), "v"(ldsaddr) : CLOB);
        } else if (MODE == 11) {  // no hand-off
            asm volatile(REP4(GROUP4("v44", "v45", "v46", "v47", "ds_read_b128 v[40:43], %6\n", "")
                              GROUP4("v40", "v41", "v42", "v43", "ds_read_b128 v[44:47], %6 offset:1024\n", ""))
                         : "+v"(x) : "s"(phi), "v"(b), "v"(BIG), "v"(TBIG), "v"(keep), "v"(ldsaddr) : CLOB);
        } else if (MODE == 8) {  // v_med3 clamp instead of a reset (NOT the same function: timing reference for a VALU-only 2-instruction step)
            asm volatile(REP16("v_fma_f32 %0, %1, %0, %2\n v_med3_f32 %0, %0, %3, %4\n") : "+v"(x) : "s"(phi), "v"(b), "v"(T), "v"(init));
        }
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (x + t + keep == 12345.678f) out[0] = 0;
}

template <int MODE> void run(const char *name, unsigned long long *dev, int steps_per_asm) {
    const int iters = 400, grid = 64;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, dev, 1.0f, iters, 2.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hst(grid);
    hipMemcpy(hst.data(), dev, grid * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : hst) s += (double)v;
    printf("%-92s %.1f cycles/step\n", name, s / grid / ((double)iters * steps_per_asm));
}

int main() {
    unsigned long long *dev;
    hipMalloc(&dev, 4096);
    run<0>("step alone: dependent v_fma_f32 (VOP3, SGPR operand)", dev, 16);
    run<1>("step + v_cmp_ge -> vcc + s_nop 1 + v_cndmask_b32 (the compiler's code today)", dev, 16);
    run<2>("the same without the s_nop (timing only)", dev, 16);
    run<7>("compare into s[20:21] + s_nop 1 + v_cndmask_b32_e64", dev, 16);
    run<3>("no SGPR on the chain: keep = sat((T - |x|) 2^100), x <- keep * x  (init == 0)", dev, 16);
    run<6>("the same, 4 steps + 2 filler slots per group", dev, 16);
    run<4>("v_cmpx_ge + v_mov_b32 + s_mov_b64 exec, -1", dev, 16);
    run<5>("round 2: Euler's two FMAs + normalise + v_cmp_gt + s_nop 1 + v_cndmask", dev, 16);
    run<8>("reference: v_fma + v_med3 (2 VALU, no reset semantics)", dev, 16);
    run<9>("keep form in groups of four: ds_read_b128 in, ds_write_b128 out, one s_waitcnt", dev, 32);
    run<10>("the same with a ds_write_b32 checkpoint instead of the four states", dev, 32);
    run<11>("the same without any hand-off", dev, 32);
    return 0;
}
