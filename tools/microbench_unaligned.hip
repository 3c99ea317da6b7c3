// Do 16-byte vector accesses work at addresses that are only dword- (or byte-) aligned on gfx950 under ROCm's default alignment mode?
//   global_store_dwordx4 at base + 4 / + 8 / + 12 (observation rows of a batch whose n_envs * n_out is not a multiple of 4),
//   a 16-byte store at a BYTE offset (done rows of a batch whose n_envs is not a multiple of 16),
//   global_load_lds_dwordx4 from a byte-misaligned global address (action staging of such a batch).
// Each case in its own launch, results checked on the host; run under `timeout`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ void st_f4(float *out, int off_dwords) {
    v4f v = {(float)(4 * threadIdx.x), (float)(4 * threadIdx.x + 1), (float)(4 * threadIdx.x + 2), (float)(4 * threadIdx.x + 3)};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(out + off_dwords) + threadIdx.x);
}
__global__ void st_b16(unsigned char *out, int off_bytes) {
    v4u v = {threadIdx.x * 4u, threadIdx.x * 4u + 1, threadIdx.x * 4u + 2, threadIdx.x * 4u + 3};
    if (threadIdx.x < 4) __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(out + off_bytes) + threadIdx.x);
}
__global__ void ld_lds(const unsigned char *in, int off_bytes, unsigned int *out) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[1024];
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(in + off_bytes + threadIdx.x * 16),
                                     (void __attribute__((address_space(3))) *)buf, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = reinterpret_cast<unsigned int *>(buf)[i];
}
int main() {
    float *d; unsigned char *b, *in; unsigned int *o;
    hipMalloc(&d, 4096); hipMalloc(&b, 4096); hipMalloc(&in, 4096); hipMalloc(&o, 1024);
    int bad = 0;
    for (int off : {0, 1, 2, 3}) {
        hipMemset(d, 0, 4096);
        hipLaunchKernelGGL(st_f4, dim3(1), dim3(64), 0, 0, d, off);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> h(1024);
        hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
        int ok = e == hipSuccess;
        for (int i = 0; i < 256 && ok; ++i) ok = h[off + i] == (float)i;
        printf("global_store_dwordx4 at +%d dwords: %s (%s)\n", off, ok ? "ok" : "WRONG", hipGetErrorString(e));
        bad += !ok;
    }
    for (int off : {0, 4, 1, 2, 7}) {
        hipMemset(b, 0xEE, 4096);
        hipLaunchKernelGGL(st_b16, dim3(1), dim3(64), 0, 0, b, off);
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned char> h(4096);
        hipMemcpy(h.data(), b, 4096, hipMemcpyDeviceToHost);
        int ok = e == hipSuccess;
        for (int i = 0; i < 16 && ok; ++i) { unsigned int v; memcpy(&v, &h[off + 4 * i], 4); ok = v == (unsigned)i; }
        printf("16-byte store at +%d bytes: %s (%s)\n", off, ok ? "ok" : "WRONG", hipGetErrorString(e));
        bad += !ok;
    }
    std::vector<unsigned char> src(4096);
    for (int i = 0; i < 4096; ++i) src[i] = (unsigned char)(i * 7 + 3);
    hipMemcpy(in, src.data(), 4096, hipMemcpyHostToDevice);
    for (int off : {0, 16, 4, 1, 3}) {
        hipMemset(o, 0, 1024);
        hipLaunchKernelGGL(ld_lds, dim3(1), dim3(64), 0, 0, in, off, o);
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned char> h(1024);
        hipMemcpy(h.data(), o, 1024, hipMemcpyDeviceToHost);
        int ok = e == hipSuccess && memcmp(h.data(), src.data() + off, 1024) == 0;
        printf("global_load_lds_dwordx4 from +%d bytes: %s (%s)\n", off, ok ? "ok" : "WRONG", hipGetErrorString(e));
        bad += !ok;
    }
    return bad;
}
