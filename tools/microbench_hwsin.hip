// Accuracy of v_sin_f32 / v_cos_f32 (input in revolutions) on gfx950 for fixed-point angles, against the polynomial sincos of Angle<float>.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const int *ang, float *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = (float)ang[i] * 2.3283064365386963e-10f;  // 2^-32 revolutions
    out[2 * i] = __builtin_amdgcn_sinf(x);
    out[2 * i + 1] = __builtin_amdgcn_cosf(x);
    // variant B: quadrant in integer arithmetic first
    unsigned ua = (unsigned)ang[i] + 0x20000000u, q = ua >> 30;
    int r = (int)(ua & 0x3FFFFFFFu) - 0x20000000;
    float xr = (float)r * 2.3283064365386963e-10f;
    float sp = __builtin_amdgcn_sinf(xr), cp = __builtin_amdgcn_cosf(xr);
    float s1 = (q & 1u) ? cp : sp, c1 = (q & 1u) ? sp : cp;
    out[2 * n + 2 * i] = (q & 2u) ? -s1 : s1;
    out[2 * n + 2 * i + 1] = ((q + 1u) & 2u) ? -c1 : c1;
    // variant C: A plus a first-order correction with the counts the float conversion dropped
    {
        const float xf = (float)ang[i];
        const int lo = ang[i] - (int)xf;  // |lo| <= 128 (exact: xf is an integer-valued float within int range except at +2^31)
        const float d = (float)lo * 1.4629180792671596e-9f;
        const float sa = out[2 * i], ca = out[2 * i + 1];
        out[4 * n + 2 * i] = fmaf(d, ca, sa);
        out[4 * n + 2 * i + 1] = fmaf(-d, sa, ca);
    }
    // variant D: the minimax polynomials used so far
    {
        float x = (float)r * 1.4629180792671596e-9f, z = x * x;
        float spp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, x, x);
        float cpp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.0f));
        float s2 = (q & 1u) ? cpp : spp, c2 = (q & 1u) ? spp : cpp;
        out[6 * n + 2 * i] = (q & 2u) ? -s2 : s2;
        out[6 * n + 2 * i + 1] = ((q + 1u) & 2u) ? -c2 : c2;
    }
}
int main() {
    const int n = 1 << 22;
    std::vector<int> h(n);
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)s; }
    for (int i = 0; i < 4096; ++i) h[i] = (i - 2048) * 1000;  // near zero
    for (int i = 4096; i < 8192; ++i) h[i] = (int)0x40000000 + (i - 6144) * 1000;  // near pi/2
    int *d; float *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, (size_t)n * 32);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, n);
    std::vector<float> r((size_t)8 * n);
    hipMemcpy(r.data(), o, (size_t)n * 32, hipMemcpyDeviceToHost);
    for (int v = 0; v < 4; ++v) {
        double es = 0, ec = 0, en = 0, rs = 0, rn = 0;
        for (int i = 0; i < n; ++i) {
            double a = (double)h[i] * 1.4629180792671596e-9;
            double ss = r[(size_t)2 * n * v + 2 * i], cc = r[(size_t)2 * n * v + 2 * i + 1];
            es = fmax(es, fabs(ss - sin(a))); ec = fmax(ec, fabs(cc - cos(a))); en = fmax(en, fabs(ss * ss + cc * cc - 1.0)); rs += (ss - sin(a)) * (ss - sin(a)) + (cc - cos(a)) * (cc - cos(a)); rn += (ss * ss + cc * cc - 1.0);
        }
        printf("variant %c: max |sin err| %.3e  max |cos err| %.3e  max |s^2+c^2-1| %.3e  rms err %.3e  mean(s^2+c^2-1) %.3e\n", 'A' + v, es, ec, en, sqrt(rs / (2.0 * n)), rn / n);
    }
    return 0;
}
