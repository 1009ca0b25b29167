// Semantics probe for v_cvt_scalef32_pk32_fp6_f16 (gfx950): 32 f16 (16 VGPRs) -> 32 e2m3 codes (6 VGPRs); element order, scale, rounding.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
__global__ void k(const float* x, const float* sc, unsigned* o) {
    f16x32 a;
    for (int i = 0; i < 32; ++i) a[i] = (_Float16)x[threadIdx.x * 32 + i];
    u32x6 r;
    const float scv = sc[threadIdx.x];
    asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(r) : "v"(a), "v"(scv));
    for (int i = 0; i < 6; ++i) o[threadIdx.x * 6 + i] = r[i];
}
static float dec(int c) { int s = c >> 5, e = (c >> 3) & 3, m = c & 7; float v = e == 0 ? m / 8.f : (1 + m / 8.f) * (1 << (e - 1)); return s ? -v : v; }
int main() {
    const int L = 64;
    std::vector<float> x(L * 32), sc(L);
    srand(3);
    for (int l = 0; l < L; ++l) {
        sc[l] = l < 8 ? 1.f : ldexpf(1.f + (l % 3) * 0.25f, (l % 9) - 4);
        for (int i = 0; i < 32; ++i) {
            float v = l == 0 ? (i * 0.25f - 4.f) : ((rand() / (float)RAND_MAX) * 16.f - 8.f) * (l < 8 ? 1.f : ldexpf(1.f, (l % 9) - 4));
            x[l * 32 + i] = (float)(_Float16)v;                 // exactly representable in f16
        }
    }
    float *dx, *ds; unsigned* dout;
    (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&ds, L * 4); (void)hipMalloc(&dout, L * 24);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(ds, sc.data(), L * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(L), 0, 0, dx, ds, dout);
    std::vector<unsigned> o(L * 6);
    (void)hipMemcpy(o.data(), dout, L * 24, hipMemcpyDeviceToHost);
    printf("lane 0 decoded (bit order): ");
    for (int j = 0; j < 32; ++j) { int bit = 6 * j, w = bit >> 5, s = bit & 31; unsigned long long v = o[w] >> s; if (s > 26) v |= (unsigned long long)o[w + 1] << (32 - s); printf("%g ", dec(v & 63)); }
    printf("\n");
    int bad = 0;
    for (int l = 0; l < L; ++l) for (int j = 0; j < 32; ++j) {
        int bit = 6 * j, w = bit >> 5, s = bit & 31; unsigned long long v = o[l * 6 + w] >> s; if (s > 26) v |= (unsigned long long)o[l * 6 + w + 1] << (32 - s);
        int ex; frexpf(sc[l], &ex); float p2 = ldexpf(1.f, ex - 1);
        float y = x[l * 32 + j] / p2, ay = fabsf(y), idx;
        if (ay < 2) idx = rintf(ay * 8); else if (ay < 4) idx = 16 + rintf((ay - 2) * 4); else idx = fminf(24 + rintf((ay - 4) * 2), 31);
        int code = (int)idx | (y < 0 || (y == 0 && signbit(y)) ? 32 : 0);
        if ((int)(v & 63) != code) { if (bad < 8) printf("lane %d j %d: hw %d (%g) sw %d (%g) y %g\n", l, j, (int)(v & 63), dec(v & 63), code, dec(code), y); ++bad; }
    }
    printf("sequential order hypothesis: %d mismatches\n", bad);
    return 0;
}
