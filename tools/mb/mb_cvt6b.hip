// Semantics probe for v_cvt_scalef32_2xpk16_fp6_f32 (gfx950): element order, scale interpretation, rounding, saturation.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
__global__ void k(const float* x, const float* sc, unsigned* o) {
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = x[threadIdx.x * 32 + i]; b[i] = x[threadIdx.x * 32 + 16 + i]; }
    u32x6 r;
    const float scv = sc[threadIdx.x];
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(scv));
    for (int i = 0; i < 6; ++i) o[threadIdx.x * 6 + i] = r[i];
}
static float dec(int c) { int s = c >> 5, e = (c >> 3) & 3, m = c & 7; float v = e == 0 ? m / 8.f : (1 + m / 8.f) * (1 << (e - 1)); return s ? -v : v; }
int main() {
    const int L = 64;
    std::vector<float> x(L * 32), sc(L);
    srand(3);
    for (int l = 0; l < L; ++l) {
        sc[l] = l < 8 ? 1.f : ldexpf(1.f + (l % 3) * 0.25f, (l % 9) - 4);          // some scales are not powers of two
        for (int i = 0; i < 32; ++i) x[l * 32 + i] = l == 0 ? (i * 0.25f - 4.f) : ((rand() / (float)RAND_MAX) * 16.f - 8.f) * (l < 8 ? 1.f : ldexpf(1.f, (l % 9) - 4));
    }
    float *dx, *ds; unsigned* dout;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&ds, L * 4); hipMalloc(&dout, L * 24);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ds, sc.data(), L * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(L), 0, 0, dx, ds, dout);
    std::vector<unsigned> o(L * 6);
    hipMemcpy(o.data(), dout, L * 24, hipMemcpyDeviceToHost);
    // lane 0: print decoded codes in bit order
    printf("lane 0 inputs: "); for (int i = 0; i < 32; ++i) printf("%g ", x[i]); printf("\nlane 0 decoded (bit order): ");
    for (int j = 0; j < 32; ++j) { int bit = 6 * j, w = bit >> 5, s = bit & 31; unsigned long long v = o[w] >> s; if (s > 26) v |= (unsigned long long)o[w + 1] << (32 - s); printf("%g ", dec(v & 63)); }
    printf("\n");
    // hypotheses: order A: code j = element j (a then b); order B: interleaved (j even -> a[j/2], j odd -> b[j/2]); scale: divide by 2^floor(log2 sc)
    for (int hyp = 0; hyp < 2; ++hyp) {
        int bad = 0;
        for (int l = 0; l < L; ++l) for (int j = 0; j < 32; ++j) {
            int bit = 6 * j, w = bit >> 5, s = bit & 31; unsigned long long v = o[l * 6 + w] >> s; if (s > 26) v |= (unsigned long long)o[l * 6 + w + 1] << (32 - s);
            int src = hyp == 0 ? j : ((j & 1) ? 16 + j / 2 : j / 2);
            int ex; frexpf(sc[l], &ex); float p2 = ldexpf(1.f, ex - 1);
            float y = x[l * 32 + src] / p2, ay = fabsf(y), idx;
            if (ay < 2) idx = rintf(ay * 8); else if (ay < 4) idx = 16 + rintf((ay - 2) * 4); else idx = fminf(24 + rintf((ay - 4) * 2), 31);
            int code = (int)idx | (y < 0 || (y == 0 && signbit(y)) ? 32 : 0);
            if ((int)(v & 63) != code) { if (bad < 6) printf("hyp %d lane %d j %d: hw %d (%g) sw %d (%g) y %g\n", hyp, l, j, (int)(v & 63), dec(v & 63), code, dec(code), y); ++bad; }
        }
        printf("hypothesis %d (%s): %d mismatches\n", hyp, hyp == 0 ? "a then b" : "interleaved", bad);
    }
    return 0;
}
