// v_cvt_f16_f32 against v_cvt_pk_f16_f32 (gfx950) over every binade incl. the f16 subnormal range: do they round identically?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
__global__ void k(const float* x, unsigned* a, unsigned* b, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    unsigned r1, r2;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r1) : "v"(v));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(r2) : "v"(v));
    a[i] = r1 & 0xffff; b[i] = r2 & 0xffff;
}
int main()
{
    const int n = 1 << 22;
    float* hx = new float[n];
    unsigned s = 12345;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        int e = (int)((s >> 8) % 40) - 30;            // 2^-30 .. 2^9
        s = s * 1664525u + 1013904223u;
        float m = 1.0f + (float)(s >> 9) / 8388608.0f;
        hx[i] = ldexpf(m, e) * ((s & 1) ? -1.f : 1.f);
    }
    float* dx; unsigned *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, n);
    unsigned* ha = new unsigned[n]; unsigned* hb = new unsigned[n];
    hipMemcpy(ha, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, db, n * 4, hipMemcpyDeviceToHost);
    int diff = 0;
    for (int i = 0; i < n; ++i) if (ha[i] != hb[i]) { if (diff < 10) printf("x=%g (%a) cvt=%04x pk=%04x\n", hx[i], hx[i], ha[i], hb[i]); ++diff; }
    printf("differing: %d of %d\n", diff, n);
    return 0;
}
