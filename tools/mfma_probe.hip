// mfma_probe.hip — raw v_mfma_f32_32x32x2_f32 issue rate from one wave per SIMD (sanity for kernels_conv.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters, long long* cyc)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int blocks, int iters)
{
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double mfma = (double)iters * 16;
        printf("NACC=%d blocks=%4d: %.3f ms, %.1f counter-ticks/MFMA, %.1f ns/MFMA/wave, chip %.1f TFLOP/s\n", NACC, blocks, ms,
               c / mfma, ms * 1e6 / mfma, 2.0 * 2048 * mfma * 4 * blocks / (ms * 1e-3) / 1e12);
    }
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<4>(43, 20000); run<4>(256, 20000); run<4>(512, 20000); run<1>(256, 20000); run<2>(256, 20000);
    return 0;
}
