// How many kernels from different HIP streams does the GPU run at once?  S streams x R launches of a kernel that spins ~T us on W workgroups.
// hipcc --offload-arch=gfx950 -O2 tools/probe/queue_probe.hip -o tools/probe/queue_probe && tools/probe/queue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, int* sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}
int main()
{
    const int R = 50;
    const long long cyc = 100 * 100;                 // wall_clock64: 100 MHz -> 100 us
    for (int W : {32, 256, 2048}) {
        for (int S : {1, 2, 3, 4, 6, 8}) {
            std::vector<hipStream_t> st(S);
            for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            for (auto& s : st) hipLaunchKernelGGL(spin, dim3(W), dim3(256), 0, s, cyc, nullptr);
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, st[0]);
            for (int i = 1; i < S; ++i) hipStreamWaitEvent(st[i], e0, 0);
            for (int r = 0; r < R; ++r)
                for (auto& s : st) hipLaunchKernelGGL(spin, dim3(W), dim3(256), 0, s, cyc, nullptr);
            std::vector<hipEvent_t> done(S);
            for (int i = 1; i < S; ++i) { hipEventCreate(&done[i]); hipEventRecord(done[i], st[i]); hipStreamWaitEvent(st[0], done[i], 0); }
            hipEventRecord(e1, st[0]);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("workgroups %4d  streams %d: %.2f ms for %d x %d launches of 100 us  -> %.2f kernels at once\n", W, S, ms, S, R, S * R * 0.1 / ms);
            for (auto& s : st) hipStreamDestroy(s);
        }
    }
    return 0;
}
