// How many kernels from different HIP streams does the GPU run at once?  Each stream gets K back-to-back launches of a kernel that
// spins for ~T us on G workgroups; wall time for S streams relative to one stream tells the concurrency the hardware queues deliver.
//   hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o /tmp/queue_probe && /tmp/queue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, int* sink)
{
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}
int main()
{
    const int K = 50;
    for (int G : {8, 64, 256, 1024, 4096}) {
        for (int S : {1, 2, 3, 4, 8}) {
            std::vector<hipStream_t> st(S);
            for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            for (int w = 0; w < 2; ++w) {
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                for (int k = 0; k < K; ++k)
                    for (auto& s : st) hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, s, 100000LL /* shader-clock cycles: ~50 us */, nullptr);
                hipDeviceSynchronize();
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (w == 1) printf("G=%5d blocks  S=%d streams: %8.1f us for %d kernels/stream = %6.2f us per kernel-slot, concurrency vs S=1 printed by hand\n", G, S, us, K, us / K);
            }
            for (auto& s : st) hipStreamDestroy(s);
        }
    }
    return 0;
}
