// Microbenchmark: do kernels from two HIP streams overlap on one MI355X when each leaves most of the chip idle?
// k_spin: grid WGs of 256 threads, each busy-waits `ticks` x 10 ns.  Chains of dependent launches per stream.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ __launch_bounds__(256) void k_spin(int ticks, int* sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(4);
    if (sink && ticks < 0) sink[0] = 1;
}

static double run(int n_streams, int chain, int grid, int ticks, bool nonblocking)
{
    hipStream_t st[8];
    for (int i = 0; i < n_streams; ++i)
        if (nonblocking) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        else hipStreamCreate(&st[i]);
    for (int i = 0; i < n_streams; ++i) k_spin<<<grid, 256, 0, st[i]>>>(ticks, nullptr);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < chain; ++c)
        for (int i = 0; i < n_streams; ++i) k_spin<<<grid, 256, 0, st[i]>>>(ticks, nullptr);
    hipDeviceSynchronize();
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < n_streams; ++i) hipStreamDestroy(st[i]);
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / chain;
}

int main()
{
    printf("us per chain step (each step = one kernel per stream); kernel = grid x 256 threads spinning 10 us\n");
    printf("%8s %10s %10s %10s %10s\n", "grid", "1 stream", "2 streams", "4 streams", "2 (blocking)");
    for (int grid : {64, 256, 512, 1024, 2048}) {
        const double a = run(1, 300, grid, 1000, true), b = run(2, 300, grid, 1000, true), c = run(4, 300, grid, 1000, true),
                     d = run(2, 300, grid, 1000, false);
        printf("%8d %10.2f %10.2f %10.2f %10.2f\n", grid, a, b, c, d);
    }
    return 0;
}
