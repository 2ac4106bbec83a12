// Microbenchmark: how fast does gfx950 dispatch workgroups?  Empty / short kernels of 256 threads, grid sweep.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/dispatch_rate.hip -o /tmp/dr && /tmp/dr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }

template <int VG>
__global__ __launch_bounds__(256) void k_regs(float* p, int n)
{
    // holds ~VG live registers so the allocation granule is realistic
    float a[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) a[i] = (float)(threadIdx.x + i);
    for (int r = 0; r < n; ++r)
#pragma unroll
        for (int i = 0; i < VG; ++i) a[i] = a[i] * 1.0001f + a[(i + 1) % VG];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += a[i];
    if (s == 12345.678f) p[0] = s;
}

__global__ __launch_bounds__(256) void k_lds(int* p)
{
    __shared__ int sm[2048];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (sm[(threadIdx.x + 1) & 255] == 99999) p[0] = 1;
}

__global__ __launch_bounds__(256) void k_scratch(int* p, int idx)
{
    volatile int loc[64];
    for (int i = 0; i < 64; ++i) loc[i] = i + threadIdx.x;
    if (loc[idx & 63] == 99999) p[0] = 1;
}

template <int VG, bool SCR, bool LDS>
__global__ __launch_bounds__(256) void k_mix(float* p, int n, int idx)
{
    __shared__ float sm[LDS ? 2048 : 1];
    float a[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) a[i] = (float)(threadIdx.x + i);
    for (int r = 0; r < n; ++r)
#pragma unroll
        for (int i = 0; i < VG; ++i) a[i] = a[i] * 1.0001f + a[(i + 1) % VG];
    float s = 0.f;
    if (SCR) {
        volatile float loc[4];
        loc[idx & 3] = a[0];
        s += loc[(idx + 1) & 3];
    }
    if (LDS) {
        sm[threadIdx.x] = a[1];
        __syncthreads();
        s += sm[(threadIdx.x + 7) & 255];
    }
#pragma unroll
    for (int i = 0; i < VG; ++i) s += a[i];
    if (s == 12345.678f) p[0] = s;
}

template <typename F>
static float time_it(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}

int main()
{
    int* d; hipMalloc(&d, 1024);
    float* df = (float*)d;
    const int grids[] = {256, 512, 1024, 2048, 4096, 8192, 16384};
    printf("%8s %10s %10s %10s %10s %10s %10s   (us per launch, back-to-back launches)\n", "WGs", "empty", "lds8k", "scratch", "regs32", "regs96", "threads64");
    for (int g : grids) {
        float t0 = time_it([&] { k_empty<<<g, 256>>>(d); }, 200);
        float t1 = time_it([&] { k_lds<<<g, 256>>>(d); }, 200);
        float t2 = time_it([&] { k_scratch<<<g, 256>>>(d, 3); }, 200);
        float t3 = time_it([&] { k_regs<32><<<g, 256>>>(df, 0); }, 200);
        float t4 = time_it([&] { k_regs<96><<<g, 256>>>(df, 0); }, 200);
        float t5 = time_it([&] { k_empty<<<g * 4, 64>>>(d); }, 200);
        printf("%8d %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f\n", g, t0, t1, t2, t3, t4, t5);
    }
    printf("\n%8s %12s %12s %12s %12s %12s\n", "WGs", "r120", "r120+scr", "r120+lds", "r120+both", "r56+scr");
    for (int g : grids) {
        float t0 = time_it([&] { k_mix<120, false, false><<<g, 256>>>(df, 0, 1); }, 200);
        float t1 = time_it([&] { k_mix<120, true, false><<<g, 256>>>(df, 0, 1); }, 200);
        float t2 = time_it([&] { k_mix<120, false, true><<<g, 256>>>(df, 0, 1); }, 200);
        float t3 = time_it([&] { k_mix<120, true, true><<<g, 256>>>(df, 0, 1); }, 200);
        float t4 = time_it([&] { k_mix<56, true, false><<<g, 256>>>(df, 0, 1); }, 200);
        printf("%8d %12.2f %12.2f %12.2f %12.2f %12.2f\n", g, t0, t1, t2, t3, t4);
    }
    return 0;
}
