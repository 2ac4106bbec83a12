// Microbenchmark: cost of a grid-wide barrier inside one persistent launch (what a one-launch iteration would pay instead of
// kernel boundaries).  All workgroups resident (grid <= 256 CUs x per-CU occupancy); monotone counter barrier with a BOUNDED spin
// (a failed assumption ends the kernel with an error flag instead of hanging the GPU).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/grid_barrier.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        unsigned polls = 0;
        while (__atomic_load_n(counter, __ATOMIC_RELAXED) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > 2000000u) { ok = false; *err = 1; break; }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

// variant: one counter per XCD-sized group of workgroups + a top counter (less contention on one address)
__global__ __launch_bounds__(256) void k_barriers(unsigned* counter, int rounds, int* err, float* sink)
{
    float acc = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < 64; ++i) acc = acc * 1.0001f + 1.0f;  // a little work between barriers
        if (!grid_barrier(counter, (unsigned)(r + 1) * gridDim.x, err)) break;
    }
    if (acc == 12345.f) sink[0] = acc;
}

int main()
{
    unsigned* counter;
    int* err;
    float* sink;
    hipMalloc(&counter, 4);
    hipMalloc(&err, 4);
    hipMalloc(&sink, 4);
    for (int grid : {256, 512, 1024, 2048}) {
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_barriers, 256, 0);
        if (grid > nb * 256) { printf("grid %d exceeds residency (%d per CU)\n", grid, nb); continue; }
        for (int rounds : {1, 101, 401}) {
            hipMemset(counter, 0, 4);
            hipMemset(err, 0, 4);
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            k_barriers<<<grid, 256>>>(counter, rounds, err, sink);
            hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            int herr = 0;
            hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            printf("grid %4d rounds %3d: %.1f us total%s\n", grid, rounds, us, herr ? "  (SPIN LIMIT HIT)" : "");
        }
    }
    return 0;
}
