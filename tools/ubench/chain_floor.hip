// Microbenchmark: what does a kernel cost in a chain of DEPENDENT launches on one stream?  Chains of 1..4 distinct kernels per
// "iteration", each either empty or doing one dependent memory round trip per workgroup, grids of 64 .. 2048 workgroups.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/chain_floor.hip -o /tmp/cf && /tmp/cf
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

template <int ID>
__global__ __launch_bounds__(256) void k_empty(int* p) { if (p && threadIdx.x == 9999) p[ID] = 1; }

// one dependent load chain of `levels` round trips (pointer chase through a small table), then a store
template <int ID>
__global__ __launch_bounds__(256) void k_chase(const int* __restrict__ tab, int* __restrict__ out, int levels)
{
    int i = (blockIdx.x * 256 + threadIdx.x) & 4095;
    for (int l = 0; l < levels; ++l) i = tab[i];
    out[blockIdx.x * 256 + threadIdx.x] = i + ID;
}

int main()
{
    int *tab, *out;
    hipMalloc(&tab, 4096 * 4);
    hipMalloc(&out, 2048 * 256 * 4);
    int h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (i * 1237 + 331) & 4095;
    hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
    hipStream_t s;
    hipStreamCreate(&s);
    const int iters = 2000;
    for (int grid : {64, 512, 1024, 2048})
        for (int levels : {-1, 1, 4}) {
            for (int nk = 1; nk <= 4; ++nk) {
                auto run = [&](int n) {
                    for (int it = 0; it < n; ++it) {
                        if (levels < 0) {
                            k_empty<0><<<grid, 256, 0, s>>>(out);
                            if (nk > 1) k_empty<1><<<grid, 256, 0, s>>>(out);
                            if (nk > 2) k_empty<2><<<grid, 256, 0, s>>>(out);
                            if (nk > 3) k_empty<3><<<grid, 256, 0, s>>>(out);
                        } else {
                            k_chase<0><<<grid, 256, 0, s>>>(tab, out, levels);
                            if (nk > 1) k_chase<1><<<grid, 256, 0, s>>>(tab, out, levels);
                            if (nk > 2) k_chase<2><<<grid, 256, 0, s>>>(tab, out, levels);
                            if (nk > 3) k_chase<3><<<grid, 256, 0, s>>>(tab, out, levels);
                        }
                    }
                };
                run(100);
                hipStreamSynchronize(s);
                auto t0 = std::chrono::steady_clock::now();
                run(iters);
                hipStreamSynchronize(s);
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                printf("grid %4d  %s  kernels/iteration %d : %.2f us per iteration, %.2f us per kernel\n", grid,
                       levels < 0 ? "empty     " : (levels == 1 ? "1 level   " : "4 levels  "), nk, us / iters, us / iters / nk);
            }
        }
    return 0;
}
