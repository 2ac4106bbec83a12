// Microbenchmark: the rate at which MI355X takes COALESCED stores of the step kernel's shape -- every lane one 16-byte record (the
// clip-space vertex) and one 8-byte record (its window snap), consecutive lanes consecutive records -- as whole 64-byte lines per
// second, for a footprint like cfg2's (64 hypotheses x 10 449 vertices: 16 MB) and a large one.  The ceiling the coalesced store
// lines of step_kernel are priced against in bench.py (roofline.lines.store_ceiling); the random-line ceiling of gather_rate.hip does
// not apply to them.   hipcc --offload-arch=gfx950 -O3 -o store_rate tools/ubench/store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void store_kernel(float4* __restrict__ a, int2* __restrict__ b, size_t n, float v)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    a[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
    b[i] = make_int2((int)i, (int)blockIdx.x);
}

int main()
{
    for (size_t n : {(size_t)64 * 10449, (size_t)64 * 10449 * 16}) {
        float4* a; int2* b;
        (void)hipMalloc(&a, n * sizeof(float4));
        (void)hipMalloc(&b, n * sizeof(int2));
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 8; ++rep) {
            (void)hipEventRecord(e0);
            store_kernel<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n, (float)rep);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double bytes = (double)n * 24.0;
        printf("{\"records\": %zu, \"bytes\": %.0f, \"us\": %.2f, \"GB_per_s\": %.1f, \"Glines64_per_s\": %.2f}\n", n, bytes, best * 1e3, bytes / (best * 1e-3) * 1e-9,
               bytes / 64.0 / (best * 1e-3) * 1e-9);
        (void)hipFree(a); (void)hipFree(b);
    }
    return 0;
}
