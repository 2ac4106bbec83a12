// Microbenchmark / counter calibration: what does rocprofv3's FETCH_SIZE report for the shading kernel's texel gathers?
// One launch reads a KNOWN number of 64-byte records (three 16-byte loads of the record, as the colour role reads a texq record) at
// pseudo-random positions of a table far larger than L2 + Infinity Cache (1 GiB), every record at most once; a second kernel streams
// the same number of bytes with wide coalesced loads (the pattern MI355X_MICROARCH.md calibrated: FETCH_SIZE = 1/2 of the bytes).
//   hipcc --offload-arch=gfx950 -O3 -o gather64 tools/ubench/gather64.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o g --output-format csv -- ./gather64
// Prints the bytes each kernel must move; compare with FETCH_SIZE (KiB) x 1024 of the two kernels.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void gather64_kernel(const float4* __restrict__ table, unsigned long long n_rec, float* __restrict__ sink)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    // an odd multiplier modulo a power of two: a permutation, every record touched at most once
    const unsigned long long r = (i * 0x9E3779B97F4A7C15ull) & (n_rec - 1);
    const float4* Q = table + r * 4;
    const float4 a = Q[0], b = Q[1], c = Q[2];
    const float s = (a.x + b.y) + c.z;
    if (s == 123456.789f) sink[0] = s;  // (never: keeps the loads alive)
}

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ table, float* __restrict__ sink)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const float4* Q = table + i * 4;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float4 v = Q[k]; s += v.x + v.w; }
    if (s == 123456.789f) sink[0] = s;
}

int main()
{
    const unsigned long long n_rec = 1ull << 24;  // 16 Mi records x 64 B = 1 GiB
    const unsigned long long n_gather = 1ull << 18;  // 262 144 gathers per launch: about the covered pixels of cfg2 (228 k)
    float4* table;
    float* sink;
    (void)hipMalloc(&table, n_rec * 64);
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(table, 0, n_rec * 64);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 10; ++rep) {
        gather64_kernel<<<(unsigned)(n_gather / 256), 256>>>(table + (size_t)rep * 4 * 1024, n_rec / 2, sink);
        stream_kernel<<<(unsigned)(n_gather / 256), 256>>>(table + (n_rec / 2 + (size_t)rep * n_gather) * 4, sink);
    }
    (void)hipDeviceSynchronize();
    printf("{\"gathers_per_launch\": %llu, \"bytes_per_gather_launch_64B_records\": %llu, \"bytes_read_per_gather_launch_48B\": %llu, \"bytes_per_stream_launch\": %llu}\n",
           n_gather, n_gather * 64, n_gather * 48, n_gather * 64);
    return 0;
}
