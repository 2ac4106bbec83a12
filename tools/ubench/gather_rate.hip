// Microbenchmark: the rate at which MI355X serves random 64-byte records (three 16-byte loads per record, the access of the colour
// role to a texq record), as a function of the table size (268 MB = cfg2's 2048^2 texq table, about the Infinity Cache; 1 GiB: HBM)
// and of the records in flight per thread.  The ceiling the close-up regime's shade kernel is priced against.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate tools/ubench/gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>

// the same random records, reading NL of the record's four 16-byte pieces: is the cost per wave-level load instruction or per 64-byte line?
template <int NL>
__global__ __launch_bounds__(256) void pieces_kernel(const float4* __restrict__ table, unsigned long long mask, unsigned long long salt, float* __restrict__ sink)
{
    const unsigned long long i0 = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long r = ((i0 ^ salt) * 0x9E3779B97F4A7C15ull >> 20) & mask;
    const float4* Q = table + r * 4;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NL; ++k) { const float4 v = Q[k]; s += v.x + v.w; }
    if (s == 123456.789f) sink[0] = s;
}

template <int NL>
static void run_pieces(const float4* table, unsigned long long n_rec, unsigned long long n_gather, float* sink)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        pieces_kernel<NL><<<(unsigned)(n_gather / 256), 256>>>(table, n_rec - 1, 0x7654321ull * (rep + 1), sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double cu_cycles_per_wave_load = best * 1e-3 * 2.4e9 * 256.0 / ((double)n_gather / 64.0 * NL);
    printf("{\"table\": \"268 MB\", \"loads_of_16B_per_record\": %d, \"gathers\": %llu, \"us\": %.1f, \"Grecords_per_s\": %.2f, \"CU_cycles_per_wave_level_load\": %.0f}\n",
           NL, n_gather, best * 1e3, n_gather / (best * 1e-3) * 1e-9, cu_cycles_per_wave_load);
}

template <int ILP>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ table, unsigned long long mask, unsigned long long salt, float* __restrict__ sink)
{
    const unsigned long long i0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * ILP;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        const unsigned long long r = (((i0 + k) ^ salt) * 0x9E3779B97F4A7C15ull >> 20) & mask;
        const float4* Q = table + r * 4;
        const float4 a = Q[0], b = Q[1], c = Q[2];
        s += (a.x + b.y) + c.z;
    }
    if (s == 123456.789f) sink[0] = s;
}

template <int ILP>
static void run(const float4* table, unsigned long long n_rec, unsigned long long n_gather, float* sink, const char* what)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned grid = (unsigned)(n_gather / 256 / ILP);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        gather_kernel<ILP><<<grid, 256>>>(table, n_rec - 1, 0x1234567ull * (rep + 1), sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("{\"table\": \"%s\", \"records_in_flight_per_thread\": %d, \"gathers\": %llu, \"us\": %.1f, \"Grecords_per_s\": %.2f, \"GBps_64B_sectors\": %.0f}\n",
           what, ILP, n_gather, best * 1e3, n_gather / (best * 1e-3) * 1e-9, n_gather * 64 / (best * 1e-3) * 1e-9);
}

int main()
{
    const unsigned long long n_big = 1ull << 24;  // 1 GiB
    float4* table; float* sink;
    (void)hipMalloc(&table, n_big * 64);
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(table, 0, n_big * 64);
    (void)hipDeviceSynchronize();
    for (unsigned long long n_gather : {1ull << 20, 1ull << 22}) {
        run<1>(table, 1ull << 22, n_gather, sink, "268 MB");
        run<2>(table, 1ull << 22, n_gather, sink, "268 MB");
        run<4>(table, 1ull << 22, n_gather, sink, "268 MB");
        run<1>(table, n_big, n_gather, sink, "1 GiB");
        run<2>(table, n_big, n_gather, sink, "1 GiB");
        run<4>(table, n_big, n_gather, sink, "1 GiB");
    }
    run_pieces<1>(table, 1ull << 22, 1ull << 22, sink);
    run_pieces<2>(table, 1ull << 22, 1ull << 22, sink);
    run_pieces<3>(table, 1ull << 22, 1ull << 22, sink);
    run_pieces<4>(table, 1ull << 22, 1ull << 22, sink);
    return 0;
}
