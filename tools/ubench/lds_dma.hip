// Check of the direct global -> LDS load (global_load_lds_dword, "LDS DMA") semantics this build relies on: lane l's dword lands at
// lds_base + 4 l, lanes switched off by the exec mask leave their LDS word alone, the wave sees the data after s_waitcnt vmcnt(0).
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma tools/ubench/lds_dma.hip && ./lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void dma_kernel(const unsigned* __restrict__ src, const int* __restrict__ idx, unsigned* __restrict__ out)
{
    __shared__ unsigned buf[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    buf[wave][lane] = 0xAAAA0000u + lane;
    buf[wave][64 + lane] = 0xBBBB0000u + lane;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int i0 = idx[(blockIdx.x * 256 + threadIdx.x) * 2 + 0], i1 = idx[(blockIdx.x * 256 + threadIdx.x) * 2 + 1];
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + i0), (lptr_t)&buf[wave][0], 4, 0, 0);
    if (lane < 36) __builtin_amdgcn_global_load_lds((gptr_t)(src + i1), (lptr_t)&buf[wave][64], 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    out[(blockIdx.x * 256 + threadIdx.x) * 2 + 0] = buf[wave][lane];
    out[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = buf[wave][64 + lane];
}

int main()
{
    const int N = 1 << 20, T = 64 * 256;
    std::vector<unsigned> h(N);
    for (int i = 0; i < N; ++i) h[i] = 0x10000000u + i;
    std::vector<int> hi(T * 2);
    unsigned s = 7;
    for (auto& v : hi) { s = s * 1664525u + 1013904223u; v = (int)(s >> 12); }
    unsigned *src, *out; int* idx;
    (void)hipMalloc(&src, N * 4); (void)hipMalloc(&out, T * 8); (void)hipMalloc(&idx, T * 8);
    (void)hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(idx, hi.data(), T * 8, hipMemcpyHostToDevice);
    dma_kernel<<<64, 256>>>(src, idx, out);
    std::vector<unsigned> ho(T * 2);
    (void)hipMemcpy(ho.data(), out, T * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < T; ++t) {
        const int lane = t & 63;
        const unsigned e0 = h[hi[t * 2]], e1 = lane < 36 ? h[hi[t * 2 + 1]] : 0xBBBB0000u + lane;
        if (ho[t * 2] != e0 || ho[t * 2 + 1] != e1) { if (bad < 5) printf("thread %d: got %08x %08x expected %08x %08x\n", t, ho[t * 2], ho[t * 2 + 1], e0, e1); ++bad; }
    }
    printf("{\"threads\": %d, \"mismatches\": %d}\n", T, bad);
    return bad != 0;
}
