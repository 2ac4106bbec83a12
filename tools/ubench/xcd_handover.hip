// Microbenchmark (round 4): can the workgroups of one hypothesis hand PLAIN-STORED data to each other inside a resident kernel
// without agent-scope fences, when they all sit on one XCD?  1024 workgroups x 256 threads in 64 groups of 16 (group b on XCD
// b % 8: workgroup id = b + 64 g).  Per round every thread stores one word of a round-dependent pattern into its workgroup's
// 1 KB slot, the workgroup waits for its stores (s_waitcnt vmcnt(0)), arrives at the group's counter (own cache line) and polls
// it; then -- variant -- invalidates its caches; then every thread reads the word a NEIGHBOUR workgroup of the group stored
// (vector load) and wave 0 reads one through the scalar cache, and both are compared with the pattern.  The same addresses are
// read every round, so a cache that is not invalidated serves stale lines.
//   variant 0: nothing      1: buffer_inv sc0 (+ s_dcache_inv)     2: buffer_inv sc1 (+ s_dcache_inv)     3: vector loads as
//   agent-scope atomic loads, no invalidate (scalar path not checked)
//   hipcc --offload-arch=gfx950 -O3 -o xcd_handover tools/ubench/xcd_handover.hip && ./xcd_handover
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned pattern(unsigned r, unsigned b, unsigned g, unsigned t) { return (r * 2654435761u) ^ (b * 40503u + g * 977u + t * 7u + 1u); }

template <int VAR>
__global__ __launch_bounds__(256, 4) void rounds(unsigned* __restrict__ W, unsigned* __restrict__ cnt, int R, int G, int NB, unsigned* __restrict__ bad, unsigned* __restrict__ bad_s,
                                                 unsigned* __restrict__ err, int spread)
{
    const unsigned L = blockIdx.x;
    const unsigned b = spread ? L / G : L % NB, g = spread ? L % G : L / NB;  // spread = 1: a group's workgroups on all XCDs (for comparison)
    const unsigned t = threadIdx.x;
    unsigned* c = cnt + (size_t)b * 32;
    unsigned nbad = 0, nbad_s = 0;
    for (int r = 0; r < R; ++r) {
        W[((size_t)b * G + g) * 256 + t] = pattern(r, b, g, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            atomicInc(c, 0xffffffffu);
            const unsigned target = (unsigned)(r + 1) * G;
            unsigned spins = 0;
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicOr(err, 1u); break; }
            }
        }
        __syncthreads();
        if (VAR == 1) asm volatile("buffer_inv sc0\n s_dcache_inv\n s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (VAR == 2) asm volatile("buffer_inv sc1\n s_dcache_inv\n s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned g2 = (g + 1 + (unsigned)r) % (unsigned)G;
        const unsigned* src = W + ((size_t)b * G + g2) * 256;
        unsigned v;
        if (VAR == 3) v = __hip_atomic_load(src + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {
            asm volatile("" ::: "memory");  // (a PLAIN load, as the kernels use: the compiler must not reuse last round's value)
            v = src[t];
        }
        nbad += v != pattern(r, b, g2, t);
        if (VAR != 3 && t < 64) {
            unsigned sv;
            const unsigned long long sp = (unsigned long long)src;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sp), hi = __builtin_amdgcn_readfirstlane((unsigned)(sp >> 32));
            const unsigned* sps = (const unsigned*)(((unsigned long long)hi << 32) | lo);
            asm volatile("s_load_dword %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(sps) : "memory");
            nbad_s += (t == 0) && sv != pattern(r, b, g2, 0);
        }
        __syncthreads();  // (everybody has read before the next round overwrites)
        // a second barrier round so that no workgroup overwrites its slot while a neighbour still reads it
        if (t == 0) {
            atomicInc(c + 16, 0xffffffffu);
            const unsigned target = (unsigned)(r + 1) * G;
            unsigned spins = 0;
            while (__hip_atomic_load(c + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicOr(err, 1u); break; }
            }
        }
        __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
    if (nbad_s) atomicAdd(bad_s, nbad_s);
}

int main()
{
    const int G = 16, NB = 64, WG = G * NB, R = 300;
    unsigned *W, *cnt, *bad, *bad_s, *err;
    (void)hipMalloc(&W, (size_t)WG * 1024); (void)hipMalloc(&cnt, NB * 128); (void)hipMalloc(&bad, 4); (void)hipMalloc(&bad_s, 4); (void)hipMalloc(&err, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int spread = 0; spread < 2; ++spread)
        for (int var = 0; var < 4; ++var) {
            float best = 1e30f;
            unsigned hb = 0, hs = 0, he = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(cnt, 0, NB * 128); (void)hipMemset(bad, 0, 4); (void)hipMemset(bad_s, 0, 4); (void)hipMemset(err, 0, 4); (void)hipMemset(W, 0, (size_t)WG * 1024);
                (void)hipEventRecord(e0);
                if (var == 0) rounds<0><<<WG, 256>>>(W, cnt, R, G, NB, bad, bad_s, err, spread);
                if (var == 1) rounds<1><<<WG, 256>>>(W, cnt, R, G, NB, bad, bad_s, err, spread);
                if (var == 2) rounds<2><<<WG, 256>>>(W, cnt, R, G, NB, bad, bad_s, err, spread);
                if (var == 3) rounds<3><<<WG, 256>>>(W, cnt, R, G, NB, bad, bad_s, err, spread);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
                (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&hs, bad_s, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost);
            }
            const char* names[4] = {"no invalidate", "buffer_inv sc0 + s_dcache_inv", "buffer_inv sc1 + s_dcache_inv", "agent-scope atomic loads, no invalidate"};
            printf("{\"group_on\": \"%s\", \"variant\": \"%s\", \"us_per_round_two_barriers\": %.2f, \"stale_vector_reads\": %u, \"stale_scalar_reads\": %u, \"of\": %d, \"timeout\": %u}\n",
                   spread ? "all XCDs" : "one XCD", names[var], best * 1e3f / R, hb, hs, WG * 256 * R, he);
        }
    return 0;
}
