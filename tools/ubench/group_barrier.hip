// Microbenchmark: what does a barrier among the G workgroups of one hypothesis cost inside a resident (persistent) kernel, when
// those workgroups sit on ONE XCD (arrival counter and polls served by that XCD's L2) and when they are spread over all eight?
// 1280 workgroups x 256 threads (cfg2's step grid), groups of 20, R rounds; per round every workgroup does a dependent chain of D
// global loads (stand-in for work), thread 0 adds 1 to its group's counter and polls it until the round's target.
//   hipcc --offload-arch=gfx950 -O3 -o group_barrier tools/ubench/group_barrier.hip && ./group_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define XCC_ID_REG ((3 << 11) | 20)  // hwreg(HW_REG_XCC_ID, 0, 4)

template <int MODE>  // 0 no barrier, 1 barrier, groups on one XCD, 2 barrier, groups spread over the XCDs, 3 as 1 with a returning add,
                     // 4 as 3 + agent-scope acquire after the poll (buffer_inv sc1), 5 as 4 + agent-scope release before the add (buffer_wbl2 sc1)
                     // (every round each thread also stores 16 bytes and does one 64-bit atomicMin: dirty lines for the write-back)
__global__ __launch_bounds__(256) void rounds_kernel(unsigned* __restrict__ cnt, const unsigned* __restrict__ chain, int R, int D, int G, int NG,
                                                     unsigned* __restrict__ xcc_of_wg, unsigned* __restrict__ err, unsigned* __restrict__ sink, unsigned* __restrict__ W)
{
    const unsigned L = blockIdx.x;
    unsigned b;
    if (MODE == 2) b = L / G;
    else b = (L % 8) + 8 * ((L / 8) % (NG / 8));
    if (threadIdx.x == 0) xcc_of_wg[L] = __builtin_amdgcn_s_getreg(XCC_ID_REG);
    unsigned* c = cnt + (size_t)b * 32;  // one 128-byte line per group
    unsigned p = threadIdx.x + L * 7;
    for (int r = 0; r < R; ++r) {
        for (int d = 0; d < D; ++d) p = chain[(p + d) & 0xFFFFF];  // dependent loads
        if (W) {
            W[((size_t)L * 256 + threadIdx.x) * 4 + (r & 3)] = p;
            atomicMin((unsigned long long*)W + (size_t)(1 << 22) + (((size_t)L * 256 + threadIdx.x + (r & 7) * 327680) & 0x3FFFFF), (unsigned long long)p);
        }
        if (MODE != 0) {
            __syncthreads();  // (all the workgroup's work issued)
            if (threadIdx.x == 0) {
                __builtin_amdgcn_s_waitcnt(0);  // stores / atomics of this wave at L2
                if (MODE == 5) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (MODE >= 3) { volatile unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); (void)old; }
                else __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)(r + 1) * G;
                unsigned spins = 0;
                while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicOr(err, 1u); break; }
                }
            }
            if (MODE >= 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
        }
    }
    if (p == 0xFFFFFFFFu) sink[0] = p;
}

int main()
{
    const int G = 20, NG = 64, WG = G * NG;
    unsigned *cnt, *chain, *xcc, *err, *sink, *W;
    (void)hipMalloc(&W, (size_t)64 << 20); (void)hipMemset(W, 0xff, (size_t)64 << 20);
    (void)hipMalloc(&cnt, NG * 128); (void)hipMalloc(&chain, 4 << 20); (void)hipMalloc(&xcc, WG * 4); (void)hipMalloc(&err, 4); (void)hipMalloc(&sink, 4);
    std::vector<unsigned> h(1 << 20);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 12; }
    (void)hipMemcpy(chain, h.data(), 4 << 20, hipMemcpyHostToDevice);
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, rounds_kernel<1>, 256, 0);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    printf("{\"resident_capacity\": %d, \"workgroups\": %d}\n", occ * prop.multiProcessorCount, WG);
    if (occ * prop.multiProcessorCount < WG) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int R = 200;
    for (int D : {0, 2, 6}) {
        float t[6];
        for (int mode = 0; mode < 6; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipMemset(cnt, 0, NG * 128); (void)hipMemset(err, 0, 4);
                (void)hipEventRecord(e0);
                if (mode == 0) rounds_kernel<0><<<WG, 256>>>(cnt, chain, R, D, G, NG, xcc, err, sink, W);
                if (mode == 1) rounds_kernel<1><<<WG, 256>>>(cnt, chain, R, D, G, NG, xcc, err, sink, W);
                if (mode == 2) rounds_kernel<2><<<WG, 256>>>(cnt, chain, R, D, G, NG, xcc, err, sink, W);
                if (mode == 3) rounds_kernel<3><<<WG, 256>>>(cnt, chain, R, D, G, NG, xcc, err, sink, W);
                if (mode == 4) rounds_kernel<4><<<WG, 256>>>(cnt, chain, R, D, G, NG, xcc, err, sink, W);
                if (mode == 5) rounds_kernel<5><<<WG, 256>>>(cnt, chain, R, D, G, NG, xcc, err, sink, W);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            t[mode] = best * 1e3f / R;
        }
        unsigned herr; (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        printf("{\"dependent_loads_per_round\": %d, \"us_per_round_no_barrier\": %.2f, \"same_xcd_barrier\": %.2f, \"spread_barrier\": %.2f, \"same_xcd_returning_add\": %.2f, \"same_xcd_returning_add_acquire\": %.2f, \"same_xcd_returning_add_release_acquire\": %.2f, \"timeout\": %u}\n",
               D, t[0], t[1], t[2], t[3], t[4], t[5], herr);
    }
    std::vector<unsigned> hx(WG);
    (void)hipMemcpy(hx.data(), xcc, WG * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int L = 0; L < WG; ++L) bad += (hx[L] != (unsigned)(L % 8));
    printf("{\"workgroups_not_on_xcd_L_mod_8\": %d, \"first_ids\": [%u,%u,%u,%u,%u,%u,%u,%u,%u,%u]}\n", bad, hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7], hx[8], hx[9]);
    return 0;
}
