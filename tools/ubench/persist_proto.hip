// Microbenchmark (round 5): the memory protocol of a ONE-LAUNCH iteration, before the engine is rewritten around it.
//
// A resident kernel in which a workgroup reads HW_REG_XCC_ID, takes a ticket from THAT XCD's queue and so joins a group of G
// workgroups that physically share one L2 (co-location by construction: nothing is assumed about block -> XCD placement).  Per
// round a group does what the engine's hypothesis does in one iteration:
//   phase A ("step"):  re-arm the depth buffer of the OTHER parity with stores (variant: plain / sc1), 64-bit atomicMin of
//                      G contributions per entry into this parity's depth buffer (variant: agent / workgroup scope), plain
//                      stores of a 1 KB record per workgroup ("clip"), byte flags; group barrier
//   phase B ("shade"): read the whole depth buffer (variant: plain / sc1 loads) and compare every entry with the analytic
//                      minimum -- a lost re-arm, a stale L1 / L2 line or a lost atomic all show --, read a neighbour's record
//                      (variant: plain / sc1 vector loads; scalar loads behind s_dcache_inv), store a "partial row" that every
//                      workgroup of the group reads back in the next round's phase A; group barrier
// with a pseudo-random delay per (round, workgroup) so that the group is skewed (the guide: test hand-offs under uneven load, L1
// warm -- the same addresses every round).  Reports errors by kind, the per-XCD census of the tickets and us per round.
//   hipcc --offload-arch=gfx950 -O3 -o persist_proto tools/ubench/persist_proto.hip && ./persist_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define XCC_ID_REG ((3 << 11) | 20)  // hwreg(HW_REG_XCC_ID, 0, 4)
#define NZ 2048                      // depth-buffer entries per group and parity (16 KB)

__host__ __device__ __forceinline__ unsigned hash3(unsigned a, unsigned b, unsigned c)
{
    unsigned h = a * 2654435761u ^ (b + 0x9e3779b9u) * 40503u ^ (c * 2246822519u + 0x85ebca6bu);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return h;
}

struct Ctl {
    unsigned* q;        // [8][32] ticket counters, one line per XCD
    unsigned* bar;      // [NB][64] two barrier counters per group, a line each
    unsigned* err;      // [8]: 0 depth mismatches, 1 record mismatches, 2 scalar mismatches, 3 partial-row mismatches, 4 timeouts, 5 excess workgroups
    unsigned* census;   // [WG] xcc | group << 8 | slot << 20
    unsigned long long* Z;  // [NB][2][NZ]
    unsigned* W;        // [NB][G][256] records
    unsigned* P;        // [NB][G][32] partial rows
    unsigned char* F;   // [NB][2][NZ] flags
};

// VAR bits: 1 = depth loads sc1 (else plain), 2 = record / row loads sc1 (else plain), 4 = re-arm with sc1 stores (else plain),
//           8 = atomicMin at workgroup scope (else agent), 16 = SMALL footprint: a workgroup checks a 64-entry window of the depth buffer only, so that
//           its reads stay L1-resident from round to round (a 16 KB sweep per workgroup evicts itself and hides a missing L1 bypass)
template <int VAR>
__global__ __launch_bounds__(256, 4) void persist(Ctl C, int R, int G, int NB, int skew)
{
    __shared__ unsigned s_t;
    const unsigned tid = threadIdx.x;
    const unsigned xcc = __builtin_amdgcn_s_getreg(XCC_ID_REG) & 15u;
    if (tid == 0) s_t = __hip_atomic_fetch_add(C.q + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned t = s_t, grp = t / (unsigned)G, k = t % (unsigned)G;
    const unsigned per_xcd = (unsigned)NB / 8u;
    if (tid == 0) C.census[blockIdx.x] = xcc | (grp << 8) | (k << 20);
    if (grp >= per_xcd) { if (tid == 0) atomicAdd(C.err + 5, 1u); return; }
    const unsigned b = xcc + 8u * grp;
    unsigned* bar = C.bar + (size_t)b * 64;
    unsigned n_z = 0, n_w = 0, n_s = 0, n_p = 0;
    bool dead = false;
    auto barrier = [&](unsigned* c, unsigned target) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            volatile unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)old;
            unsigned spins = 0;
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 21)) { atomicAdd(C.err + 4, 1u); break; }
            }
        }
        __syncthreads();
    };
    for (int r = 0; r < R && !dead; ++r) {
        const unsigned par = (unsigned)r & 1u;
        unsigned long long* Zc = C.Z + ((size_t)b * 2 + par) * NZ;
        unsigned long long* Zo = C.Z + ((size_t)b * 2 + (par ^ 1u)) * NZ;
        // ---- phase A
        if (skew) {
            const unsigned d = hash3((unsigned)r, b, k) % (unsigned)skew;
            for (unsigned i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(8);
        }
        // last round's partial rows of the whole group (written in phase B of round r - 1)
        if (r > 0 && tid < (unsigned)G * 32u) {
            const unsigned g2 = tid / 32u, j = tid % 32u;
            const unsigned* src = C.P + ((size_t)b * G + g2) * 32 + j;
            unsigned v;
            if (VAR & 2) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else { asm volatile("" ::: "memory"); v = *src; }
            n_p += v != hash3((unsigned)(r - 1), b * 64 + g2, j + 77u);
        }
        // re-arm the other parity (entries j with j % G == k), as update_head does
        if (r > 0)
            for (unsigned j = k + (unsigned)G * tid; j < NZ; j += (unsigned)G * 256u) {
                if (VAR & 4) __hip_atomic_store(Zo + j, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else Zo[j] = ~0ull;
            }
        // G contributions per entry, one from each workgroup: the minimum is base(r, j)
        for (unsigned j = tid; j < NZ; j += 256u) {
            const unsigned long long base = ((unsigned long long)(hash3((unsigned)r, b, j) >> 1) << 32) | j;
            const unsigned long long key = base + ((unsigned long long)((k + j + (unsigned)r) % (unsigned)G) << 32);
            if (VAR & 8) __hip_atomic_fetch_min(Zc + j, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_min(Zc + j, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        C.W[((size_t)b * G + k) * 256 + tid] = hash3((unsigned)r, b * 64 + k, tid);
        C.F[((size_t)b * 2 + par) * NZ + (k * 97u + tid) % NZ] = (unsigned char)(r + 1);
        barrier(bar, (unsigned)(r + 1) * G);
        // ---- phase B
        asm volatile("s_dcache_inv\n s_waitcnt lgkmcnt(0)" ::: "memory");
        for (unsigned j = (VAR & 16) ? k * 64u + tid : tid; j < ((VAR & 16) ? (tid < 64u ? NZ : 0u) : NZ); j += (VAR & 16) ? NZ : 256u) {
            unsigned long long v;
            if (VAR & 1) v = __hip_atomic_load(Zc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else { asm volatile("" ::: "memory"); v = Zc[j]; }
            const unsigned long long base = ((unsigned long long)(hash3((unsigned)r, b, j) >> 1) << 32) | j;
            n_z += v != base;
        }
        {
            const unsigned g2 = (k + 1u + (unsigned)r) % (unsigned)G;
            const unsigned* src = C.W + ((size_t)b * G + g2) * 256;
            unsigned v;
            if (VAR & 2) v = __hip_atomic_load(src + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else { asm volatile("" ::: "memory"); v = src[tid]; }
            n_w += v != hash3((unsigned)r, b * 64 + g2, tid);
            if (tid < 64) {
                unsigned sv;
                const unsigned long long sp = (unsigned long long)src;
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sp), hi = __builtin_amdgcn_readfirstlane((unsigned)(sp >> 32));
                const unsigned* sps = (const unsigned*)(((unsigned long long)hi << 32) | lo);
                asm volatile("s_load_dword %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(sps) : "memory");
                n_s += (tid == 0) && sv != hash3((unsigned)r, b * 64 + g2, 0u);
            }
        }
        if (tid < 32) C.P[((size_t)b * G + k) * 32 + tid] = hash3((unsigned)r, b * 64 + k, tid + 77u);
        barrier(bar + 32, (unsigned)(r + 1) * G);
    }
    if (n_z) atomicAdd(C.err + 0, n_z);
    if (n_w) atomicAdd(C.err + 1, n_w);
    if (n_s) atomicAdd(C.err + 2, n_s);
    if (n_p) atomicAdd(C.err + 3, n_p);
}

int main()
{
    const int G = 16, NB = 64, WG = G * NB, R = 400;
    Ctl C;
    (void)hipMalloc(&C.q, 8 * 128); (void)hipMalloc(&C.bar, (size_t)NB * 256); (void)hipMalloc(&C.err, 32); (void)hipMalloc(&C.census, WG * 4);
    (void)hipMalloc(&C.Z, (size_t)NB * 2 * NZ * 8); (void)hipMalloc(&C.W, (size_t)NB * G * 1024); (void)hipMalloc(&C.P, (size_t)NB * G * 128);
    (void)hipMalloc(&C.F, (size_t)NB * 2 * NZ);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<unsigned> cen(WG);
    const int vars[] = {0, 1, 2, 3, 7, 11, 16, 19};
    for (int skew : {0, 6})
        for (int var : vars) {
            float best = 1e30f;
            unsigned he[8] = {0};
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipMemset(C.q, 0, 8 * 128); (void)hipMemset(C.bar, 0, (size_t)NB * 256); (void)hipMemset(C.err, 0, 32);
                (void)hipMemset(C.Z, 0xff, (size_t)NB * 2 * NZ * 8); (void)hipMemset(C.W, 0, (size_t)NB * G * 1024); (void)hipMemset(C.P, 0, (size_t)NB * G * 128);
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0);
                switch (var) {
                    case 0: persist<0><<<WG, 256>>>(C, R, G, NB, skew); break;
                    case 16: persist<16><<<WG, 256>>>(C, R, G, NB, skew); break;
                    case 19: persist<19><<<WG, 256>>>(C, R, G, NB, skew); break;
                    case 1: persist<1><<<WG, 256>>>(C, R, G, NB, skew); break;
                    case 2: persist<2><<<WG, 256>>>(C, R, G, NB, skew); break;
                    case 3: persist<3><<<WG, 256>>>(C, R, G, NB, skew); break;
                    case 7: persist<7><<<WG, 256>>>(C, R, G, NB, skew); break;
                    case 11: persist<11><<<WG, 256>>>(C, R, G, NB, skew); break;
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
                (void)hipMemcpy(he, C.err, 32, hipMemcpyDeviceToHost);
            }
            (void)hipMemcpy(cen.data(), C.census, WG * 4, hipMemcpyDeviceToHost);
            int per[8] = {0}, off_mod = 0;
            for (int i = 0; i < WG; ++i) { per[cen[i] & 15]++; off_mod += (int)(cen[i] & 15) != i % 8; }
            printf("{\"skew\": %d, \"depth_loads\": \"%s\", \"data_loads\": \"%s\", \"rearm\": \"%s\", \"atomics\": \"%s\", \"us_per_round\": %.2f, "
                   "\"footprint\": \"%s\", \"bad_depth\": %u, \"bad_record\": %u, \"bad_scalar\": %u, \"bad_row\": %u, \"timeouts\": %u, \"excess_wg\": %u, "
                   "\"of_depth\": %lld, \"per_xcd\": [%d,%d,%d,%d,%d,%d,%d,%d], \"blocks_not_on_id_mod_8\": %d}\n",
                   skew, (var & 1) ? "sc1" : "plain", (var & 2) ? "sc1" : "plain", (var & 4) ? "sc1 store" : "plain store", (var & 8) ? "workgroup" : "agent",
                   best * 1e3f / R, (var & 16) ? "small (L1-warm)" : "16 KB sweep", he[0], he[1], he[2], he[3], he[4], he[5], (long long)WG * NZ * R, per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7], off_mod);
        }
    return 0;
}
