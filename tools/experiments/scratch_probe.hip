// Reproducer (round 4) for the run-to-run differences seen in round 3 when shade_kernel's private segment grew from 16 to 512 bytes
// per lane: does a kernel's private (scratch) memory keep its contents when the launch needs more scratch
// than the runtime's per-queue limit?  Every lane fills a private array with a pattern of its global id, waits, reads it back
// through run-time indices (so the array cannot live in registers) and counts mismatches.  Grids of more workgroups than the
// chip holds (scratch slots are reused), alternated with a kernel of another private size and with a scratch-free one, on one
// stream and on two.
//   hipcc --offload-arch=gfx950 -O3 -o scratch_probe scratch_probe.hip && ./scratch_probe
//   HSA_SCRATCH_SINGLE_LIMIT=... ./scratch_probe    (runtime knob, for comparison)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NDW>
__global__ __launch_bounds__(256, 4) void probe(unsigned* bad, const int* __restrict__ perm, int wait_ticks, unsigned salt)
{
    volatile unsigned a[NDW];
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    for (int i = 0; i < NDW; ++i) a[perm[i] % NDW] = (gid * 2654435761u + salt) ^ (unsigned)(perm[i] % NDW * 40503);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)wait_ticks) __builtin_amdgcn_s_sleep(2);
    unsigned nb = 0;
    for (int i = 0; i < NDW; ++i) {
        const int j = perm[(i * 7 + 3) % NDW] % NDW;
        nb += a[j] != ((gid * 2654435761u + salt) ^ (unsigned)(j * 40503));
    }
    if (nb) atomicAdd(bad, nb);
}

__global__ void plain(unsigned* sink) { if (sink && threadIdx.x == 12345) sink[0] = 1; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int NDW, int NOTHER>
static unsigned run_case(int grid, int reps, int n_streams, int wait_ticks, const int* perm, unsigned* bad)
{
    hipStream_t st[2];
    for (int i = 0; i < n_streams; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    CK(hipMemset(bad, 0, 4));
    for (int r = 0; r < reps; ++r)
        for (int i = 0; i < n_streams; ++i) {
            probe<NDW><<<grid, 256, 0, st[i]>>>(bad, perm, wait_ticks, (unsigned)(r * 2 + i));
            plain<<<64, 256, 0, st[i]>>>(nullptr);
            if (NOTHER > 0) probe<(NOTHER > 0 ? NOTHER : 4)><<<grid / 4 + 1, 256, 0, st[i]>>>(bad, perm, wait_ticks / 4, (unsigned)(r * 2 + i + 77));
        }
    CK(hipDeviceSynchronize());
    unsigned h = 0;
    CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n_streams; ++i) CK(hipStreamDestroy(st[i]));
    return h;
}

int main()
{
    unsigned* bad;
    int* perm;
    CK(hipMalloc(&bad, 4));
    std::vector<int> hp(4096);
    for (int i = 0; i < 4096; ++i) hp[i] = (i * 2654435761u) % 4096;
    CK(hipMalloc(&perm, hp.size() * 4));
    CK(hipMemcpy(perm, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const char* lim = getenv("HSA_SCRATCH_SINGLE_LIMIT");
    printf("device %s, %d CUs; HSA_SCRATCH_SINGLE_LIMIT=%s\n", prop.name, prop.multiProcessorCount, lim ? lim : "(default)");
    printf("%-28s %8s %8s %8s %10s\n", "private bytes/lane (+other)", "grid", "streams", "reps", "mismatches");
#define CASE(NDW, NOTHER, grid, ns)                                                                                     \
    do {                                                                                                                \
        const unsigned m = run_case<NDW, NOTHER>(grid, 40, ns, 300, perm, bad);                                         \
        printf("%6d (+%5d)               %8d %8d %8d %10u\n", NDW * 4, NOTHER * 4, grid, ns, 40, m);                     \
        fflush(stdout);                                                                                                 \
    } while (0)
    for (int grid : {1024, 4096, 16384}) {
        CASE(30, 0, grid, 1);
        CASE(64, 0, grid, 1);
        CASE(68, 0, grid, 1);
        CASE(72, 0, grid, 1);
        CASE(128, 0, grid, 1);
        CASE(128, 30, grid, 1);
        CASE(128, 56, grid, 2);
        CASE(256, 0, grid, 1);
        CASE(1024, 128, grid, 2);
    }
    return 0;
}
