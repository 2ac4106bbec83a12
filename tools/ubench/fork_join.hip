// Microbenchmark: what forking a chain of kernels onto a second stream and joining it again costs on MI355X, per mechanism.
// Two chains of n kernels of ~20 us each (one workgroup spinning on the constant clock: no resource contention, the chains overlap
// perfectly), one kernel before the fork and one behind the join on the main stream -- the shape of a two-chain run of the engine
// (engine.hip engine_run_impl).  Ideal = (n + 2) x 20 us; everything above it is the mechanism.
//   events      hipEventRecord + hipStreamWaitEvent, events created with hipEventDisableTiming (what the engine uses)
//   events_t    the same with default (timing) events
//   value       hipStreamWriteValue32 on the producer stream + hipStreamWaitValue32 on the consumer stream (signal memory)
//   flag        no stream dependency at all: the producer chain's last kernel stores a word, the consumer kernel spins on it
//               (legal only without a data hand-over across XCDs: a lower bound, not an option for the engine)
//   one_chain   2 n + 2 kernels on one stream (what the fork is competing with when the chains do NOT overlap anything)
// hipcc --offload-arch=gfx950 -O3 -o fork_join tools/ubench/fork_join.hip && ./fork_join
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long ticks, unsigned* wait_on, unsigned wait_val, unsigned* post, unsigned post_val)
{
    if (wait_on)
        while (__hip_atomic_load(wait_on, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != wait_val) __builtin_amdgcn_s_sleep(1);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    if (post) __hip_atomic_store(post, post_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main()
{
    const unsigned long long T = 2000;  // 100 MHz clock: 20 us
    hipStream_t s, side;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ef, ej, tf, tj;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    CK(hipEventCreate(&tf));
    CK(hipEventCreate(&tj));
    unsigned* sig = nullptr;
    const bool have_sig = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory) == hipSuccess;
    unsigned* flags = nullptr;
    CK(hipMalloc(&flags, 256));
    CK(hipMemset(flags, 0, 256));
    if (have_sig) CK(hipMemset(sig, 0, 64));
    CK(hipDeviceSynchronize());
    unsigned epoch = 0;
    auto run = [&](int mode, int n) -> double {
        ++epoch;
        const auto t0 = std::chrono::steady_clock::now();
        spin_kernel<<<1, 64, 0, s>>>(T, nullptr, 0, mode == 3 ? flags : nullptr, epoch);  // (flag: "iteration 0 done")
        if (mode == 4) {
            for (int i = 0; i < 2 * n; ++i) spin_kernel<<<1, 64, 0, s>>>(T, nullptr, 0, nullptr, 0);
        } else {
            if (mode == 0) { (void)hipEventRecord(ef, s); (void)hipStreamWaitEvent(side, ef, 0); }
            if (mode == 1) { (void)hipEventRecord(tf, s); (void)hipStreamWaitEvent(side, tf, 0); }
            if (mode == 2) { (void)hipStreamWriteValue32(s, sig, epoch, 0); (void)hipStreamWaitValue32(side, sig, epoch, hipStreamWaitValueEq, 0xffffffffu); }
            for (int i = 0; i < n; ++i) {
                spin_kernel<<<1, 64, 0, s>>>(T, nullptr, 0, nullptr, 0);
                spin_kernel<<<1, 64, 0, side>>>(T, (mode == 3 && i == 0) ? flags : nullptr, epoch, (mode == 3 && i == n - 1) ? flags + 32 : nullptr, epoch);
            }
            if (mode == 0) { (void)hipEventRecord(ej, side); (void)hipStreamWaitEvent(s, ej, 0); }
            if (mode == 1) { (void)hipEventRecord(tj, side); (void)hipStreamWaitEvent(s, tj, 0); }
            if (mode == 2) { (void)hipStreamWriteValue32(side, sig + 8, epoch, 0); (void)hipStreamWaitValue32(s, sig + 8, epoch, hipStreamWaitValueEq, 0xffffffffu); }
        }
        spin_kernel<<<1, 64, 0, s>>>(T, mode == 3 ? flags + 32 : nullptr, epoch, nullptr, 0);
        (void)hipStreamSynchronize(s);
        (void)hipStreamSynchronize(side);
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    const char* names[5] = {"events", "events_t", "value", "flag", "one_chain"};
    for (int n : {5, 20}) {
        for (int mode = 0; mode < 5; ++mode) {
            if (mode == 2 && !have_sig) { printf("{\"mode\": \"value\", \"skipped\": \"no signal memory\"}\n"); continue; }
            for (int w = 0; w < 5; ++w) run(mode, n);
            std::vector<double> v;
            for (int r = 0; r < 25; ++r) v.push_back(run(mode, n));
            std::sort(v.begin(), v.end());
            const double ideal = (mode == 4 ? 2 * n + 2 : n + 2) * 20.0;
            printf("{\"mode\": \"%s\", \"kernels_per_chain\": %d, \"median_us\": %.1f, \"p10_us\": %.1f, \"ideal_us\": %.1f, \"over_ideal_us\": %.1f}\n", names[mode], n,
                   v[v.size() / 2], v[v.size() / 10], ideal, v[v.size() / 2] - ideal);
        }
    }
    return 0;
}
