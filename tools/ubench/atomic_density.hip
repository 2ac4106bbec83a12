// Microbenchmark: does the cost of 64-bit global atomicMin follow the number of atomic INSTRUCTIONS (waves x issues) or the
// number of active LANES?  Same total number of lane-atomics, issued with 64 / 16 / 8 / 4 active lanes per instruction, on a
// zbuf-like footprint (64 regions of 60x60 pixels in 640x480 frames, ~2 hits per address).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_atomics(unsigned long long* z, const unsigned* addr, int per_lane_total, int active_lanes)
{
    // each wave owns `per_lane_total * 64` lane-atomics; with `active_lanes` active per instruction it issues
    // per_lane_total * 64 / active_lanes instructions
    const int lane = threadIdx.x & 63;
    const size_t wave_global = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned* a = addr + wave_global * per_lane_total * 64;
    const int n_instr = per_lane_total * 64 / active_lanes;
    for (int i = 0; i < n_instr; ++i) {
        if (lane < active_lanes) {
            const unsigned ad = a[i * active_lanes + lane];
            atomicMin(z + ad, ((unsigned long long)(i * 977u + lane) << 32) | (unsigned)wave_global);
        }
    }
}

static void fill(std::vector<unsigned>& h, int mode, int B, int H, int W)
{
    // mode 0: random pixel of a 60x60 region of a random hypothesis per lane-atomic
    // mode 1: every group of 64 consecutive lane-atomics = 64 consecutive pixels of one row (one or two cache lines of 16)
    // mode 2: groups of 4 consecutive pixels (a small triangle's fragments), groups random
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (size_t i = 0; i < h.size(); ++i) {
        const int g = mode == 1 ? 64 : (mode == 2 ? 4 : 1);
        if (i % g == 0 || mode == 0) {
            const unsigned b = rnd() % B, px = 290 + rnd() % (60 - (g > 1 ? g : 0) + (g > 1 ? 1 : 0)), py = 210 + rnd() % 60;
            h[i] = (b * H + py) * W + (g == 64 ? 288 : px);
        } else {
            h[i] = h[i - 1] + 1;
        }
    }
}

int main()
{
    const int B = 64, H = 480, W = 640, waves = 10240, per_lane = 1;  // ~655k lane-atomics in total, like cfg2's 454k fragments
    std::vector<unsigned> h((size_t)waves * per_lane * 64);
    unsigned* d_addr; unsigned long long* d_z;
    hipMalloc(&d_addr, h.size() * 4); hipMalloc(&d_z, (size_t)B * H * W * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
    fill(h, mode, B, H, W);
    hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("address pattern %d\n%8s %12s %14s\n", mode, "lanes", "us/launch", "ns/lane-atomic");
    for (int act : {64, 16, 4}) {
        hipMemset(d_z, 0xFF, (size_t)B * H * W * 8);
        for (int i = 0; i < 3; ++i) k_atomics<<<waves / 4, 256>>>(d_z, d_addr, per_lane, act);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) k_atomics<<<waves / 4, 256>>>(d_z, d_addr, per_lane, act);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%8d %12.2f %14.3f\n", act, ms * 1000 / 20, ms * 1e6 / 20 / (waves * 64.0 * per_lane));
    }
    }
    return 0;
}
