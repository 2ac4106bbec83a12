// What exactly do v_permlane32_swap / v_permlane16_swap return on gfx950?  Prints the lane ids seen by each lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
    const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1]; out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
}
int main()
{
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 [0] (old=lane)", "swap32 [1] (src=100+lane)", "swap16 [0]", "swap16 [1]"};
    for (int v = 0; v < 4; ++v) {
        printf("%s:\n", names[v]);
        for (int i = 0; i < 64; ++i) printf("%4u%s", h[v * 64 + i], (i % 16 == 15) ? "\n" : "");
    }
    return 0;
}
