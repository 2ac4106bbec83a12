// renderops.hip -- op-level renderer kernels behind diffdope_amd.render (the nvdiffrast-shaped API
// render_texture_batch is written against, diffdope/diffdope.py:143-231): rasterize backward,
// interpolate, texture(linear, wrap), antialias -- forward and backward.  These are the
// compatibility path (user loss functions that need materialised renders); the timed path is the
// fused engine (engine.hip), which shares raster_math.h with these kernels.
// All kernels: one lane per pixel, 256-thread workgroups.  The full-frame passes of render_texture_batch (antialias, gbuffer,
// masked L1) launch per hypothesis -- blockIdx.y = hypothesis, a workgroup takes PIX_PER_WG consecutive pixels of it -- so
// all pixel arithmetic is 32-bit: with a flat B*H*W index the 64-bit divisions (x, y, hypothesis of a pixel) cost more VALU
// issue than the kernels' memory traffic takes time (antialias 136 -> 60 us on 64 x 640x480).  The remaining ops grid-stride.
#include "raster.h"

#define PIX_GRID(n) ((n + 255) / 256 > 16384 ? 16384 : (int)((n + 255) / 256))
#define PIX_ROUNDS 4
#define PIX_PER_WG (256 * PIX_ROUNDS)
static inline dim3 pix_grid2(long long HW, int B) { return dim3((unsigned)((HW + PIX_PER_WG - 1) / PIX_PER_WG), (unsigned)B); }
#define DDX_REQUIRE_FRAME(B, H, W, what) \
    DDX_REQUIRE((long long)(H) * (W) < (1ll << 30) && (W) <= 65535 && (B) <= 65535, DDX_E_SHAPE, what ": frame of %d x %d pixels / %d hypotheses exceeds the launch limits", H, W, B)

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rasterize_bwd_kernel(const float* __restrict__ pos, const int* __restrict__ tri,
                                                            int V, int T, int B, int H, int W,
                                                            const float* __restrict__ rast,
                                                            const float* __restrict__ drast, float* __restrict__ dpos, int compat)
{
    const long long n = (long long)B * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float4 r = ld4(rast + i * 4);
        const int t = (int)r.w - 1;
        if (t < 0 || t >= T) continue;
        const float4 g = ld4(drast + i * 4);
        if (g.x == 0.f && g.y == 0.f) continue;
        const int px = (int)(i % W), py = (int)((i / W) % H), b = (int)(i / ((long long)W * H));
        const int vi[3] = {tri[t * 3 + 0], tri[t * 3 + 1], tri[t * 3 + 2]};
        const float* P = pos + (size_t)b * V * 4;
        const float4 p0 = ld4(P + (size_t)vi[0] * 4), p1 = ld4(P + (size_t)vi[1] * 4), p2 = ld4(P + (size_t)vi[2] * 4);
        Bary bc;
        if (!pixel_bary(p0, p1, p2, px, py, H, W, bc)) continue;
        float gx[3], gy[3], gw[3];
        bary_backward(bc, g.x, g.y, gx, gy, gw, (compat & DDX_COMPAT_UNCLAMPED_BARY_GRAD) != 0);
        float* D = dpos + (size_t)b * V * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            atomicAdd(D + (size_t)vi[k] * 4 + 0, gx[k]);
            atomicAdd(D + (size_t)vi[k] * 4 + 1, gy[k]);
            atomicAdd(D + (size_t)vi[k] * 4 + 3, gw[k]);
        }
    }
}

extern "C" int ddx_rasterize_bwd(const float* pos, const int32_t* tri, int B, int V, int T, int H, int W,
                                 const float* rast, const float* drast, float* dpos, void* stream)
{
    DDX_REQUIRE(pos && tri && rast && drast && dpos, DDX_E_NULL, "rasterize_bwd: NULL pointer");
    DDX_REQUIRE(B >= 1 && V >= 1 && T >= 1 && H >= 1 && W >= 1, DDX_E_SHAPE, "rasterize_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    DDX_HIP(hipMemsetAsync(dpos, 0, (size_t)B * V * 4 * sizeof(float), s));
    const long long n = (long long)B * H * W;
    rasterize_bwd_kernel<<<PIX_GRID(n), 256, 0, s>>>(pos, tri, V, T, B, H, W, rast, drast, dpos, ddx_compat_flags());
    DDX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interpolate_fwd_kernel(const float* __restrict__ attr, long long abs_, int Va,
                                                              int A, const float* __restrict__ rast,
                                                              const int* __restrict__ tri, int T, long long HW,
                                                              long long n, float* __restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float4 r = ld4(rast + i * 4);
        const int t = (int)r.w - 1;
        float* o = out + i * A;
        if (t < 0 || t >= T) {
            for (int c = 0; c < A; ++c) o[c] = 0.f;
            continue;
        }
        const int b = (int)(i / HW);
        const float* AT = attr + (size_t)b * abs_;
        const int i0 = tri[t * 3 + 0], i1 = tri[t * 3 + 1], i2 = tri[t * 3 + 2];
        if ((unsigned)i0 >= (unsigned)Va || (unsigned)i1 >= (unsigned)Va || (unsigned)i2 >= (unsigned)Va) {
            for (int c = 0; c < A; ++c) o[c] = 0.f;
            continue;
        }
        const float u = r.x, v = r.y, w2 = (1.0f - u) - v;
        const float *a0 = AT + (size_t)i0 * A, *a1 = AT + (size_t)i1 * A, *a2 = AT + (size_t)i2 * A;
        for (int c = 0; c < A; ++c) o[c] = __fmaf_rn(w2, a2[c], __fmaf_rn(v, a1[c], u * a0[c]));
    }
}

__global__ __launch_bounds__(256) void interpolate_bwd_kernel(const float* __restrict__ attr, long long abs_, int Va,
                                                              int A, const float* __restrict__ rast,
                                                              const int* __restrict__ tri, int T, long long HW,
                                                              long long n, const float* __restrict__ dout,
                                                              float* __restrict__ dattr, float* __restrict__ drast)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float4 r = ld4(rast + i * 4);
        const int t = (int)r.w - 1;
        float gu = 0.f, gv = 0.f;
        if (t >= 0 && t < T) {
            const int b = (int)(i / HW);
            const int i0 = tri[t * 3 + 0], i1 = tri[t * 3 + 1], i2 = tri[t * 3 + 2];
            if ((unsigned)i0 < (unsigned)Va && (unsigned)i1 < (unsigned)Va && (unsigned)i2 < (unsigned)Va) {
                const size_t ab = (size_t)b * abs_;
                const float u = r.x, v = r.y, w2 = (1.0f - u) - v;
                for (int c = 0; c < A; ++c) {
                    const float g = dout[i * A + c];
                    const float a0 = attr[ab + (size_t)i0 * A + c], a1 = attr[ab + (size_t)i1 * A + c],
                                a2 = attr[ab + (size_t)i2 * A + c];
                    gu = __fmaf_rn(g, a0 - a2, gu);
                    gv = __fmaf_rn(g, a1 - a2, gv);
                    if (dattr && g != 0.f) {
                        atomicAdd(dattr + ab + (size_t)i0 * A + c, u * g);
                        atomicAdd(dattr + ab + (size_t)i1 * A + c, v * g);
                        atomicAdd(dattr + ab + (size_t)i2 * A + c, w2 * g);
                    }
                }
            }
        }
        *reinterpret_cast<float4*>(drast + i * 4) = make_float4(gu, gv, 0.f, 0.f);
    }
}

extern "C" int ddx_interpolate_fwd(const float* attr, long long abs_, int Va, int A, const float* rast,
                                   const int32_t* tri, int T, int B, int H, int W, float* out, void* stream)
{
    DDX_REQUIRE(attr && rast && tri && out, DDX_E_NULL, "interpolate_fwd: NULL pointer");
    DDX_REQUIRE(Va >= 1 && A >= 1 && A <= 64 && B >= 1 && H >= 1 && W >= 1 && T >= 1, DDX_E_SHAPE, "interpolate_fwd: bad shape");
    const long long n = (long long)B * H * W;
    interpolate_fwd_kernel<<<PIX_GRID(n), 256, 0, (hipStream_t)stream>>>(attr, abs_, Va, A, rast, tri, T, (long long)H * W, n, out);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_interpolate_bwd(const float* attr, long long abs_, int Va, int A, const float* rast,
                                   const int32_t* tri, int T, int B, int H, int W, const float* dout, float* dattr,
                                   float* drast, void* stream)
{
    DDX_REQUIRE(attr && rast && tri && dout && drast, DDX_E_NULL, "interpolate_bwd: NULL pointer");
    DDX_REQUIRE(Va >= 1 && A >= 1 && A <= 64 && B >= 1 && H >= 1 && W >= 1 && T >= 1, DDX_E_SHAPE, "interpolate_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (dattr) DDX_HIP(hipMemsetAsync(dattr, 0, (size_t)(abs_ == 0 ? 1 : B) * Va * A * sizeof(float), s));
    const long long n = (long long)B * H * W;
    interpolate_bwd_kernel<<<PIX_GRID(n), 256, 0, s>>>(attr, abs_, Va, A, rast, tri, T, (long long)H * W, n, dout, dattr, drast);
    DDX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void texture_fwd_kernel(const float* __restrict__ tex, long long tbs, int Th, int Tw,
                                                          int C, const float* __restrict__ uv, long long HW, long long n,
                                                          float* __restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float2 q = *reinterpret_cast<const float2*>(uv + i * 2);
        TexelSetup s;
        tex_setup(q.x, q.y, Th, Tw, s);
        const float* TX = tex + (size_t)(i / HW) * tbs;
        const float *t00 = TX + ((size_t)s.y0 * Tw + s.x0) * C, *t10 = TX + ((size_t)s.y0 * Tw + s.x1) * C,
                    *t01 = TX + ((size_t)s.y1 * Tw + s.x0) * C, *t11 = TX + ((size_t)s.y1 * Tw + s.x1) * C;
        for (int c = 0; c < C; ++c) {
            const float a = __fmaf_rn(s.fx, t10[c] - t00[c], t00[c]);
            const float bq = __fmaf_rn(s.fx, t11[c] - t01[c], t01[c]);
            out[i * C + c] = __fmaf_rn(s.fy, bq - a, a);
        }
    }
}

__global__ __launch_bounds__(256) void texture_bwd_kernel(const float* __restrict__ tex, long long tbs, int Th, int Tw,
                                                          int C, const float* __restrict__ uv, long long HW, long long n,
                                                          const float* __restrict__ dout, float* __restrict__ duv,
                                                          float* __restrict__ dtex)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float2 q = *reinterpret_cast<const float2*>(uv + i * 2);
        TexelSetup s;
        tex_setup(q.x, q.y, Th, Tw, s);
        const size_t tb = (size_t)(i / HW) * tbs;
        const size_t o00 = tb + ((size_t)s.y0 * Tw + s.x0) * C, o10 = tb + ((size_t)s.y0 * Tw + s.x1) * C,
                     o01 = tb + ((size_t)s.y1 * Tw + s.x0) * C, o11 = tb + ((size_t)s.y1 * Tw + s.x1) * C;
        float gu = 0.f, gv = 0.f;
        for (int c = 0; c < C; ++c) {
            const float g = dout[i * C + c];
            if (g == 0.f) continue;
            const float t00 = tex[o00 + c], t10 = tex[o10 + c], t01 = tex[o01 + c], t11 = tex[o11 + c];
            gu = __fmaf_rn(g, __fmaf_rn(s.fy, (t11 - t01) - (t10 - t00), t10 - t00), gu);
            gv = __fmaf_rn(g, __fmaf_rn(s.fx, (t11 - t10) - (t01 - t00), t01 - t00), gv);
            if (dtex) {
                atomicAdd(dtex + o00 + c, g * (1.f - s.fx) * (1.f - s.fy));
                atomicAdd(dtex + o10 + c, g * s.fx * (1.f - s.fy));
                atomicAdd(dtex + o01 + c, g * (1.f - s.fx) * s.fy);
                atomicAdd(dtex + o11 + c, g * s.fx * s.fy);
            }
        }
        *reinterpret_cast<float2*>(duv + i * 2) = make_float2(gu * (float)Tw, gv * (float)Th);
    }
}

extern "C" int ddx_texture_linear_fwd(const float* tex, long long tbs, int Th, int Tw, int C, const float* uv, int B,
                                      int H, int W, float* out, void* stream)
{
    DDX_REQUIRE(tex && uv && out, DDX_E_NULL, "texture_linear_fwd: NULL pointer");
    DDX_REQUIRE(Th >= 1 && Tw >= 1 && C >= 1 && C <= 64 && B >= 1 && H >= 1 && W >= 1, DDX_E_SHAPE, "texture_linear_fwd: bad shape");
    DDX_REQUIRE(((uintptr_t)uv & 7) == 0, DDX_E_ALIGN, "texture_linear_fwd: uv must be 8-byte aligned");
    const long long n = (long long)B * H * W;
    texture_fwd_kernel<<<PIX_GRID(n), 256, 0, (hipStream_t)stream>>>(tex, tbs, Th, Tw, C, uv, (long long)H * W, n, out);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_texture_linear_bwd(const float* tex, long long tbs, int Th, int Tw, int C, const float* uv, int B,
                                      int H, int W, const float* dout, float* duv, float* dtex, int Bt, void* stream)
{
    DDX_REQUIRE(tex && uv && dout && duv, DDX_E_NULL, "texture_linear_bwd: NULL pointer");
    DDX_REQUIRE(Th >= 1 && Tw >= 1 && C >= 1 && C <= 64 && B >= 1 && H >= 1 && W >= 1, DDX_E_SHAPE, "texture_linear_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (dtex) DDX_HIP(hipMemsetAsync(dtex, 0, (size_t)(Bt < 1 ? 1 : Bt) * Th * Tw * C * sizeof(float), s));
    const long long n = (long long)B * H * W;
    texture_bwd_kernel<<<PIX_GRID(n), 256, 0, s>>>(tex, tbs, Th, Tw, C, uv, (long long)H * W, n, dout, duv, dtex);
    DDX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// antialias: one lane per pixel analyses the pairs to its right (d=0) and above (d=1, row+1).
// COVERAGE: the colour is the coverage image itself -- interpolate(ones) of the pixel's rast entry, three equal channels
// (diffdope.py:212-214) -- recomputed from rast instead of read: the forward adds its blends IN PLACE onto the coverage image
// gbuffer_fwd wrote (no copy of the frame), the backward produces d pos only (coverage has no gradient path: no d colour).
static __device__ __forceinline__ float coverage_of(const float4& r, int T)
{
    const int t = (int)r.w - 1;
    if (t < 0 || t >= T) return 0.f;
    const float w2 = (1.0f - r.x) - r.y;
    return __fmaf_rn(w2, 1.0f, __fmaf_rn(r.y, 1.0f, r.x * 1.0f));  // == gbuffer_fwd_kernel's cv
}

#define AA_QUEUE 4096
template <bool BWD, bool COVERAGE>
__global__ __launch_bounds__(256) void antialias_kernel(const float* __restrict__ color, int C,
                                                        const float* __restrict__ rast, const float* __restrict__ pos,
                                                        const int* __restrict__ tri, const int* __restrict__ opp,
                                                        int V, int T, int H, int W, const float* __restrict__ dout,
                                                        float* __restrict__ out /* fwd: out ; bwd: dcolor */,
                                                        float* __restrict__ dpos, const int* __restrict__ row_range = nullptr)
{
    // (row_range: a block of rows that lies outside the rows of the hypothesis' active tiles holds no pair -- and `rast` is not
    // filled there by the restricted emit; blocks that pass read rows within EMIT_ROW_MARGIN of the active ones)
    if (row_range && ((int)blockIdx.x * AA_ROWS > row_range[blockIdx.y * 2 + 1] || (int)blockIdx.x * AA_ROWS + AA_ROWS < row_range[blockIdx.y * 2])) return;
    // A workgroup takes AA_ROWS full rows of one hypothesis (blockIdx.y), 256 columns at a time.
    // (1) streaming: each lane reads the triangle ids of its column in those rows plus the row above -- the upper neighbour of
    // a row is the next row's own id, the right neighbour comes from the next lane -- and queues the pairs whose ids differ (a
    // few per cent of a row that crosses the object, none elsewhere).  (2) the queued pairs are analysed with all lanes busy:
    // a pair is a chain of dependent loads (triangle, vertices, neighbour triangle, ...) that should be walked once per
    // workgroup, not once per row and direction.
    // Measured on 64 x 640x480 (silhouette forward): flat B*H*W index with 64-bit divisions 136 us; 32-bit indices 120; own /
    // right / upper id loaded separately 102 (own ids alone 52 = the 315 MB of rast lines at 6 TB/s; each extra id load
    // +25..30 us of address processing although it hits the cache); tiles of 8x8 or 64x16 pixels per wave: 155 / 135.
    __shared__ unsigned s_pair[AA_QUEUE];
    __shared__ int s_n;
    const int b = blockIdx.y, y0 = blockIdx.x * AA_ROWS, lane = threadIdx.x & 63;
    const long long ib = (long long)b * H * W;
    const float* P = pos + (size_t)b * V * 4;
    auto drain = [&]() {  // all 256 threads
        __syncthreads();
        const int n_pair = s_n;
        for (int q = threadIdx.x; q < n_pair; q += 256) {
            const unsigned e = s_pair[q];
            const int d = (int)(e & 1u), px = (int)((e >> 1) & 0xffffu), py = y0 + (int)(e >> 17);
            const long long i = ib + (long long)py * W + px, j = i + (d == 0 ? 1 : W);
            const float4 r0 = ld4(rast + i * 4), r1 = ld4(rast + j * 4);
            const int t0 = (int)r0.w - 1, t1 = (int)r1.w - 1;
            AAPair pr;
            aa_eval_pair(P, tri, opp, H, W, px, py, d, t0, t1, r0.z, r1.z, pr);
            if (!pr.valid) continue;
            const long long tg = pr.alpha > 0.f ? i : j;
            const float cdiff = COVERAGE ? coverage_of(r1, T) - coverage_of(r0, T) : 0.f;
            if (!BWD) {
                for (int c = 0; c < C; ++c)
                    atomicAdd(out + tg * C + c, pr.alpha * (COVERAGE ? cdiff : color[j * C + c] - color[i * C + c]));
            } else {
                float galpha = 0.f;
                for (int c = 0; c < C; ++c) {
                    const float g = dout[tg * C + c];
                    galpha = __fmaf_rn(g, COVERAGE ? cdiff : color[j * C + c] - color[i * C + c], galpha);
                    if (!COVERAGE && g != 0.f) {
                        atomicAdd(out + j * C + c, pr.alpha * g);
                        atomicAdd(out + i * C + c, -pr.alpha * g);
                    }
                }
                if (pr.clamped || galpha == 0.f) continue;
                float g[2][3];
                aa_pair_backward(pr, P, H, W, galpha, g);
                float* D = dpos + (size_t)b * V * 4;
                atomicAdd(D + (size_t)pr.va * 4 + 0, g[0][0]);
                atomicAdd(D + (size_t)pr.va * 4 + 1, g[0][1]);
                atomicAdd(D + (size_t)pr.va * 4 + 3, g[0][2]);
                atomicAdd(D + (size_t)pr.vb * 4 + 0, g[1][0]);
                atomicAdd(D + (size_t)pr.vb * 4 + 1, g[1][1]);
                atomicAdd(D + (size_t)pr.vb * 4 + 3, g[1][2]);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    };
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    bool need_drain = false;
    for (int x0 = 0; x0 < W; x0 += 256) {  // (workgroup-uniform)
        if (need_drain) drain();  // (uniform: decided between the two barriers at the end of the previous round)
        const int px = x0 + threadIdx.x;
        const bool inx = px < W;
        float id[AA_ROWS + 1];  // own rows and the row above the last
#pragma unroll
        for (int r = 0; r <= AA_ROWS; ++r) id[r] = (inx && y0 + r < H) ? rast[(ib + (long long)(y0 + r) * W + px) * 4 + 3] : 0.f;
#pragma unroll
        for (int r = 0; r < AA_ROWS; ++r) {
            float right = __shfl_down(id[r], 1);
            if (lane == 63 && px + 1 < W && y0 + r < H) right = rast[(ib + (long long)(y0 + r) * W + px + 1) * 4 + 3];
            if (!inx || y0 + r >= H) continue;
            const int t0 = (int)id[r] - 1;
            if (px + 1 < W) {
                const int t1 = (int)right - 1;
                if (t0 != t1 && t0 < T && t1 < T) s_pair[atomicAdd(&s_n, 1)] = ((unsigned)r << 17) | ((unsigned)px << 1);
            }
            if (y0 + r + 1 < H) {
                const int t1 = (int)id[r + 1] - 1;
                if (t0 != t1 && t0 < T && t1 < T) s_pair[atomicAdd(&s_n, 1)] = ((unsigned)r << 17) | ((unsigned)px << 1) | 1u;
            }
        }
        // every thread reads the queue length BETWEEN two barriers: no wave pushes into the next round before all waves have
        // taken the same decision (a wave that read s_n late could otherwise enter drain() -- which has barriers -- alone)
        __syncthreads();
        need_drain = s_n > AA_QUEUE - 2 * AA_ROWS * 256;
        __syncthreads();
    }
    drain();
}

static inline dim3 aa_grid(int H, int B) { return dim3((unsigned)((H + AA_ROWS - 1) / AA_ROWS), (unsigned)B); }

extern "C" int ddx_antialias_fwd(const float* color, int C, const float* rast, const float* pos, const int32_t* tri,
                                 const int32_t* opp, int B, int V, int T, int H, int W, float* out, void* stream)
{
    DDX_REQUIRE(color && rast && pos && tri && opp && out, DDX_E_NULL, "antialias_fwd: NULL pointer");
    DDX_REQUIRE(C >= 1 && C <= 64 && B >= 1 && V >= 1 && T >= 1 && H >= 1 && W >= 1, DDX_E_SHAPE, "antialias_fwd: bad shape");
    DDX_REQUIRE_FRAME(B, H, W, "antialias_fwd");
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)B * H * W;
    DDX_HIP(hipMemcpyAsync(out, color, (size_t)n * C * sizeof(float), hipMemcpyDeviceToDevice, s));
    antialias_kernel<false, false><<<aa_grid(H, B), 256, 0, s>>>(color, C, rast, pos, tri, opp, V, T, H, W, nullptr, out, nullptr);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_antialias_bwd(const float* color, int C, const float* rast, const float* pos, const int32_t* tri,
                                 const int32_t* opp, int B, int V, int T, int H, int W, const float* dout,
                                 float* dcolor, float* dpos, void* stream)
{
    DDX_REQUIRE(color && rast && pos && tri && opp && dout && dcolor && dpos, DDX_E_NULL, "antialias_bwd: NULL pointer");
    DDX_REQUIRE(C >= 1 && C <= 64 && B >= 1 && V >= 1 && T >= 1 && H >= 1 && W >= 1, DDX_E_SHAPE, "antialias_bwd: bad shape");
    DDX_REQUIRE_FRAME(B, H, W, "antialias_bwd");
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)B * H * W;
    DDX_HIP(hipMemcpyAsync(dcolor, dout, (size_t)n * C * sizeof(float), hipMemcpyDeviceToDevice, s));
    DDX_HIP(hipMemsetAsync(dpos, 0, (size_t)B * V * 4 * sizeof(float), s));
    antialias_kernel<true, false><<<aa_grid(H, B), 256, 0, s>>>(color, C, rast, pos, tri, opp, V, T, H, W, dout, dcolor, dpos);
    DDX_LAUNCH_CHECK();
    return 0;
}

// The silhouette of render_texture_batch (diffdope.py:212-214: antialias of the interpolation of a tensor of ones) without a
// colour operand.  Forward: `mask` [B,H,W,3] holds the coverage image (ddx_gbuffer_fwd's `cover`) on entry and the antialiased
// silhouette on return.  Backward: d pos [B,V,4] from d mask.  Bit-identical to ddx_antialias_fwd / _bwd on that colour.
extern "C" int ddx_silhouette_fwd(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T,
                                  int H, int W, float* mask, void* stream)
{
    return ddx_silhouette_fwd_rows(rast, pos, tri, opp, B, V, T, H, W, nullptr, mask, stream);
}

extern "C" int ddx_silhouette_fwd_rows(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T,
                                       int H, int W, const int32_t* row_range, float* mask, void* stream)
{
    return ddx_silhouette_fwd_rows_c(rast, pos, tri, opp, B, V, T, H, W, row_range, mask, 3, stream);
}

// ... on an image of `channels` equal channels (1: the single copy ddx_gbuffer_fwd_rows_c keeps; 3: the reference's layout)
extern "C" int ddx_silhouette_fwd_rows_c(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T,
                                         int H, int W, const int32_t* row_range, float* mask, int channels, void* stream)
{
    DDX_REQUIRE(rast && pos && tri && opp && mask, DDX_E_NULL, "silhouette_fwd: NULL pointer");
    DDX_REQUIRE(B >= 1 && V >= 1 && T >= 1 && H >= 1 && W >= 1, DDX_E_SHAPE, "silhouette_fwd: bad shape");
    DDX_REQUIRE(channels == 1 || channels == 3, DDX_E_SHAPE, "silhouette_fwd: channels=%d (1 or 3)", channels);
    DDX_REQUIRE_FRAME(B, H, W, "silhouette_fwd");
    antialias_kernel<false, true><<<aa_grid(H, B), 256, 0, (hipStream_t)stream>>>(nullptr, channels, rast, pos, tri, opp, V, T, H, W, nullptr, mask, nullptr, row_range);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_silhouette_bwd(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T,
                                  int H, int W, const float* dmask, float* dpos, void* stream)
{
    return ddx_silhouette_bwd_rows(rast, pos, tri, opp, B, V, T, H, W, nullptr, dmask, dpos, stream);
}

extern "C" int ddx_silhouette_bwd_rows(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T,
                                       int H, int W, const int32_t* row_range, const float* dmask, float* dpos, void* stream)
{
    return ddx_silhouette_bwd_rows_c(rast, pos, tri, opp, B, V, T, H, W, row_range, dmask, 3, dpos, stream);
}

// ... dmask [B,H,W,channels]; with one channel it is the sum of the reference's three
extern "C" int ddx_silhouette_bwd_rows_c(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T,
                                         int H, int W, const int32_t* row_range, const float* dmask, int channels, float* dpos, void* stream)
{
    DDX_REQUIRE(rast && pos && tri && opp && dmask && dpos, DDX_E_NULL, "silhouette_bwd: NULL pointer");
    DDX_REQUIRE(B >= 1 && V >= 1 && T >= 1 && H >= 1 && W >= 1, DDX_E_SHAPE, "silhouette_bwd: bad shape");
    DDX_REQUIRE(channels == 1 || channels == 3, DDX_E_SHAPE, "silhouette_bwd: channels=%d (1 or 3)", channels);
    DDX_REQUIRE_FRAME(B, H, W, "silhouette_bwd");
    hipStream_t s = (hipStream_t)stream;
    DDX_HIP(hipMemsetAsync(dpos, 0, (size_t)B * V * 4 * sizeof(float), s));
    antialias_kernel<true, true><<<aa_grid(H, B), 256, 0, s>>>(nullptr, channels, rast, pos, tri, opp, V, T, H, W, dmask, nullptr, dpos, row_range);
    DDX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Masked L1 mean per hypothesis -- the image-space part of l1_rgb_with_mask / l1_depth_with_mask / l1_mask
// (diffdope.py:547-613) for the materialising op-by-op path: out[b] = mean_i |(x[b,i] - y[i]) * m[i * m_stride]| with the
// observed image y and mask m shared by all hypotheses (m == NULL: no mask).  In torch the expression is ~10 full-frame
// element-wise / reduction kernels forward and as many backward, each a round trip through HBM of a [B,H,W,3] tensor; here
// the forward reads x once and the backward writes d x once.  Fixed two-stage reduction order: bit-reproducible.
#define ML1_CHUNKS 128
__global__ __launch_bounds__(256) void masked_l1_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ m, int m_stride, long long N,
                                                                float* __restrict__ partial)
{
    const int b = blockIdx.y, c = blockIdx.x;
    const long long per = (N + ML1_CHUNKS - 1) / ML1_CHUNKS;
    const long long i0 = (long long)c * per, i1 = i0 + per < N ? i0 + per : N;
    const float* xb = x + (size_t)b * N;
    float acc = 0.f;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float mk = m ? m[i * m_stride] : 1.0f;
        // (x is not read where the mask is exactly zero: the observed segmentation covers a few per cent of the frame, and a
        // wave whose 64 elements are all masked out issues no load for them.  0 * finite = 0 either way; a NaN / Inf render
        // under a zero mask contributes 0 here and NaN in the torch expression)
        if (mk != 0.f) acc += fabsf((xb[i] - y[i]) * mk);
    }
    acc = wave_sum(acc);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * ML1_CHUNKS + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(ML1_CHUNKS) void masked_l1_final_kernel(const float* __restrict__ partial, long long N, float* __restrict__ out)
{
    const int b = blockIdx.x;
    float v = partial[(size_t)b * ML1_CHUNKS + threadIdx.x];
    v = wave_sum(v);
    __shared__ float red[2];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[b] = __fdiv_rn(red[0] + red[1], (float)N);
}

// ... and the batch-weighted sum of the rows in the same launch (ddx_masked_l1_fwd_sum): sum_out[0] = sum_b out[b] * bw[b], the
// built-in losses' (v * learning_rates).mean() * weight (diffdope.py:534-544, :562, :580, :613) with bw = learning_rates * weight /
// B.  One workgroup; wave w takes rows w, w + 4, ... in order, the four wave sums are added pairwise: a fixed order.
__global__ __launch_bounds__(256) void masked_l1_final_sum_kernel(const float* __restrict__ partial, long long N, const float* __restrict__ bw,
                                                                  int B, float* __restrict__ out, float* __restrict__ sum_out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int b0 = wave; b0 < B; b0 += 4 * 8) {  // (eight rows' loads in flight at a time: the rows are one memory level, not sixteen)
        float lo[8], hi[8], wgt[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + k * 4;
            const bool in = b < B;
            lo[k] = in ? partial[(size_t)b * ML1_CHUNKS + lane] : 0.f;
            hi[k] = in ? partial[(size_t)b * ML1_CHUNKS + 64 + lane] : 0.f;
            wgt[k] = in ? bw[b] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + k * 4;
            if (b >= B) break;
            // (the row sum in masked_l1_final_kernel's order: lanes 0-63 and 64-127 of its workgroup, then the two halves)
            const float v = __fdiv_rn(wave_sum(lo[k]) + wave_sum(hi[k]), (float)N);
            if (lane == 0) out[b] = v;
            acc += v * wgt[k];
        }
    }
    __shared__ float red[4];
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sum_out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// the factor of row b in a backward pass: d out[b] (gout, may be NULL) plus d sum * bw[b] (gsum, may be NULL), over the row's N terms
static __device__ __forceinline__ float ml1_row_scale(const float* __restrict__ gout, const float* __restrict__ gsum,
                                                      const float* __restrict__ bw, int b, float n)
{
    float g = gout ? gout[b] : 0.f;
    if (gsum) g = gout ? g + gsum[0] * bw[b] : gsum[0] * bw[b];
    return g / n;
}

__global__ __launch_bounds__(256) void masked_l1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ m, int m_stride, const float* __restrict__ gout,
                                                            long long N, float* __restrict__ dx, const float* __restrict__ gsum = nullptr,
                                                            const float* __restrict__ bw = nullptr)
{
    const int b = blockIdx.y;
    const float scale = ml1_row_scale(gout, gsum, bw, b, (float)N);
    const float* xb = x + (size_t)b * N;
    float* db = dx + (size_t)b * N;
#pragma unroll
    for (int k = 0; k < PIX_ROUNDS; ++k) {
        const long long i = (long long)blockIdx.x * PIX_PER_WG + k * 256 + threadIdx.x;
        if (i >= N) continue;
        const float mk = m ? m[i * m_stride] : 1.0f;
        float g = 0.f;
        if (mk != 0.f) {  // (as the forward: x is only read where the mask lets it through)
            const float d = (xb[i] - y[i]) * mk;
            g = (float)((d > 0.f) - (d < 0.f)) * mk * scale;
        }
        db[i] = g;
    }
}

// Four consecutive elements per lane (N % 4 == 0, 16-byte aligned operands): 1 KB per wave instruction instead of 256 B, a
// quarter of the loop trips.  The terms of a group are added pairwise before they join the lane's sum (fixed order).
template <bool STRIDE1>
static __device__ __forceinline__ float4 mask4(const float* __restrict__ m, int m_stride, long long i4)
{
    if (!m) return make_float4(1.f, 1.f, 1.f, 1.f);
    if (STRIDE1) return ld4(m + i4 * 4);
    const long long e = i4 * 4;
    return make_float4(m[e * m_stride], m[(e + 1) * m_stride], m[(e + 2) * m_stride], m[(e + 3) * m_stride]);
}

template <bool STRIDE1>
__global__ __launch_bounds__(256) void masked_l1_partial4_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                 const float* __restrict__ m, int m_stride, long long N,
                                                                 float* __restrict__ partial)
{
    const int b = blockIdx.y, c = blockIdx.x;
    const long long N4 = N >> 2, per = (N4 + ML1_CHUNKS - 1) / ML1_CHUNKS;
    const long long i0 = (long long)c * per, i1 = i0 + per < N4 ? i0 + per : N4;
    const float* xb = x + (size_t)b * N;
    float acc = 0.f;
#pragma unroll 2
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float4 mk = mask4<STRIDE1>(m, m_stride, i);
        if (mk.x != 0.f || mk.y != 0.f || mk.z != 0.f || mk.w != 0.f) {  // (see masked_l1_partial_kernel)
            const float4 xv = ld4(xb + i * 4), yv = ld4(y + i * 4);
            acc += (fabsf((xv.x - yv.x) * mk.x) + fabsf((xv.y - yv.y) * mk.y)) + (fabsf((xv.z - yv.z) * mk.z) + fabsf((xv.w - yv.w) * mk.w));
        }
    }
    acc = wave_sum(acc);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * ML1_CHUNKS + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <bool STRIDE1>
__global__ __launch_bounds__(256) void masked_l1_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ m, int m_stride, const float* __restrict__ gout,
                                                             long long N, float* __restrict__ dx, const float* __restrict__ gsum = nullptr,
                                                             const float* __restrict__ bw = nullptr)
{
    const int b = blockIdx.y;
    const float scale = ml1_row_scale(gout, gsum, bw, b, (float)N);
    const float* xb = x + (size_t)b * N;
    float* db = dx + (size_t)b * N;
    const long long N4 = N >> 2;
    auto one = [&](float xv, float yv, float mk) {
        const float d = (xv - yv) * mk;
        return (float)((d > 0.f) - (d < 0.f)) * mk * scale;
    };
#pragma unroll
    for (int k = 0; k < PIX_ROUNDS; ++k) {
        const long long i = (long long)blockIdx.x * PIX_PER_WG + k * 256 + threadIdx.x;
        if (i >= N4) continue;
        const float4 mk = mask4<STRIDE1>(m, m_stride, i);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mk.x != 0.f || mk.y != 0.f || mk.z != 0.f || mk.w != 0.f) {
            const float4 xv = ld4(xb + i * 4), yv = ld4(y + i * 4);
            g = make_float4(mk.x != 0.f ? one(xv.x, yv.x, mk.x) : 0.f, mk.y != 0.f ? one(xv.y, yv.y, mk.y) : 0.f,
                            mk.z != 0.f ? one(xv.z, yv.z, mk.z) : 0.f, mk.w != 0.f ? one(xv.w, yv.w, mk.w) : 0.f);
        }
        *reinterpret_cast<float4*>(db + i * 4) = g;
    }
}

// x with ONE channel against an observed image of three (l1_mask, diffdope.py:583-613: the rendered silhouette's three channels are
// one number, kept once -- ddx_gbuffer_fwd_rows_c --, the observed segmentation's are not): out[b] = mean over (i, c) of
// |(x[b,i] - y[i,c]) * m[i,c]|, d x[b,i] = the sum over c.  Four pixels per lane when P % 4 == 0 and the operands are 16-byte
// aligned (VEC), one otherwise; per pixel (|d0| + |d1|) + |d2|, the pixels of a group added pairwise: a fixed order.
template <bool VEC>
__global__ __launch_bounds__(256) void masked_l1_bc3_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                    const float* __restrict__ m, long long P, float* __restrict__ partial)
{
    const int b = blockIdx.y, c = blockIdx.x;
    const long long G = VEC ? P >> 2 : P, per = (G + ML1_CHUNKS - 1) / ML1_CHUNKS;
    const long long i0 = (long long)c * per, i1 = i0 + per < G ? i0 + per : G;
    const float* xb = x + (size_t)b * P;
    auto px = [&](float xv, const float* yy, const float* mm) {
        const float m0 = m ? mm[0] : 1.f, m1 = m ? mm[1] : 1.f, m2 = m ? mm[2] : 1.f;
        return (fabsf((xv - yy[0]) * m0) + fabsf((xv - yy[1]) * m1)) + fabsf((xv - yy[2]) * m2);
    };
    float acc = 0.f;
#pragma unroll 2
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        if (VEC) {
            float yy[12], mm[12];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 v = ld4(y + i * 12 + k * 4);
                yy[k * 4] = v.x; yy[k * 4 + 1] = v.y; yy[k * 4 + 2] = v.z; yy[k * 4 + 3] = v.w;
                const float4 w = m ? ld4(m + i * 12 + k * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
                mm[k * 4] = w.x; mm[k * 4 + 1] = w.y; mm[k * 4 + 2] = w.z; mm[k * 4 + 3] = w.w;
            }
            bool any = false;
#pragma unroll
            for (int k = 0; k < 12; ++k) any |= mm[k] != 0.f;
            if (any) {  // (as masked_l1_partial_kernel: x is only read where the mask lets something through)
                const float4 xv = ld4(xb + i * 4);
                acc += (px(xv.x, yy, mm) + px(xv.y, yy + 3, mm + 3)) + (px(xv.z, yy + 6, mm + 6) + px(xv.w, yy + 9, mm + 9));
            }
        } else {
            const float* mm = m ? m + i * 3 : nullptr;
            if (!m || mm[0] != 0.f || mm[1] != 0.f || mm[2] != 0.f) acc += px(xb[i], y + i * 3, mm);
        }
    }
    acc = wave_sum(acc);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * ML1_CHUNKS + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <bool VEC>
__global__ __launch_bounds__(256) void masked_l1_bc3_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ m, const float* __restrict__ gout, long long P,
                                                                float* __restrict__ dx, const float* __restrict__ gsum = nullptr,
                                                                const float* __restrict__ bw = nullptr)
{
    const int b = blockIdx.y;
    const float scale = ml1_row_scale(gout, gsum, bw, b, (float)(P * 3));
    const float* xb = x + (size_t)b * P;
    float* db = dx + (size_t)b * P;
    const long long G = VEC ? P >> 2 : P;
    auto px = [&](float xv, const float* yy, const float* mm) {
        float g = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float mk = m ? mm[c] : 1.f;
            const float d = (xv - yy[c]) * mk;
            g += (float)((d > 0.f) - (d < 0.f)) * mk * scale;  // (a zero mask adds +-0)
        }
        return g;
    };
#pragma unroll
    for (int k = 0; k < PIX_ROUNDS; ++k) {
        const long long i = (long long)blockIdx.x * PIX_PER_WG + k * 256 + threadIdx.x;
        if (i >= G) continue;
        if (VEC) {
            float yy[12], mm[12];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4 v = ld4(y + i * 12 + q * 4);
                yy[q * 4] = v.x; yy[q * 4 + 1] = v.y; yy[q * 4 + 2] = v.z; yy[q * 4 + 3] = v.w;
                const float4 w = m ? ld4(m + i * 12 + q * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
                mm[q * 4] = w.x; mm[q * 4 + 1] = w.y; mm[q * 4 + 2] = w.z; mm[q * 4 + 3] = w.w;
            }
            bool any = false;
#pragma unroll
            for (int q = 0; q < 12; ++q) any |= mm[q] != 0.f;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (any) {
                const float4 xv = ld4(xb + i * 4);
                g = make_float4(px(xv.x, yy, mm), px(xv.y, yy + 3, mm + 3), px(xv.z, yy + 6, mm + 6), px(xv.w, yy + 9, mm + 9));
            }
            *reinterpret_cast<float4*>(db + i * 4) = g;
        } else {
            const float* mm = m ? m + i * 3 : nullptr;
            db[i] = (!m || mm[0] != 0.f || mm[1] != 0.f || mm[2] != 0.f) ? px(xb[i], y + i * 3, mm) : 0.f;
        }
    }
}

static inline bool ml1_vec_ok(const void* x, const void* y, const void* m, int m_stride, const void* dx, long long N)
{
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return (N & 3) == 0 && al(x) && al(y) && (!dx || al(dx)) && (!m || m_stride != 1 || al(m));
}

extern "C" int ddx_masked_l1_fwd(const float* x, const float* y, const float* m, int m_stride, int B, long long N, float* partial,
                                 float* out, void* stream)
{
    DDX_REQUIRE(x && y && partial && out, DDX_E_NULL, "masked_l1_fwd: NULL pointer");
    DDX_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && m_stride >= 1, DDX_E_SHAPE, "masked_l1_fwd: bad shape B=%d N=%lld stride=%d", B, N, m_stride);
    if (ml1_vec_ok(x, y, m, m_stride, nullptr, N)) {
        if (!m || m_stride == 1) masked_l1_partial4_kernel<true><<<dim3(ML1_CHUNKS, B), 256, 0, (hipStream_t)stream>>>(x, y, m, m_stride, N, partial);
        else masked_l1_partial4_kernel<false><<<dim3(ML1_CHUNKS, B), 256, 0, (hipStream_t)stream>>>(x, y, m, m_stride, N, partial);
    } else
        masked_l1_partial_kernel<<<dim3(ML1_CHUNKS, B), 256, 0, (hipStream_t)stream>>>(x, y, m, m_stride, N, partial);
    masked_l1_final_kernel<<<B, ML1_CHUNKS, 0, (hipStream_t)stream>>>(partial, N, out);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_masked_l1_bwd(const float* x, const float* y, const float* m, int m_stride, const float* gout, int B, long long N,
                                 float* dx, void* stream)
{
    DDX_REQUIRE(x && y && gout && dx, DDX_E_NULL, "masked_l1_bwd: NULL pointer");
    DDX_REQUIRE(B >= 1 && N >= 1 && m_stride >= 1, DDX_E_SHAPE, "masked_l1_bwd: bad shape");
    DDX_REQUIRE(B <= 65535 && N < (1ll << 40), DDX_E_SHAPE, "masked_l1_bwd: B=%d N=%lld exceed the launch limits", B, N);
    if (ml1_vec_ok(x, y, m, m_stride, dx, N)) {
        if (!m || m_stride == 1) masked_l1_bwd4_kernel<true><<<pix_grid2(N >> 2, B), 256, 0, (hipStream_t)stream>>>(x, y, m, m_stride, gout, N, dx);
        else masked_l1_bwd4_kernel<false><<<pix_grid2(N >> 2, B), 256, 0, (hipStream_t)stream>>>(x, y, m, m_stride, gout, N, dx);
    } else
        masked_l1_bwd_kernel<<<pix_grid2(N, B), 256, 0, (hipStream_t)stream>>>(x, y, m, m_stride, gout, N, dx);
    DDX_LAUNCH_CHECK();
    return 0;
}

// ... x [B,P] with one channel against y, m [P,3] (see masked_l1_bc3_partial_kernel); out[b] = the mean over the 3 P terms
extern "C" int ddx_masked_l1_bc3_fwd(const float* x, const float* y, const float* m, int B, long long P, float* partial, float* out,
                                     void* stream)
{
    DDX_REQUIRE(x && y && partial && out, DDX_E_NULL, "masked_l1_bc3_fwd: NULL pointer");
    DDX_REQUIRE(B >= 1 && B <= 65535 && P >= 1 && P < (1ll << 38), DDX_E_SHAPE, "masked_l1_bc3_fwd: bad shape B=%d P=%lld", B, P);
    if (ml1_vec_ok(x, y, m, 1, nullptr, P)) masked_l1_bc3_partial_kernel<true><<<dim3(ML1_CHUNKS, B), 256, 0, (hipStream_t)stream>>>(x, y, m, P, partial);
    else masked_l1_bc3_partial_kernel<false><<<dim3(ML1_CHUNKS, B), 256, 0, (hipStream_t)stream>>>(x, y, m, P, partial);
    masked_l1_final_kernel<<<B, ML1_CHUNKS, 0, (hipStream_t)stream>>>(partial, P * 3, out);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_masked_l1_bc3_bwd(const float* x, const float* y, const float* m, const float* gout, int B, long long P, float* dx,
                                     void* stream)
{
    DDX_REQUIRE(x && y && gout && dx, DDX_E_NULL, "masked_l1_bc3_bwd: NULL pointer");
    DDX_REQUIRE(B >= 1 && B <= 65535 && P >= 1 && P < (1ll << 38), DDX_E_SHAPE, "masked_l1_bc3_bwd: bad shape B=%d P=%lld", B, P);
    if (ml1_vec_ok(x, y, m, 1, dx, P)) masked_l1_bc3_bwd_kernel<true><<<pix_grid2(P >> 2, B), 256, 0, (hipStream_t)stream>>>(x, y, m, gout, P, dx);
    else masked_l1_bc3_bwd_kernel<false><<<pix_grid2(P, B), 256, 0, (hipStream_t)stream>>>(x, y, m, gout, P, dx);
    DDX_LAUNCH_CHECK();
    return 0;
}

// The loss term of a built-in loss in two launches each way (see masked_l1_final_sum_kernel): out [B] as ddx_masked_l1_fwd (bc3 == 0: x
// [B,N], y [N], m with m_stride) or ddx_masked_l1_bc3_fwd (bc3 != 0: x [B,N], y and m [N,3]) -- the same bits --, and sum_out[0] =
// sum_b out[b] * bw[b].  Backward: d x from d out (gout [B] or NULL) and d sum (gsum [1] or NULL; then bw [B]) in ONE launch.
extern "C" int ddx_masked_l1_fwd_sum(const float* x, const float* y, const float* m, int m_stride, int bc3, int B, long long N, const float* bw,
                                     float* partial, float* out, float* sum_out, void* stream)
{
    DDX_REQUIRE(x && y && partial && out && bw && sum_out, DDX_E_NULL, "masked_l1_fwd_sum: NULL pointer");
    DDX_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && N < (1ll << 38) && m_stride >= 1, DDX_E_SHAPE, "masked_l1_fwd_sum: bad shape B=%d N=%lld stride=%d", B, N, m_stride);
    hipStream_t s = (hipStream_t)stream;
    if (bc3) {
        if (ml1_vec_ok(x, y, m, 1, nullptr, N)) masked_l1_bc3_partial_kernel<true><<<dim3(ML1_CHUNKS, B), 256, 0, s>>>(x, y, m, N, partial);
        else masked_l1_bc3_partial_kernel<false><<<dim3(ML1_CHUNKS, B), 256, 0, s>>>(x, y, m, N, partial);
    } else if (ml1_vec_ok(x, y, m, m_stride, nullptr, N)) {
        if (!m || m_stride == 1) masked_l1_partial4_kernel<true><<<dim3(ML1_CHUNKS, B), 256, 0, s>>>(x, y, m, m_stride, N, partial);
        else masked_l1_partial4_kernel<false><<<dim3(ML1_CHUNKS, B), 256, 0, s>>>(x, y, m, m_stride, N, partial);
    } else
        masked_l1_partial_kernel<<<dim3(ML1_CHUNKS, B), 256, 0, s>>>(x, y, m, m_stride, N, partial);
    masked_l1_final_sum_kernel<<<1, 256, 0, s>>>(partial, bc3 ? N * 3 : N, bw, B, out, sum_out);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_masked_l1_bwd_sum(const float* x, const float* y, const float* m, int m_stride, int bc3, const float* gout, const float* gsum,
                                     const float* bw, int B, long long N, float* dx, void* stream)
{
    DDX_REQUIRE(x && y && dx && (gout || gsum) && (!gsum || bw), DDX_E_NULL, "masked_l1_bwd_sum: NULL pointer");
    DDX_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && N < (1ll << 38) && m_stride >= 1, DDX_E_SHAPE, "masked_l1_bwd_sum: bad shape B=%d N=%lld", B, N);
    hipStream_t s = (hipStream_t)stream;
    if (bc3) {
        if (ml1_vec_ok(x, y, m, 1, dx, N)) masked_l1_bc3_bwd_kernel<true><<<pix_grid2(N >> 2, B), 256, 0, s>>>(x, y, m, gout, N, dx, gsum, bw);
        else masked_l1_bc3_bwd_kernel<false><<<pix_grid2(N, B), 256, 0, s>>>(x, y, m, gout, N, dx, gsum, bw);
    } else if (ml1_vec_ok(x, y, m, m_stride, dx, N)) {
        if (!m || m_stride == 1) masked_l1_bwd4_kernel<true><<<pix_grid2(N >> 2, B), 256, 0, s>>>(x, y, m, m_stride, gout, N, dx, gsum, bw);
        else masked_l1_bwd4_kernel<false><<<pix_grid2(N >> 2, B), 256, 0, s>>>(x, y, m, m_stride, gout, N, dx, gsum, bw);
    } else
        masked_l1_bwd_kernel<<<pix_grid2(N, B), 256, 0, s>>>(x, y, m, m_stride, gout, N, dx, gsum, bw);
    DDX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// gbuffer: the per-pixel middle of render_texture_batch (diffdope/diffdope.py:203-231) as ONE forward and ONE backward pass
// over the frame -- interpolate(pos) -> pose transform -> depth; interpolate(uv) -> texture(linear) (or interpolate(vertex
// colour)) -> * clamp(id, 0, 1) -> rgb; interpolate(ones) -> coverage (the input of antialias) -- instead of three
// interpolate, one texture, two xfm launches and ~15 framework elementwise kernels each way, every one a pass over 80-300 MB.
// Same arithmetic, in the same order, as the individual ops above (the tests hold both against the oracle).  The backward
// turns d rgb / d depth straight into d clip (x, y, w of the pixel's triangle: what rasterize_bwd would produce from the
// (u, v) gradients the individual ops hand it; a clamped barycentric passes none) and d mtx (row 2: depth reads the pose).
// Gradients with respect to the mesh attributes or the texture are not produced: callers that need them take the op-by-op ops.
template <bool TEXTURED>
__global__ __launch_bounds__(256) void gbuffer_fwd_kernel(const float* __restrict__ rast, const float* __restrict__ mtx,
                                                          const float* __restrict__ pos, const int* __restrict__ tri,
                                                          const float* __restrict__ uv, const float* __restrict__ tex, int Th, int Tw,
                                                          const float* __restrict__ vcol, int V, int T, int HW,
                                                          float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ cover,
                                                          const int* __restrict__ row_range, int W, int cover_c)
{
    const int b = blockIdx.y;
    const float* M = mtx + (size_t)b * 16;
    // (row_range: pixels outside the rows of the hypothesis' active tiles are background by construction -- written without
    // reading `rast`, which the restricted emit has not filled there)
    const int plo = row_range ? row_range[b * 2] * W : 0, phi = row_range ? (row_range[b * 2 + 1] + 1) * W : HW;
#pragma unroll
    for (int rd = 0; rd < PIX_ROUNDS; ++rd) {
        const int p = blockIdx.x * PIX_PER_WG + rd * 256 + threadIdx.x;
        if (p >= HW) continue;
        const long long i = (long long)b * HW + p;
        const float4 r = (p >= plo && p < phi) ? ld4(rast + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int t = (int)r.w - 1;
        float col[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f}, cv = 0.f;
        int i0 = 0, i1 = 0, i2 = 0;
        bool in = t >= 0 && t < T;
        if (in) {
            i0 = tri[t * 3 + 0]; i1 = tri[t * 3 + 1]; i2 = tri[t * 3 + 2];
            in = (unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V;
        }
        if (in) {
            const float u = r.x, v = r.y, w2 = (1.0f - u) - v;
#pragma unroll
            for (int c = 0; c < 3; ++c) gb[c] = __fmaf_rn(w2, pos[(size_t)i2 * 3 + c], __fmaf_rn(v, pos[(size_t)i1 * 3 + c], u * pos[(size_t)i0 * 3 + c]));
            cv = __fmaf_rn(w2, 1.0f, __fmaf_rn(v, 1.0f, u * 1.0f));  // interpolate of a tensor of ones (diffdope.py:212)
            if (TEXTURED) {
                const float tu = __fmaf_rn(w2, uv[(size_t)i2 * 2], __fmaf_rn(v, uv[(size_t)i1 * 2], u * uv[(size_t)i0 * 2]));
                const float tv = __fmaf_rn(w2, uv[(size_t)i2 * 2 + 1], __fmaf_rn(v, uv[(size_t)i1 * 2 + 1], u * uv[(size_t)i0 * 2 + 1]));
                TexelSetup s;
                tex_setup(tu, tv, Th, Tw, s);
                const float *t00 = tex + ((size_t)s.y0 * Tw + s.x0) * 3, *t10 = tex + ((size_t)s.y0 * Tw + s.x1) * 3,
                            *t01 = tex + ((size_t)s.y1 * Tw + s.x0) * 3, *t11 = tex + ((size_t)s.y1 * Tw + s.x1) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float a = __fmaf_rn(s.fx, t10[c] - t00[c], t00[c]);
                    const float bq = __fmaf_rn(s.fx, t11[c] - t01[c], t01[c]);
                    col[c] = __fmaf_rn(s.fy, bq - a, a);
                }
            } else if (rgb) {  // (rgb == NULL: the caller asked for depth and coverage only)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    col[c] = __fmaf_rn(w2, vcol[(size_t)i2 * 3 + c], __fmaf_rn(v, vcol[(size_t)i1 * 3 + c], u * vcol[(size_t)i0 * 3 + c]));
            }
        }
        // depth = -(mtx . [gb; 1])_z, the k-ordered fma chain of xfm_points (a background pixel interpolates to the origin: -mtx[2][3])
        if (depth) {
            float zc = __fmaf_rn(M[8], gb[0], 0.f);
            zc = __fmaf_rn(M[9], gb[1], zc);
            zc = __fmaf_rn(M[10], gb[2], zc);
            zc = __fmaf_rn(M[11], 1.0f, zc);
            depth[i] = -zc;
        }
        const float k = r.w < 0.f ? 0.f : (r.w > 1.f ? 1.f : r.w);  // clamp(rast[..., -1:], 0, 1) (diffdope.py:228,231)
        if (rgb) { rgb[i * 3 + 0] = col[0] * k; rgb[i * 3 + 1] = col[1] * k; rgb[i * 3 + 2] = col[2] * k; }
        // (the three channels of interpolate(ones) are one number: cover_c == 1 keeps one copy of it, see ddx_gbuffer_fwd_rows_c)
        if (cover_c == 1) cover[i] = cv;
        else { cover[i * 3 + 0] = cv; cover[i * 3 + 1] = cv; cover[i * 3 + 2] = cv; }
    }
}

template <bool TEXTURED>
__global__ __launch_bounds__(256) void gbuffer_bwd_kernel(const float* __restrict__ rast, const float* __restrict__ clip,
                                                          const float* __restrict__ mtx, const float* __restrict__ pos,
                                                          const int* __restrict__ tri, const float* __restrict__ uv,
                                                          const float* __restrict__ tex, int Th, int Tw, const float* __restrict__ vcol,
                                                          int V, int T, int H, int W, const float* __restrict__ drgb,
                                                          const float* __restrict__ ddepth, float* __restrict__ dclip,
                                                          float* __restrict__ dmtx, int compat, const int* __restrict__ row_range)
{
    const int HW = H * W;
    const int plo = row_range ? row_range[blockIdx.y * 2] * W : 0, phi = row_range ? (row_range[blockIdx.y * 2 + 1] + 1) * W : HW;
    __shared__ float s_dm[4][4];
    // the pixels of this workgroup belong to ONE hypothesis (blockIdx.y): their contributions to d mtx[b][2][:] are summed over
    // the workgroup and added with four atomics at the end
    float dm[4] = {0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bb = blockIdx.y;
    float4 rr[PIX_ROUNDS];  // (all rounds' visibility records in flight before the first is used)
#pragma unroll
    for (int rd = 0; rd < PIX_ROUNDS; ++rd) {
        const int p = blockIdx.x * PIX_PER_WG + rd * 256 + threadIdx.x;
        rr[rd] = (p < HW && p >= plo && p < phi) ? ld4(rast + ((long long)bb * HW + p) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);  // (outside the active rows: background)
    }
#pragma unroll
    for (int rd = 0; rd < PIX_ROUNDS; ++rd) {
        const int p = blockIdx.x * PIX_PER_WG + rd * 256 + threadIdx.x;
        const bool mine = p < HW;
        const long long i = (long long)bb * HW + p;
        if (mine) {
            const float4 r = rr[rd];
            const int t = (int)r.w - 1;
            const float gd = ddepth ? ddepth[i] : 0.f;
            const float* M = mtx + (size_t)bb * 16;
            int i0 = 0, i1 = 0, i2 = 0;
            bool in = t >= 0 && t < T;
            if (in) {
                i0 = tri[t * 3 + 0]; i1 = tri[t * 3 + 1]; i2 = tri[t * 3 + 2];
                in = (unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V;
            }
            if (!in) {
                dm[3] += -gd;  // depth of a background pixel = -mtx[2][3]
            } else {
                const float u = r.x, v = r.y, w2 = (1.0f - u) - v;
                float p0[3], p1[3], p2[3], gb[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    p0[c] = pos[(size_t)i0 * 3 + c]; p1[c] = pos[(size_t)i1 * 3 + c]; p2[c] = pos[(size_t)i2 * 3 + c];
                    gb[c] = __fmaf_rn(w2, p2[c], __fmaf_rn(v, p1[c], u * p0[c]));
                }
                float gu = 0.f, gv = 0.f;
                // depth = -(M2 . [gb;1]):  d/d M2 = -g [gb;1],  d/d gb = -g M2[0..2]
                dm[0] += -gd * gb[0]; dm[1] += -gd * gb[1]; dm[2] += -gd * gb[2]; dm[3] += -gd;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float ggb = -gd * M[8 + c];
                    gu = __fmaf_rn(ggb, p0[c] - p2[c], gu);
                    gv = __fmaf_rn(ggb, p1[c] - p2[c], gv);
                }
                if (drgb) {
                    const float k = r.w < 0.f ? 0.f : (r.w > 1.f ? 1.f : r.w);
                    const float g[3] = {drgb[i * 3 + 0] * k, drgb[i * 3 + 1] * k, drgb[i * 3 + 2] * k};
                    if (TEXTURED) {
                        const float a0x = uv[(size_t)i0 * 2], a0y = uv[(size_t)i0 * 2 + 1], a1x = uv[(size_t)i1 * 2], a1y = uv[(size_t)i1 * 2 + 1],
                                    a2x = uv[(size_t)i2 * 2], a2y = uv[(size_t)i2 * 2 + 1];
                        const float tu = __fmaf_rn(w2, a2x, __fmaf_rn(v, a1x, u * a0x)), tv = __fmaf_rn(w2, a2y, __fmaf_rn(v, a1y, u * a0y));
                        TexelSetup s;
                        tex_setup(tu, tv, Th, Tw, s);
                        const float *t00 = tex + ((size_t)s.y0 * Tw + s.x0) * 3, *t10 = tex + ((size_t)s.y0 * Tw + s.x1) * 3,
                                    *t01 = tex + ((size_t)s.y1 * Tw + s.x0) * 3, *t11 = tex + ((size_t)s.y1 * Tw + s.x1) * 3;
                        float gtu = 0.f, gtv = 0.f;  // d loss / d (tu, tv), as texture_bwd_kernel
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float c00 = t00[c], c10 = t10[c], c01 = t01[c], c11 = t11[c];
                            gtu = __fmaf_rn(g[c], __fmaf_rn(s.fy, (c11 - c01) - (c10 - c00), c10 - c00), gtu);
                            gtv = __fmaf_rn(g[c], __fmaf_rn(s.fx, (c11 - c10) - (c01 - c00), c01 - c00), gtv);
                        }
                        gtu *= (float)Tw; gtv *= (float)Th;
                        gu = __fmaf_rn(gtu, a0x - a2x, gu); gu = __fmaf_rn(gtv, a0y - a2y, gu);
                        gv = __fmaf_rn(gtu, a1x - a2x, gv); gv = __fmaf_rn(gtv, a1y - a2y, gv);
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float c0 = vcol[(size_t)i0 * 3 + c], c1 = vcol[(size_t)i1 * 3 + c], c2 = vcol[(size_t)i2 * 3 + c];
                            gu = __fmaf_rn(g[c], c0 - c2, gu);
                            gv = __fmaf_rn(g[c], c1 - c2, gv);
                        }
                    }
                }
                if (gu != 0.f || gv != 0.f) {
                    const int py = p / W, px = p - py * W;
                    const float* P = clip + (size_t)bb * V * 4;
                    const float4 c0 = ld4(P + (size_t)i0 * 4), c1 = ld4(P + (size_t)i1 * 4), c2 = ld4(P + (size_t)i2 * 4);
                    Bary bc;
                    if (pixel_bary(c0, c1, c2, px, py, H, W, bc)) {
                        float gx[3], gy[3], gw[3];
                        bary_backward(bc, gu, gv, gx, gy, gw, (compat & DDX_COMPAT_UNCLAMPED_BARY_GRAD) != 0);
                        float* D = dclip + (size_t)bb * V * 4;
                        const int vi[3] = {i0, i1, i2};
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            atomicAdd(D + (size_t)vi[q] * 4 + 0, gx[q]);
                            atomicAdd(D + (size_t)vi[q] * 4 + 1, gy[q]);
                            atomicAdd(D + (size_t)vi[q] * 4 + 3, gw[q]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float sum = wave_sum(dm[c]);
        if (lane == 0) s_dm[wave][c] = sum;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float tot = (s_dm[0][threadIdx.x] + s_dm[1][threadIdx.x]) + (s_dm[2][threadIdx.x] + s_dm[3][threadIdx.x]);
        if (tot != 0.f) atomicAdd(dmtx + (size_t)bb * 16 + 8 + threadIdx.x, tot);
    }
}

extern "C" int ddx_gbuffer_fwd(const float* rast, const float* mtx, const float* pos, const int32_t* tri, const float* uv,
                               const float* tex, int Th, int Tw, const float* vtx_color, int B, int V, int T, int H, int W, float* rgb,
                               float* depth, float* cover, void* stream)
{
    return ddx_gbuffer_fwd_rows(rast, mtx, pos, tri, uv, tex, Th, Tw, vtx_color, B, V, T, H, W, nullptr, rgb, depth, cover, stream);
}

extern "C" int ddx_gbuffer_fwd_rows(const float* rast, const float* mtx, const float* pos, const int32_t* tri, const float* uv,
                                    const float* tex, int Th, int Tw, const float* vtx_color, int B, int V, int T, int H, int W,
                                    const int32_t* row_range, float* rgb, float* depth, float* cover, void* stream)
{
    DDX_REQUIRE(rgb && depth, DDX_E_NULL, "gbuffer_fwd: NULL pointer");
    return ddx_gbuffer_fwd_rows_c(rast, mtx, pos, tri, uv, tex, Th, Tw, vtx_color, B, V, T, H, W, row_range, rgb, depth, cover, 3, stream);
}

extern "C" int ddx_gbuffer_fwd_rows_c(const float* rast, const float* mtx, const float* pos, const int32_t* tri, const float* uv,
                                      const float* tex, int Th, int Tw, const float* vtx_color, int B, int V, int T, int H, int W,
                                      const int32_t* row_range, float* rgb, float* depth, float* cover, int cover_channels, void* stream)
{
    DDX_REQUIRE(rast && mtx && pos && tri && cover, DDX_E_NULL, "gbuffer_fwd: NULL pointer");
    DDX_REQUIRE(!rgb || (uv && tex && Th >= 1 && Tw >= 1) || vtx_color, DDX_E_NULL, "gbuffer_fwd: needs (uv, tex) or vtx_color");
    DDX_REQUIRE(B >= 1 && V >= 1 && T >= 1 && H >= 1 && W >= 1, DDX_E_SHAPE, "gbuffer_fwd: bad shape");
    DDX_REQUIRE(cover_channels == 1 || cover_channels == 3, DDX_E_SHAPE, "gbuffer_fwd: cover_channels=%d (1 or 3)", cover_channels);
    DDX_REQUIRE(((uintptr_t)rast & 15) == 0, DDX_E_ALIGN, "gbuffer_fwd: rast must be 16-byte aligned");
    DDX_REQUIRE_FRAME(B, H, W, "gbuffer_fwd");
    hipStream_t s = (hipStream_t)stream;
    if (rgb && uv && tex) gbuffer_fwd_kernel<true><<<pix_grid2((long long)H * W, B), 256, 0, s>>>(rast, mtx, pos, tri, uv, tex, Th, Tw, nullptr, V, T, H * W, rgb, depth, cover, row_range, W, cover_channels);
    else gbuffer_fwd_kernel<false><<<pix_grid2((long long)H * W, B), 256, 0, s>>>(rast, mtx, pos, tri, nullptr, nullptr, 0, 0, vtx_color, V, T, H * W, rgb, depth, cover, row_range, W, cover_channels);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_gbuffer_bwd(const float* rast, const float* clip, const float* mtx, const float* pos, const int32_t* tri,
                               const float* uv, const float* tex, int Th, int Tw, const float* vtx_color, int B, int V, int T, int H, int W,
                               const float* drgb, const float* ddepth, float* dclip, float* dmtx, void* stream)
{
    return ddx_gbuffer_bwd_rows(rast, clip, mtx, pos, tri, uv, tex, Th, Tw, vtx_color, B, V, T, H, W, nullptr, drgb, ddepth, dclip, dmtx, stream);
}

extern "C" int ddx_gbuffer_bwd_rows(const float* rast, const float* clip, const float* mtx, const float* pos, const int32_t* tri,
                                    const float* uv, const float* tex, int Th, int Tw, const float* vtx_color, int B, int V, int T, int H, int W,
                                    const int32_t* row_range, const float* drgb, const float* ddepth, float* dclip, float* dmtx, void* stream)
{
    DDX_REQUIRE(rast && clip && mtx && pos && tri && dclip && dmtx, DDX_E_NULL, "gbuffer_bwd: NULL pointer");
    DDX_REQUIRE((uv && tex && Th >= 1 && Tw >= 1) || vtx_color, DDX_E_NULL, "gbuffer_bwd: needs (uv, tex) or vtx_color");
    DDX_REQUIRE(B >= 1 && V >= 1 && T >= 1 && H >= 1 && W >= 1, DDX_E_SHAPE, "gbuffer_bwd: bad shape");
    DDX_REQUIRE(((uintptr_t)rast & 15) == 0 && ((uintptr_t)clip & 15) == 0, DDX_E_ALIGN, "gbuffer_bwd: rast / clip must be 16-byte aligned");
    DDX_REQUIRE_FRAME(B, H, W, "gbuffer_bwd");
    hipStream_t s = (hipStream_t)stream;
    DDX_HIP(hipMemsetAsync(dclip, 0, (size_t)B * V * 4 * sizeof(float), s));
    DDX_HIP(hipMemsetAsync(dmtx, 0, (size_t)B * 16 * sizeof(float), s));
    if (uv && tex) gbuffer_bwd_kernel<true><<<pix_grid2((long long)H * W, B), 256, 0, s>>>(rast, clip, mtx, pos, tri, uv, tex, Th, Tw, nullptr, V, T, H, W, drgb, ddepth, dclip, dmtx, ddx_compat_flags(), row_range);
    else gbuffer_bwd_kernel<false><<<pix_grid2((long long)H * W, B), 256, 0, s>>>(rast, clip, mtx, pos, tri, nullptr, nullptr, 0, 0, vtx_color, V, T, H, W, drgb, ddepth, dclip, dmtx, ddx_compat_flags(), row_range);
    DDX_LAUNCH_CHECK();
    return 0;
}
