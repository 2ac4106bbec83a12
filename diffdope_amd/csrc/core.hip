// core.hip -- version, thread-local error string, host-side mesh topology.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "ddx_common.h"

static thread_local char g_err[512] = "";

void ddx_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int g_compat = 0;
int ddx_compat_flags(void) { return g_compat; }
extern "C" int ddx_set_compat(int flags)
{
    const int old = g_compat;
    g_compat = flags;
    return old;
}

extern "C" int ddx_version(void) { return DDX_VERSION; }
extern "C" const char* ddx_last_error(void) { return g_err; }

// Edge -> opposite-vertex topology for antialias (nvdiffrast rebuilds an edge hash on the GPU on
// every call, diffdope.py:214 passes no topology_hash; here it is built once per mesh on the host).
extern "C" int ddx_topology_build(const int32_t* tri, int T, int32_t* opp)
{
    DDX_REQUIRE(tri && opp, DDX_E_NULL, "topology_build: NULL pointer");
    DDX_REQUIRE(T >= 1, DDX_E_SHAPE, "topology_build: T=%d", T);
    struct E { int32_t a, b, t, k; };
    std::vector<E> e((size_t)T * 3);
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k) {
            int32_t a = tri[t * 3 + (k + 1) % 3], b = tri[t * 3 + (k + 2) % 3];
            if (a > b) std::swap(a, b);
            e[(size_t)t * 3 + k] = {a, b, t, k};
        }
    std::sort(e.begin(), e.end(), [](const E& x, const E& y) {
        if (x.a != y.a) return x.a < y.a;
        if (x.b != y.b) return x.b < y.b;
        if (x.t != y.t) return x.t < y.t;
        return x.k < y.k;
    });
    std::fill(opp, opp + (size_t)T * 3, -1);
    for (size_t i = 0; i < e.size();) {
        size_t j = i;
        while (j < e.size() && e[j].a == e[i].a && e[j].b == e[i].b) ++j;
        for (size_t m = i; m < j; ++m)
            for (size_t n = i; n < j; ++n) {
                if (e[n].t == e[m].t) continue;
                opp[(size_t)e[m].t * 3 + e[m].k] = tri[(size_t)e[n].t * 3 + e[n].k];
                break;
            }
        i = j;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// matrix_batch_44_from_position_quat (diffdope/diffdope.py:46-89) as one kernel each way: the reference builds the matrix
// from ~30 framework ops (and autograd adds ~60 for the backward).  q is used as given (the caller normalises it,
// diffdope.py:1091), same row formulas as :57-80.
__global__ void pose_matrix_fwd_kernel(const float* __restrict__ q, const float* __restrict__ p, int B, float* __restrict__ M)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float x = q[b * 4 + 0], y = q[b * 4 + 1], z = q[b * 4 + 2], w = q[b * 4 + 3];
    float* m = M + (size_t)b * 16;
    m[0] = 1.f - 2.f * y * y - 2.f * z * z; m[1] = 2.f * x * y - 2.f * z * w; m[2] = 2.f * x * z + 2.f * y * w; m[3] = p[b * 3 + 0];
    m[4] = 2.f * x * y + 2.f * z * w; m[5] = 1.f - 2.f * x * x - 2.f * z * z; m[6] = 2.f * y * z - 2.f * x * w; m[7] = p[b * 3 + 1];
    m[8] = 2.f * x * z - 2.f * y * w; m[9] = 2.f * y * z + 2.f * x * w; m[10] = 1.f - 2.f * x * x - 2.f * y * y; m[11] = p[b * 3 + 2];
    m[12] = 0.f; m[13] = 0.f; m[14] = 0.f; m[15] = 1.f;
}

__global__ void pose_matrix_bwd_kernel(const float* __restrict__ q, const float* __restrict__ dM, int B, float* __restrict__ dq,
                                       float* __restrict__ dp)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float x = q[b * 4 + 0], y = q[b * 4 + 1], z = q[b * 4 + 2], w = q[b * 4 + 3];
    const float* G = dM + (size_t)b * 16;
    dq[b * 4 + 0] = G[1] * 2 * y + G[2] * 2 * z + G[4] * 2 * y + G[5] * (-4 * x) + G[6] * (-2 * w) + G[8] * 2 * z + G[9] * 2 * w + G[10] * (-4 * x);
    dq[b * 4 + 1] = G[0] * (-4 * y) + G[1] * 2 * x + G[2] * 2 * w + G[4] * 2 * x + G[6] * 2 * z + G[8] * (-2 * w) + G[9] * 2 * z + G[10] * (-4 * y);
    dq[b * 4 + 2] = G[0] * (-4 * z) + G[1] * (-2 * w) + G[2] * 2 * x + G[4] * 2 * w + G[5] * (-4 * z) + G[6] * 2 * y + G[8] * 2 * x + G[9] * 2 * y;
    dq[b * 4 + 3] = G[1] * (-2 * z) + G[2] * 2 * y + G[4] * 2 * z + G[6] * (-2 * x) + G[8] * (-2 * y) + G[9] * 2 * x;
    dp[b * 3 + 0] = G[3]; dp[b * 3 + 1] = G[7]; dp[b * 3 + 2] = G[11];
}

// Object3D.forward's pose head (diffdope/diffdope.py:1085-1098): the seven per-hypothesis parameters, each its own [B] tensor, to
// quat = (qx, qy, qz, qw) / |.| [B,4] and trans [B,3] -- stack, norm, divide, stack in the reference (and ~20 framework kernels in
// the backward of those) as one kernel each way.  |q| = sqrt(((x x + y y) + z z) + w w), the components divided by it; the
// backward is d q = (g - qn (qn . g)) / |q| written into rows 0-3 of d params [7,B], d trans into rows 4-6.
__global__ void pose_pack_fwd_kernel(const float* __restrict__ qx, const float* __restrict__ qy, const float* __restrict__ qz,
                                     const float* __restrict__ qw, const float* __restrict__ x, const float* __restrict__ y,
                                     const float* __restrict__ z, int B, float* __restrict__ quat, float* __restrict__ trans)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float a = qx[b], bq = qy[b], c = qz[b], d = qw[b];
    const float n = sqrtf(((a * a + bq * bq) + c * c) + d * d);
    quat[b * 4 + 0] = a / n; quat[b * 4 + 1] = bq / n; quat[b * 4 + 2] = c / n; quat[b * 4 + 3] = d / n;
    trans[b * 3 + 0] = x[b]; trans[b * 3 + 1] = y[b]; trans[b * 3 + 2] = z[b];
}

__global__ void pose_pack_bwd_kernel(const float* __restrict__ qx, const float* __restrict__ qy, const float* __restrict__ qz,
                                     const float* __restrict__ qw, const float* __restrict__ dquat, const float* __restrict__ dtrans,
                                     int B, float* __restrict__ dparams)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float a = qx[b], bq = qy[b], c = qz[b], d = qw[b];
    const float n = sqrtf(((a * a + bq * bq) + c * c) + d * d);
    const float u0 = a / n, u1 = bq / n, u2 = c / n, u3 = d / n;
    const float g0 = dquat ? dquat[b * 4 + 0] : 0.f, g1 = dquat ? dquat[b * 4 + 1] : 0.f, g2 = dquat ? dquat[b * 4 + 2] : 0.f,
                g3 = dquat ? dquat[b * 4 + 3] : 0.f;
    const float dot = ((u0 * g0 + u1 * g1) + u2 * g2) + u3 * g3;
    dparams[0 * (size_t)B + b] = (g0 - u0 * dot) / n;
    dparams[1 * (size_t)B + b] = (g1 - u1 * dot) / n;
    dparams[2 * (size_t)B + b] = (g2 - u2 * dot) / n;
    dparams[3 * (size_t)B + b] = (g3 - u3 * dot) / n;
    dparams[4 * (size_t)B + b] = dtrans ? dtrans[b * 3 + 0] : 0.f;
    dparams[5 * (size_t)B + b] = dtrans ? dtrans[b * 3 + 1] : 0.f;
    dparams[6 * (size_t)B + b] = dtrans ? dtrans[b * 3 + 2] : 0.f;
}

extern "C" int ddx_pose_pack_fwd(const float* qx, const float* qy, const float* qz, const float* qw, const float* x, const float* y,
                                 const float* z, int B, float* quat, float* trans, void* stream)
{
    DDX_REQUIRE(qx && qy && qz && qw && x && y && z && quat && trans, DDX_E_NULL, "pose_pack_fwd: NULL pointer");
    DDX_REQUIRE(B >= 1, DDX_E_SHAPE, "pose_pack_fwd: B=%d", B);
    pose_pack_fwd_kernel<<<ddx_cdiv(B, 256), 256, 0, (hipStream_t)stream>>>(qx, qy, qz, qw, x, y, z, B, quat, trans);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_pose_pack_bwd(const float* qx, const float* qy, const float* qz, const float* qw, const float* dquat,
                                 const float* dtrans, int B, float* dparams, void* stream)
{
    DDX_REQUIRE(qx && qy && qz && qw && dparams, DDX_E_NULL, "pose_pack_bwd: NULL pointer");
    DDX_REQUIRE(B >= 1, DDX_E_SHAPE, "pose_pack_bwd: B=%d", B);
    pose_pack_bwd_kernel<<<ddx_cdiv(B, 256), 256, 0, (hipStream_t)stream>>>(qx, qy, qz, qw, dquat, dtrans, B, dparams);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_pose_matrix_fwd(const float* q, const float* p, int B, float* mtx, void* stream)
{
    DDX_REQUIRE(q && p && mtx, DDX_E_NULL, "pose_matrix_fwd: NULL pointer");
    DDX_REQUIRE(B >= 1, DDX_E_SHAPE, "pose_matrix_fwd: B=%d", B);
    pose_matrix_fwd_kernel<<<ddx_cdiv(B, 256), 256, 0, (hipStream_t)stream>>>(q, p, B, mtx);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_pose_matrix_bwd(const float* q, const float* dmtx, int B, float* dq, float* dp, void* stream)
{
    DDX_REQUIRE(q && dmtx && dq && dp, DDX_E_NULL, "pose_matrix_bwd: NULL pointer");
    DDX_REQUIRE(B >= 1, DDX_E_SHAPE, "pose_matrix_bwd: B=%d", B);
    pose_matrix_bwd_kernel<<<ddx_cdiv(B, 256), 256, 0, (hipStream_t)stream>>>(q, dmtx, B, dq, dp);
    DDX_LAUNCH_CHECK();
    return 0;
}
