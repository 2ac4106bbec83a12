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
