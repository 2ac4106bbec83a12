// raster.h -- scratch layout and entry points of the rasteriser (raster.hip), shared with the renderer
// ops and the fused engine.
#pragma once
#include "raster_math.h"

#define RASTER_SMALL_PX 64    // bbox of at most this many pixel centres => resolved in scatter_kernel (16 / 32 / 64: midpoly 4.7k / 7.5k / 7.6k it/s, lowpoly 10.7k / 9.8k / 11.2k, cfg2 unchanged)
#define RASTER_BIG_GRID 1024  // workgroups of the large-triangle pass
// The row-restricted materialising path: the emit (raster.hip) writes the rows of a hypothesis' active tiles and EMIT_ROW_MARGIN
// rows either side; an antialias workgroup (renderops.hip) that survives the row test reads its own AA_ROWS rows and the one above.
#define AA_ROWS 4
#define EMIT_ROW_MARGIN 8
static_assert(EMIT_ROW_MARGIN >= AA_ROWS + 1, "the antialias blocks next to the active rows read rows the restricted emit must have written");

struct RasterScratch {
    int* counters;            // [16]: 3 + parity = "a large triangle exists in this pass" (plain stores of 1)
    unsigned char* tile_flag; // [npar][B,NTp] != 0: tile holds or borders a possibly covered pixel (plain byte stores of 1)
    unsigned char* tile_big;  // [npar][B,NTp] != 0: a large triangle overlaps the tile
    int* active;              // [B,NT] per-hypothesis ordered list of active tiles, packed ty << 16 | tx (first b_count[b] entries)
    int* b_count;             // [B] active tiles of each hypothesis
    int* row_range;           // [B][2] first / last pixel row of the hypothesis' active tiles (lo > hi: none); op-level compaction only
    int2* snap;               // [B,V] window coordinates in 1/256 px (x = INT_MIN if w <= 0)
    uint2* biglist;           // [B,T] the LARGE triangles of each hypothesis: (triangle id, packed tile range tx0 | ty0<<8 | (nx-1)<<16 | (ny-1)<<24)
    int* bigcount;            // [npar][B] entries of biglist (appended by the scatter pass, one atomic per wave; re-armed by the consumer)
    int* bigarrive;           // [npar][B] workgroups of shade_kernel that finished their share of the hypothesis' tile pass (engine, inline tile pass)
    unsigned long long* zbuf; // [npar][B, zper] (depth key << 32 | triangle id), all ones = background; per hypothesis the frame is
                              // stored in 4x4-pixel blocks (16 entries = one 128-byte line), see zaddr()
    size_t zbuf_bytes;        // all parities
    size_t zper;              // entries per hypothesis = zwb * ceil(H/4) * 16
    int zwb;                  // 4x4 blocks per row = ceil(W/4)
    // The fused engine keeps TWO copies ("parities", by iteration index & 1) of everything a pass dirties -- zbuf, the tile flags,
    // bigcount, the "large triangle" word -- so that the kernel that rasterises iteration i + 1 can re-arm what iteration i
    // dirtied while it draws (engine.hip: step_kernel).  The op-level entry has one copy (npar = 1) and memsets it.
    int npar;
    size_t zero_bytes;        // bytes from `counters` that must be zero before a pass (counters + tile_flag + tile_big + bigcount + bigarrive)
    int ntx, nty, NT;
    int NTp;                  // bytes per hypothesis row of tile_flag / tile_big: NT rounded up to 256 (dword loads of a row stay inside it)
    PixNdc ndc;               // pixel index -> NDC centre constants for (H, W)
#ifdef DDX_TRACE
    unsigned long long* trace;
#endif
};

#ifdef DDX_TRACE
#define DDX_TRACE_BEGIN() const unsigned long long ddx_t0 = __builtin_amdgcn_s_memrealtime()
#define DDX_TRACE_END(buf, kidx, info)                                                                                   \
    do {                                                                                                                 \
        if (threadIdx.x == 0 && (buf)) {                                                                                 \
            const size_t wg = blockIdx.x + gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);                    \
            if (wg < 8192) {                                                                                             \
                unsigned long long* q = (buf) + ((size_t)(kidx) * 8192 + wg) * 4;                                        \
                q[0] = ddx_t0;                                                                                           \
                q[1] = __builtin_amdgcn_s_memrealtime();                                                                 \
                q[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |                 \
                       ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);         \
                q[3] = (info);                                                                                           \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
#else
#define DDX_TRACE_BEGIN()
#define DDX_TRACE_END(buf, kidx, info)
#endif

size_t raster_layout(RasterScratch& L, void* base, int B, int V, int T, int H, int W, int npar = 1);

// zbuf address of pixel (px, py) inside one hypothesis' frame.  4x4 blocks instead of rows: the 64-bit atomicMin stream of
// the rasteriser is bound by the number of distinct 128-byte lines an instruction touches (tools/ubench/atomic_density.hip:
// 48 ps per lane-atomic on random lines, 9 ps on one line), and the fragments of a small triangle and of its neighbours in
// the mesh form a compact 2-D patch -- 16x1 row segments cut it into twice as many lines as 4x4 blocks do.
__device__ __forceinline__ unsigned zaddr(int px, int py, int zwb)
{
    return (unsigned)(((__mul24(py >> 2, zwb) + (px >> 2)) << 4) | ((py & 3) << 2) | (px & 3));
}
// window-coordinate snap of clip positions (the fused engine does this inside its transform kernel)
int raster_snap(const float* pos, int B, int V, int H, int W, const RasterScratch& L, hipStream_t s);
// scatter + (compaction | large-triangle raster) (no emit); asynchronous on s.  Needs L.snap filled.
// clear: memset counters/tile_flag and re-arm zbuf first (the engine maintains both itself).
// ev (nullable): 2 events recorded before the scatter / compact_big launches (profiling)
int raster_run(const float* pos, const int* tri, int B, int V, int T, int H, int W, const RasterScratch& L, hipStream_t s,
               bool clear, hipEvent_t* ev);

// Active-tile bookkeeping without atomics on hot words (device-scope atomics on one address serialise
// across the 8 XCDs: a per-tile counter cost 27 us per pass here): the scatter pass flags tiles with plain byte
// stores; the consumers turn a hypothesis' flags into its ordered tile list themselves (op-level: compact_big_kernel, one
// workgroup per hypothesis; engine: every shading wave scans the hypothesis' flag row, engine.hip tile_scan).
#define WORK_MAX_B 4096
