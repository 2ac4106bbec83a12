// raster.h -- scratch layout and entry points of the tile rasteriser (raster.hip), shared with the
// renderer ops and the fused engine.
#pragma once
#include "raster_math.h"

#define RASTER_GRID 2048  // persistent workgroups striding over the active tiles (256 CUs x 8)
#define SNAP_INVALID INT_MIN  // snapped X of a vertex with clip w <= 0
#define RASTER_SMALL_PX 16    // triangles whose bbox holds at most this many pixel centres are resolved in scatter_kernel

struct RasterScratch {
    int* counters;        // [16]: 0 overflow flag, 1 binned (tile,triangle) pairs, 2 active tiles, 3 binned triangles
    int* tile_count;      // [B,NT] binned (large) triangles per tile
    int* tile_flag;       // [B,NT] != 0: tile is active (holds or borders a possibly covered pixel)
    int* tile_cursor;     // [B,NT] (zeroed by scan_kernel)
    int* tile_offset;     // [B,NT] start of each tile's list in items
    int* active;          // [B*NT] compacted flat ids (b*NT + tile) of non-empty tiles, hypothesis-major
    int* b_active;        // [B,2] (first slot, count) of each hypothesis' active tiles
    int2* snap;           // [B,V] window coordinates in 1/256 px (x = SNAP_INVALID if w <= 0)
    unsigned* trirange;   // [B,T] packed tile range tx0 | ty0<<8 | (nx-1)<<16 | (ny-1)<<24, ~0u = culled
    int* items;           // [capacity] triangle ids
    unsigned long long* zbuf;  // [B,H,W] (depth key << 32 | triangle id), all ones = background
    size_t zbuf_bytes;
    size_t zero_bytes;    // bytes from `counters` that must be zero before a pass (counters + tile_count + tile_flag)
    int capacity;
    int ntx, nty, NT;
};

size_t raster_layout(RasterScratch& L, void* base, int B, int V, int T, int H, int W, long long pairs_hint);
// window-coordinate snap of clip positions (the fused engine does this inside its transform kernel)
int raster_snap(const float* pos, int B, int V, int H, int W, const RasterScratch& L, hipStream_t s);
// scatter + scan + bin fill + big-triangle raster (no emit); asynchronous on s.  Needs L.snap filled.
// clear: memset counters/tile_count/tile_flag and re-arm zbuf first (the engine maintains both itself).
// ev (nullable): 4 events recorded before the count / scan / fill / raster launches (profiling)
int raster_run(const float* pos, const int* tri, int B, int V, int T, int H, int W, const RasterScratch& L, hipStream_t s,
               bool clear, hipEvent_t* ev);
