// raster.h -- scratch layout and entry points of the tile rasteriser (raster.hip), shared with the
// renderer ops and the fused engine.
#pragma once
#include "raster_math.h"

#define RASTER_GRID 2048  // persistent workgroups striding over the active tiles (256 CUs x 8)

struct RasterScratch {
    int* counters;     // [16]: 0 overflow flag, 1 (tile,triangle) pairs, 2 active tiles
    int* tile_count;   // [B,NT]
    int* tile_cursor;  // [B,NT]
    int* tile_offset;  // [B,NT] start of each tile's list in items
    int* active;       // [B*NT] compacted flat ids (b*NT + tile) of non-empty tiles, hypothesis-major
    int* b_active;     // [B,2] (first slot, count) of each hypothesis' active tiles
    int* items;        // [capacity] triangle ids
    unsigned* vis;     // [B,H,W] triangle id + 1, valid inside active tiles only
    size_t zero_bytes; // bytes from `counters` to clear at the start of every pass
    int capacity;
    int ntx, nty, NT;
};

size_t raster_layout(RasterScratch& L, void* base, int B, int T, int H, int W, long long pairs_hint);
// bin + scan + fill + tile raster (no emit); asynchronous on s.  ev (nullable): 4 events recorded before the
// count / scan / fill / raster launches (profiling)
int raster_run(const float* pos, const int* tri, int B, int V, int T, int H, int W, const RasterScratch& L, hipStream_t s,
               hipEvent_t* ev);
