// engine.hip -- fused render-and-compare refinement engine for gfx950.
//
// One call of ddx_engine_run(it0, n) executes n iterations of DiffDope.run_optimization's loop body
// (diffdope/diffdope.py:1656-1714) for B pose hypotheses with the built-in losses
// (diffdope.py:547-613), entirely on the device.  An iteration is THREE launches:
//
//   step_kernel    workgroup = (meshlet, hypothesis).  Head (every workgroup of a hypothesis, redundantly -- a few KB from L2):
//                  the optimiser step of the PREVIOUS iteration: fixed-order sum of the hypothesis' partial rows, whole-frame
//                  constants for the pixels outside the active tiles, proj^T chain, quaternion chain, SGD/Adam, loss log
//                  (diffdope.py:558,576,604,1713-1714).  Then, with the new pose: q/|q|, [R|t] (diffdope.py:46-89,1085-1098),
//                  final = proj . mtx (:195), and the workgroup's MESHLET -- up to 512 triangles and the <= 512 vertices they
//                  use, a static table -- goes through the pipeline without leaving the CU: clip = final . [pos;1] on the matrix
//                  core (MFMA 4x4x1, as xfm.hip) (:196) and the 1/256-pixel window snap into LDS, then the per-triangle scatter
//                  rasteriser (raster_dev.h) reads its vertices from LDS: exact integer coverage, fp32 depth, 64-bit atomicMin
//                  of (depth key, id) into zbuf, byte flags for the touched 16x16 tiles, large / near-clipped triangles to the
//                  hypothesis' list (:198).  Each vertex is also stored once to clip / snap in HBM (by the meshlet that owns
//                  it) for the antialias pass and the tile pass.  zbuf, the tile flags and the large-triangle counters exist
//                  twice, by iteration parity: while drawing iteration i the workgroups re-arm what iteration i - 1 dirtied.
//   big_pass_kernel  the tile pass for large triangles; exits on one scalar load when the batch has none.
//   shade_kernel   every wave scans its hypothesis' tile-flag row (no list kernel, no atomics) and takes the tiles of its
//                  slice; per pixel of the active tiles, in registers: barycentrics, uv/colour/position
//                  interpolation (:203,:218,:230), bilinear texture (:221), depth (:204-209),
//                  antialiased coverage (:212-214), the three L1 terms against the observed images,
//                  AND the whole analytic backward down to d loss / d(final, mtx) -- the per-pixel
//                  loss gradient is known locally (sign * seg * lr_b * w / (B n_px)), so forward and
//                  backward are one pass and no G-buffer (rast, gb_pos, texc, color, mask: 1.1 GB per
//                  iteration at 64 x 640x480 in the reference) is ever written.  Vertex gradients are
//                  contracted with [pos;1] in registers (the xfm_bwd_mtx product, mesh.cu:165-214),
//                  reduced per workgroup and written as one 24-float partial row per (slice, role): no atomics,
//                  bit-reproducible.
//   edge_kernel    (edge extension only) Sobel-gradient L1 of the rendered luminance against the observed image, from the
//                  luminance + unit gradients the colour role of shade_kernel wrote; owner-computes, texture-free
// and finish_kernel (the head of step_kernel alone) closes a run with the optimiser step of its last iteration.
//
// Whole-frame semantics without whole-frame work: a pixel outside every active tile renders
// rgb = 0, mask = 0, depth = -mtx[2][3], so its loss terms are
// |gt*seg|, |seg| and |(-t_z - gt_d) seg0|.  The first two are constants of the observed images
// (summed once at engine creation); the third is evaluated per hypothesis and iteration from the depth-sorted
// list of pixels with seg0 != 0.  Pixels of active tiles add (actual - background) terms.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>
#include <cstdio>

#include <dlfcn.h>

#include "raster_dev.h"

// roctx ranges around the runs, iterations and launches of the engine (rocprofv3 --marker-trace): on with
// DDX_ROCTX=1 in the environment (read once).  libroctx64 is looked up at run time, and only then: the library neither links
// against roctracer nor needs it installed while the switch is off.
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
static const RoctxApi& ddx_roctx()
{
    static const RoctxApi api = [] {
        RoctxApi a;
        const char* v = getenv("DDX_ROCTX");
        if (!(v && atoi(v))) return a;
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            fprintf(stderr, "ddx: DDX_ROCTX=1 but libroctx64.so cannot be loaded (%s): no ranges\n", dlerror());
            return a;
        }
        a.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        a.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!a.push || !a.pop) a.push = nullptr, a.pop = nullptr;
        return a;
    }();
    return api;
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(ddx_roctx().push != nullptr) { if (on) ddx_roctx().push(name); }
    ~RoctxRange() { if (on) ddx_roctx().pop(); }
};

#define NPART 24  // floats per tile partial: 12 dFinal(x,y,w rows) | 4 dMtx row 2 | 4 losses (rgb, depth, mask, edge) | pad
#define NVALS 20  // of which are used
#define MAX_ROLES 3
#define NROLE 3  // partial rows per slice: colour + depth, mask, edge -- always three, so that an engine sums in the same order whatever kernels of a group it runs under

#ifndef SCATTER_EXCHANGE_PER_TRI
#define SCATTER_EXCHANGE_PER_TRI 2.0  // measured crossover with the meshlet vertices in LDS (round 3): plain ahead at 1.4 (cfg2 at 4.7 % coverage: 12.7 k vs 12.2 k it/s), the hybrid at 3.1 (10.7 %: 7.6 k vs 6.7 k)
#endif

struct EngineState {  // device-resident; the first 8 ints are what ddx_engine_status_ptr exposes
    int overflow;
    int last_pairs;
    int last_active;
    int it;          // iteration the next step_kernel draws (written by one lane of shade_kernel, read by step / finish)
    int n_seg;       // entries in the compact seg list
    int it_next;     // iteration being drawn + 1 (written by one lane of step_kernel, read by big_pass / shade / edge): no kernel
                     // reads a word that the same launch writes
    int outside;     // hypotheses of the last iteration whose bounding box left the view volume (w <= 0 or |z| > w at a corner):
                     // their triangles at w <= 0 were clipped at the near plane by the tile pass and their back faces drawn (D5 off)
    int flags;       // ENGINE_FLAG_* bits (ddx.h, status word 7): sticky until ddx_engine_run_check has acted on them
    double c_rgb;    // sum over the frame of |gt_rgb * seg|
    double c_mask;   // sum over the frame of |seg|
    double c_edge;   // sum over the frame of |Gx| + |Gy| of the observed, masked image (edge extension)
    // selection of the best hypothesis inside finish_kernel (ddx_engine_run_select): the hypotheses' writer workgroups fold
    // (order-preserving bits of the mean loss << 32 | index) into sel_key with atomicMin and count themselves in sel_arrive; the
    // last one writes the [18] row.  Both words are all-ones / zero between runs (re-armed by that last workgroup).
    unsigned long long sel_key;
    int sel_arrive;
    int sel_pad;
};

// EngineState::flags
#define ENGINE_FLAG_INLINE_TIMEOUT 1  // a shading workgroup gave up waiting for the workers of the in-launch tile pass (big_wait)

// step_kernel(mode): STEP_FIRST draws the first iteration of a run from the caller's parameters (no optimiser step);
// STEP_NORMAL steps the optimiser for iteration it - 1 and draws iteration it
enum { STEP_FIRST = 0, STEP_NORMAL = 1, STEP_EVAL = 2 };  // STEP_EVAL: STEP_FIRST of an evaluation / profile pass (leaves run_snap alone)

struct EngineDev {
    ddx_engine_desc d;
    ddx_engine_buffers b;
    RasterScratch L;  // two parities (npar = 2) of zbuf / tile flags / large-triangle counters
    float* clip;      // [B,V,4] clip-space vertices in sorted vertex order, stored once per vertex by the meshlet that owns it
                      // (read by the mask role and the tile pass; the colour role recomputes its vertices from crec)
    float* mats;      // [2][B,2,16]: mtx | final, by iteration parity
    float* params2;   // [2][7,B]: the parameters, by iteration parity (b.params is the user-visible copy)
    float* partials;  // [B, pslices, NROLE, NPART]: one row per slice (workgroup of the shade / edge grid) and role
    float2* gtedge;   // [H*W] Sobel gradients of lum(gt_rgb * seg) (edge extension), or null
    float* lumbuf;    // [B,H*W] luminance of the rendered colour at covered pixels (written by the colour role, read by
    float* ubuf;      // [B,H*W,12] edge_kernel; garbage where zbuf says "background") and U = d lum / d final per pixel
    float* adam;      // [2][2,7,B]: first and second moments, by iteration parity
    float2* seglist;  // [H*W] (gt_depth, seg0) of pixels with seg0 != 0 (setup only)
    struct SetupPart* setup_part;  // [ceil(H W / SETUP_CHUNK)] per-chunk partials of the observation set-up
    // the same pixels sorted by observed depth, with prefix sums (double) of |seg0| and |seg0| * depth: the whole-frame
    // background depth term sum_i |seg0_i| |d - gt_i| and its derivative for a hypothesis' background depth d are two
    // searches and six loads instead of a pass over the list (10 700 entries per workgroup on cfg5: 10 of the update's 21 us)
    float* seg_gd;    // [n_seg] ascending
    double* seg_W;    // [n_seg + 1] W[k] = sum_{i<k} |seg0_i|
    double* seg_G;    // [n_seg + 1] G[k] = sum_{i<k} |seg0_i| gt_i
    int nseg;
    // Internal, spatially sorted copy of the mesh (built once per engine): vertices renumbered in Morton order of their
    // object-space position, so that the vertex data of neighbouring triangles / pixels are neighbours in memory whatever
    // order the mesh file had (a randomly ordered vertex list cost 15-20 % otherwise); triangle ids are NOT renumbered.
    float* spos;      // [V,3] positions, sorted vertex order
    float* suv;       // [V,2] or null
    float* scol;      // [V,3] or null
    int* stri;        // [T,3] triangles (original triangle order) with sorted vertex ids
    int* vnew;        // [V] old vertex id -> sorted id
    int* vold;        // [V] sorted id -> old vertex id
    // MESHLETS: the triangles in Morton order of their object-space centroids, cut into runs of at most mesh_ntri triangles that
    // use at most mesh_nvc distinct vertices (fixed-size slots, so a workgroup's loads do not depend on a header):
    float4* mvert;    // [M, mesh_nvc] (x, y, z, bits): object-space position; bits = sorted vertex id | owner << 31, or all ones
                      // for an unused slot.  Every referenced vertex is owned by exactly one meshlet (the first that uses it).
    int2* mtri;       // [M, mesh_ntri] (l0 | l1 << 10 | l2 << 20 local vertex slots, original triangle id or -1 for an unused slot)
    int n_meshlets, mesh_ntri, mesh_nvc;
    unsigned long long* trace;  // DDX_TRACE=1: [3 kernels][TRACE_WG workgroups][8] s_memrealtime stamps of thread 0 (tools/trace_kernels.py), else null
    // Which meshlets a slot (workgroup of a hypothesis) draws.  slot_table = 0: equal contiguous shares of the interleaved order.
    // 1: slot s draws slot_items[slot_off[s] .. + slot_cnt[s]) -- shares balanced on the host from MEASURED per-meshlet times (mcost,
    // recorded by the first step launch after a set-up) and from the speed of the slot's place in the dispatch order (launch_step).
    int slot_table;
    unsigned char slot_cnt[64];
    unsigned short slot_off[64];
    unsigned short slot_items[192];
    unsigned short* mcost;      // [B, M] ticks (10 ns) a workgroup spent on meshlet m of hypothesis b
    int mcost_rec;              // this launch records mcost
    int step_xcd;     // step_kernel grid: 0 = (slots, B); 1 = (B, slots): all workgroups of hypothesis b on XCD b % 8 (observed placement)
    int scatter_mode; // scatter_resolve MODE of the dense variant: 0 plain, 2 hybrid, 3 compacting (the small-mesh variant is MODE 1)
    // Back-face culling for CLOSED meshes (deviation D5).  A closed, consistently oriented surface that lies
    // entirely inside the view volume covers every pixel centre with as many front- as back-facing triangles, and the nearest one
    // is front-facing: skipping the back faces changes nothing in exact arithmetic and halves the fragments.  cull_sign: 0 = off;
    // +1 / -1 = triangles whose SNAPPED area has this sign are back faces (sign(signed volume) * sign(det of proj's x,y,w rows),
    // decided once per engine on the host).  A hypothesis culls only while the 8 corners of the mesh's object-space bounding box
    // (bbox), transformed like vertices, all have w > 0 and -w <= z <= w -- then so has every vertex (the three conditions are
    // half-spaces of object space) and the drawn surface is closed.
    int cull_sign;
    float bbox[6];    // lo x,y,z, hi x,y,z over all vertices (cull_sign = 0 if any coordinate is not finite)
    int* inside;      // [B] the box test of the iteration last drawn (status only)
    float4* crec;     // [T,4] (textured) or [T,5] (vertex colours): what the colour role needs of a covered triangle, in ONE
                      // record fetched by triangle id: object-space positions of the 3 vertices (9), then uv (6) or colours (9)
    int4* trirec;     // [T,2] {v0,v1,v2,opp0} {opp1,opp2,0,0}: one record per triangle for the antialias pass
    float4* texq;     // [Th*Tw,4] the texture as one 64-byte record per texel (x,y): the 2x2 bilinear footprint whose corner it
                      // is -- (x,y), (x+1,y), (x,y+1), (x+1,y+1) with wrap, rgb each, 4 floats of padding -- or null.  A
                      // sample is then ONE 64-byte sector instead of two 24-byte row pieces (2.75 sectors on average): the
                      // object is minified (cfg2: 30 texels per pixel), so no two samples share a line anyway and the texel
                      // gathers are most of the shading kernel's DRAM traffic.  4x the memory of the [Th,Tw,3] texture.
    EngineState* st;
    int st_role;      // shade role that advances the iteration counter (= roles[0])
    int n_roles;      // enabled shade roles (grid z of shade_kernel): 0 colour+depth, 1 antialiased mask, 2 edge
    int roles[MAX_ROLES];    // grid z -> role
    int role_mask;           // bit r set = role r runs
    int s_shade, s_edge;     // slices per hypothesis of shade_kernel / edge_kernel (grid y)
    int big_inline;          // 1: the shading launch carries the tile pass for large triangles itself (worker workgroups in its first
                             // slab, big_worker_wg) and big_pass_kernel is not launched; 0: the launch between step_kernel and shade_kernel
    int big_workers;         // ... worker workgroups per hypothesis that look at its count (<= slices; the others leave on their block id)
    int pslices;             // rows of the partial table per hypothesis: max(s_shade, s_edge) slices
    float* eval_grad;        // [7,B] or null.  Non-null = evaluation pass (ddx_engine_eval): d loss / d params and the
    float* eval_loss;        // [4,B] losses are written here, no optimiser step
    float* eval_tmp;         // [7,B] gradient sink of ddx_render_loss_fwd
    float* sel_out;          // [18] or null.  Non-null (finish_kernel of ddx_engine_run_select): (mean loss of the best hypothesis, its
    int sel_lo;              // global index = sel_lo + local index, its 4x4 pose) is written here by the last writer workgroup
    int b_off;               // first hypothesis of this launch (0 except in the half-batch launches of a two-stream run, engine_run_impl)
    float* run_snap;         // [21,B] parameters (7) and optimiser moments (14) as the LAST run / evaluation found them, written by its first
                             // step launch: what ddx_engine_run_check restores before it repeats a run whose in-launch tile pass timed out
    unsigned wait_ticks;     // big_wait's budget in ticks of the 100 MHz clock (DDX_BIG_WAIT_US; default 20 ms)
    int dbg_reverse;         // DDX_DEBUG_REVERSE_SLABS=1 (tests): the worker slab of the in-launch tile pass BEHIND the shading slabs --
                             // the dispatch order in which the wait cannot be satisfied while the shading workgroups fill the chip
};

// The optimiser head of step_kernel runs at a raised wave priority (s_setprio; 0 = off).  Its tail is one wave's chain of ~500 dependent
// instructions, and the five workgroups of a CU keep theirs on the same SIMD: a workgroup whose partial rows arrive late shares
// that SIMD with four neighbours that are already rasterising and used to get a fifth of its issue slots -- the head took 5.1 us in the
// median and 8.9 / 13.8 us at the 90th percentile / worst, and the late tenth of a launch's workgroups was late BECAUSE of it
// (tools/trace_tail.py, profiles/r6i_*).  With the priority: 7.1 / 8.4 us; cfg2 one chain 42.8 -> 41.8 us, two chains 40.8 -> 40.3
// (priority 2 the same; the scan phase of shade_kernel raised likewise: no difference).
#ifndef DDX_HEAD_PRIO
#define DDX_HEAD_PRIO 3
#endif
#define TRACE_WG 4096
// stamp i of kernel slot k (0 step, 1 shade, 2 other) for this workgroup: 100 MHz constant clock
#define STAMP(E, k, wg, i)                                                                                  \
    do {                                                                                                    \
        if ((E).trace && threadIdx.x == 0 && (wg) < TRACE_WG)                                               \
            (E).trace[((size_t)(k) * TRACE_WG + (wg)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();         \
    } while (0)

// ... and where the workgroup runs (row 2 of the trace, word 7; words 0..6: the head's own stamps of a DDX_TRACE_HEAD build): HW_ID (cu 11:8, sh 12, se 15:13, simd 5:4) | XCC_ID << 32
#define STAMP_HW(E, wg)                                                                                                  \
    do {                                                                                                                 \
        if ((E).trace && threadIdx.x == 0 && (wg) < TRACE_WG)                                                            \
            (E).trace[((size_t)2 * TRACE_WG + (wg)) * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | \
                                                           ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); \
    } while (0)

struct ddx_engine {
    EngineDev dev;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int graph_chunk = 1;  // iterations per captured graph
    bool setup_done = false;
    bool inline_ok = false;    // the tile pass MAY run inside shade_kernel (batch shape, residency, desc): decided at creation
    int inline_env = -1;       // DDX_BIG_INLINE: 0 = never, 1 = whenever inline_ok, -1 = the set-up's estimate decides
    double mesh_area = 0.0, mesh_max_edge = 0.0;  // object space: sum of the triangle areas, longest edge (mesh half of the set-up)
    unsigned setup_gen = 0;  // bumped by every engine_setup: a group re-uploads the table row of a member whose set-up ran outside it
    bool mesh_done = false;  // the mesh half of the setup (sorted copies, meshlets, triangle / texel records, closedness) survives ddx_engine_new_observation
    int step_resident = 0;     // step_kernel workgroups the chip holds at once (step_capacity, asked once); DDX_STEP_RESIDENT overrides
    int balance = 1;           // balanced shares of the meshlets (balance_slots); DDX_STEP_BALANCE=0: equal shares
    bool balanced = false;     // the table of this set-up exists
    int balance_min_per_slot = 4;  // DDX_STEP_BALANCE_MIN
    bool small_mesh = false; // step_kernel variant: one triangle per lane in 64-thread workgroups (few triangles x hypotheses)
    int adam_parity = 0;  // which half of dev.adam holds the optimiser state of the last finished iteration
    int two_streams = 1;     // DDX_TWO_STREAMS: 1 = the iterations of a run after its first as two half-batch chains on two streams, 0 = never
    bool two_min_env = false;
    int two_min_iters = 16;  // ... for runs of at least this many iterations (DDX_TWO_MIN)
    hipStream_t side = nullptr;  // ... the second stream for the caller stream of the current run (the process-wide registry's: ensure_side_stream), and the events that fork it from / join it to the caller's
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int probe_outcome = -1;  // ddx_engine_two_chains: the last answer of ensure_side_stream (-1: never asked)
    // the last run that ddx_engine_run_check has not yet seen clean, as it would have to be repeated (kind 0: none)
    struct { int kind = 0, it0 = 0, n = 0, use_graph = 0, sel_lo = 0; float* sel_out = nullptr; } last;
    int fwd_cached_it = -1;  // >= 0: dev.eval_tmp holds d loss / d params of the ddx_render_loss_fwd pass at this iteration (for the
                             // ddx_render_loss_bwd that follows); any other pass of the engine invalidates it
};

// ENGINE GROUPS (ddx_engine_group_*): several engines -- the objects of one frame, BASELINE config 5 -- advance one iteration with
// ONE launch of each kernel.  Every member keeps its own scratch, state and launch geometry (slices, partial rows: the same bits as
// when it runs alone); the group kernels only decode (member, block) from the block index and run the member's body on its
// EngineDev, read from a device table.  A latency-bound 64-hypothesis launch fills a fraction of the chip; four of them in one
// grid run in the large-batch regime (tools/large_batch.py).
#define GROUP_MAX 32
struct GroupHdr {
    int n;                     // members of this launch
    int idx[GROUP_MAX];        // their rows of the device table
    int bpre[GROUP_MAX + 1];   // prefix sums of their hypothesis counts
    int sl[GROUP_MAX];         // step_kernel slots per hypothesis of each
};

__device__ __forceinline__ int group_find(const GroupHdr& G, int g)  // the member that owns hypothesis g of the launch
{
    int o = 0;
    for (int i = 1; i < G.n; ++i) o += g >= G.bpre[i] ? 1 : 0;
    return o;
}

// meshlet geometry of the two step_kernel variants (triangles, vertex slots): dense = (2, 256), small = (1, 64) threads
static inline void mesh_geometry(bool small_mesh, int& ntri, int& nvc)
{
    ntri = small_mesh ? 64 : 512;
    nvc = small_mesh ? 128 : 512;
}

// small meshes: 512-triangle meshlets would leave most of the chip without a workgroup (a 384-triangle CAD model x 64
// hypotheses = 64 workgroups)
static inline bool mesh_is_small(const ddx_engine_desc& d) { return (long long)ddx_cdiv(d.T, 512) * d.B < 1024; }

// ---------------------------------------------------------------------------------------------
static size_t engine_layout(EngineDev& E, const ddx_engine_desc& d, void* base)
{
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    char* p = (char*)base;
    const size_t o_state = carve(sizeof(EngineState));
    // per-hypothesis state that the optimiser step both reads and writes is double-buffered by iteration parity: every
    // workgroup of a hypothesis reads buffer it & 1 and one of them writes buffer (it + 1) & 1, so a workgroup that starts late
    // can never see the next iteration's values
    const size_t o_mats = carve((size_t)2 * d.B * 32 * sizeof(float));
    const size_t o_adam = carve((size_t)2 * 14 * d.B * sizeof(float));
    const size_t o_par = carve((size_t)2 * 7 * d.B * sizeof(float));
    const size_t o_etmp = carve((size_t)7 * d.B * sizeof(float));
    const size_t o_rsnap = carve((size_t)21 * d.B * sizeof(float));
    const size_t o_inside = carve((size_t)d.B * sizeof(int));
    const size_t o_clip = carve((size_t)d.B * d.V * 4 * sizeof(float));
    const size_t o_seg = carve((size_t)d.H * d.W * sizeof(float2));
    const size_t o_spart = carve(((size_t)d.H * d.W / 4096 + 1) * 40);  // SetupPart per chunk of SETUP_CHUNK = 4096 pixels
    const size_t o_sgd = carve(d.use_depth ? (size_t)d.H * d.W * sizeof(float) : 0);
    const size_t o_sW = carve(d.use_depth ? ((size_t)d.H * d.W + 1) * sizeof(double) : sizeof(double));
    const size_t o_sG = carve(d.use_depth ? ((size_t)d.H * d.W + 1) * sizeof(double) : sizeof(double));
    const size_t o_rec = carve((size_t)d.T * 2 * sizeof(int4));
    // meshlets: a run closes at mesh_ntri triangles or mesh_nvc vertices; a triangle brings at most 3 new vertices, so runs
    // that close on the vertex limit hold at least (mesh_nvc - 2) / 3 triangles: M <= T / ((nvc - 2) / 3) + T / ntri + 1
    int ntri, nvc;
    mesh_geometry(mesh_is_small(d), ntri, nvc);
    const size_t Mmax = (size_t)d.T / ((nvc - 2) / 3) + (size_t)d.T / ntri + 2;
    const size_t o_mvert = carve(Mmax * nvc * sizeof(float4));
    const size_t o_mtri = carve(Mmax * ntri * sizeof(int2));
    const size_t o_crec = carve((size_t)d.T * 5 * sizeof(float4));
    const size_t o_texq = carve(d.Th > 0 ? (size_t)d.Th * d.Tw * 4 * sizeof(float4) : 0);
    const size_t o_spos = carve((size_t)d.V * 3 * sizeof(float));
    const size_t o_suv = carve((size_t)d.V * 2 * sizeof(float));
    const size_t o_scol = carve((size_t)d.V * 3 * sizeof(float));
    const size_t o_stri = carve((size_t)d.T * 3 * sizeof(int));
    const size_t o_vnew = carve((size_t)d.V * sizeof(int));
    const size_t o_vold = carve((size_t)d.V * sizeof(int));
    const size_t o_part = carve((size_t)d.B * 64 * MAX_ROLES * NPART * sizeof(float));  // per (slice <= 64, role)
    const size_t o_edge = carve(d.use_edge ? (size_t)d.H * d.W * sizeof(float2) : 0);
    const size_t o_mcost = carve((size_t)d.B * Mmax * sizeof(unsigned short));
    const size_t o_lum = carve(d.use_edge ? (size_t)d.B * d.H * d.W * sizeof(float) : 0);
    const size_t o_ubuf = carve(d.use_edge ? (size_t)d.B * d.H * d.W * 12 * sizeof(float) : 0);
    const size_t o_rast = carve(0);
    const size_t rast_bytes = raster_layout(E.L, p + o_rast, d.B, d.V, d.T, d.H, d.W, 2);
    off += rast_bytes;
    E.st = (EngineState*)(p + o_state);
    E.mats = (float*)(p + o_mats);
    E.adam = (float*)(p + o_adam);
    E.params2 = (float*)(p + o_par);
    E.eval_tmp = (float*)(p + o_etmp);
    E.run_snap = (float*)(p + o_rsnap);
    E.inside = (int*)(p + o_inside);
    E.clip = (float*)(p + o_clip);
    E.seglist = (float2*)(p + o_seg);
    E.setup_part = (struct SetupPart*)(p + o_spart);
    E.seg_gd = (float*)(p + o_sgd);
    E.seg_W = (double*)(p + o_sW);
    E.seg_G = (double*)(p + o_sG);
    E.trirec = (int4*)(p + o_rec);
    E.mvert = (float4*)(p + o_mvert);
    E.mtri = (int2*)(p + o_mtri);
    E.mesh_ntri = ntri;
    E.mesh_nvc = nvc;
    E.crec = (float4*)(p + o_crec);
    E.texq = d.Th > 0 ? (float4*)(p + o_texq) : nullptr;
    E.spos = (float*)(p + o_spos);
    E.suv = (float*)(p + o_suv);
    E.scol = (float*)(p + o_scol);
    E.stri = (int*)(p + o_stri);
    E.vnew = (int*)(p + o_vnew);
    E.vold = (int*)(p + o_vold);
    E.partials = (float*)(p + o_part);
    E.gtedge = d.use_edge ? (float2*)(p + o_edge) : nullptr;
    E.mcost = (unsigned short*)(p + o_mcost);
    E.lumbuf = d.use_edge ? (float*)(p + o_lum) : nullptr;
    E.ubuf = d.use_edge ? (float*)(p + o_ubuf) : nullptr;
    return off;
}

// ---------------------------------------------------------------------------------------------
// Observation set-up (once per frame): frame constants -- sum |gt_rgb seg|, sum |seg|, and for the edge extension (no reference
// counterpart; definition: oracle/ddx_oracle.c orc_loss_edge) the Sobel gradients of the luminance of the observed image masked by
// its segmentation and their whole-frame L1 norm -- and the compact list of pixels with seg0 != 0 for the depth term.
// Three launches over chunks of SETUP_CHUNK pixels: per-chunk partials, one ordered scan over the chunks, ordered compaction.
// (One 1024-thread workgroup did all of it at first: 1.3 ms + 3.2 ms for the edge map at 1280x720 -- a quarter of a 58-iteration
// frame per object.)  Fixed shapes and orders throughout: the result does not depend on scheduling.
#define SETUP_CHUNK 4096
__device__ __forceinline__ float lum3(float r, float g, float b) { return ((r + g) + b) * (1.0f / 3.0f); }

__device__ __forceinline__ void sobel3(const float v[3][3], float& gx, float& gy)
{
    gx = (((v[0][2] - v[0][0]) + 2.0f * (v[1][2] - v[1][0])) + (v[2][2] - v[2][0])) * 0.125f;
    gy = (((v[2][0] - v[0][0]) + 2.0f * (v[2][1] - v[0][1])) + (v[2][2] - v[0][2])) * 0.125f;
}

struct SetupPart { double s_rgb, s_mask, s_edge; int count, offset; };  // per chunk (EngineDev::setup_part)
static_assert(sizeof(SetupPart) <= 40, "engine_layout carves 40 bytes per chunk");

// block sum of a double per thread, fixed order: wave shuffles, then the four waves
__device__ __forceinline__ double block_sum_256(double v, double* red /* LDS [4] */)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void setup_part_kernel(EngineDev E)
{
    const int H = E.d.H, W = E.d.W, n = H * W, tid = threadIdx.x;
    __shared__ double red[4];
    double s_rgb = 0.0, s_mask = 0.0, s_edge = 0.0;
    int cnt = 0;
    for (int k = 0; k < SETUP_CHUNK / 256; ++k) {
        const int i = blockIdx.x * SETUP_CHUNK + k * 256 + tid;
        if (i >= n) break;
        const float s0 = E.b.gt_seg[i * 3 + 0], s1 = E.b.gt_seg[i * 3 + 1], s2 = E.b.gt_seg[i * 3 + 2];
        s_mask += (double)(fabsf(s0) + fabsf(s1) + fabsf(s2));
        if (E.b.gt_rgb)
            s_rgb += (double)(fabsf(E.b.gt_rgb[i * 3 + 0] * s0) + fabsf(E.b.gt_rgb[i * 3 + 1] * s1) + fabsf(E.b.gt_rgb[i * 3 + 2] * s2));
        cnt += (E.b.gt_depth && s0 != 0.f) ? 1 : 0;
        if (E.gtedge) {
            const int y = i / W, x = i - y * W;
            float v[3][3];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    float l = 0.f;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        const size_t q = ((size_t)yy * W + xx) * 3;
                        l = lum3(E.b.gt_rgb[q] * E.b.gt_seg[q], E.b.gt_rgb[q + 1] * E.b.gt_seg[q + 1], E.b.gt_rgb[q + 2] * E.b.gt_seg[q + 2]);
                    }
                    v[dy + 1][dx + 1] = l;
                }
            float gx, gy;
            sobel3(v, gx, gy);
            E.gtedge[i] = make_float2(gx, gy);
            s_edge += (double)(fabsf(gx) + fabsf(gy));
        }
    }
    const double a = block_sum_256(s_rgb, red), bq = block_sum_256(s_mask, red), c = block_sum_256(s_edge, red);
    const int total = (int)block_sum_256((double)cnt, red);
    if (tid == 0) {
        SetupPart& P = E.setup_part[blockIdx.x];
        P.s_rgb = a; P.s_mask = bq; P.s_edge = c; P.count = total; P.offset = 0;
    }
}

// one workgroup: the chunks' partials in chunk order -> frame constants, list size, every chunk's offset into the list
__global__ __launch_bounds__(256) void setup_scan_kernel(EngineDev E, int n_chunks)
{
    const int tid = threadIdx.x;
    const int per = (n_chunks + 255) / 256, c0 = tid * per, c1 = min(n_chunks, c0 + per);
    __shared__ double s_a[256], s_b[256], s_c[256];
    __shared__ int s_n[256];
    double a = 0.0, bq = 0.0, c = 0.0;
    int cnt = 0;
    for (int i = c0; i < c1; ++i) {
        const SetupPart& P = E.setup_part[i];
        a += P.s_rgb; bq += P.s_mask; c += P.s_edge; cnt += P.count;
    }
    s_a[tid] = a; s_b[tid] = bq; s_c[tid] = c; s_n[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        double ta = 0.0, tb = 0.0, tc = 0.0;
        int run = 0;
        for (int t = 0; t < 256; ++t) {
            ta += s_a[t]; tb += s_b[t]; tc += s_c[t];
            const int m = s_n[t];
            s_n[t] = run;
            run += m;
        }
        E.st->c_rgb = ta; E.st->c_mask = tb; E.st->c_edge = tc; E.st->n_seg = run;
    }
    __syncthreads();
    int off = s_n[tid];
    for (int i = c0; i < c1; ++i) {
        E.setup_part[i].offset = off;
        off += E.setup_part[i].count;
    }
}

// ordered compaction of the pixels with seg0 != 0 (ballot ranks + wave offsets, in pixel order)
__global__ __launch_bounds__(256) void setup_fill_kernel(EngineDev E)
{
    const int n = E.d.H * E.d.W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int wcnt[4];
    int carry = E.setup_part[blockIdx.x].offset;
    for (int k = 0; k < SETUP_CHUNK / 256; ++k) {  // (workgroup-uniform)
        const int i = blockIdx.x * SETUP_CHUNK + k * 256 + tid;
        float gd = 0.f, s0 = 0.f;
        bool flag = false;
        if (i < n) {
            s0 = E.b.gt_seg[i * 3 + 0];
            if (s0 != 0.f) { flag = true; gd = E.b.gt_depth[i]; }
        }
        const unsigned long long m = __ballot(flag);
        __syncthreads();
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        if (flag) E.seglist[off + __popcll(m & ((1ull << lane) - 1ull))] = make_float2(gd, s0);
        carry += (wcnt[0] + wcnt[1]) + (wcnt[2] + wcnt[3]);
    }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void quat_to_matrix(const float q[4], const float t[3], float M[16])
{
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    M[0] = 1.f - 2.f * y * y - 2.f * z * z; M[1] = 2.f * x * y - 2.f * z * w; M[2] = 2.f * x * z + 2.f * y * w; M[3] = t[0];
    M[4] = 2.f * x * y + 2.f * z * w; M[5] = 1.f - 2.f * x * x - 2.f * z * z; M[6] = 2.f * y * z - 2.f * x * w; M[7] = t[1];
    M[8] = 2.f * x * z - 2.f * y * w; M[9] = 2.f * y * z + 2.f * x * w; M[10] = 1.f - 2.f * x * x - 2.f * y * y; M[11] = t[2];
    M[12] = 0.f; M[13] = 0.f; M[14] = 0.f; M[15] = 1.f;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// q/|q| (diffdope.py:1091), [R|t] (diffdope.py:46-89) and final = proj . mtx (torch.matmul at :195, k-ordered fma)
__device__ __forceinline__ void pose_matrices(float q[4], const float t[3], const float* proj, float M[16], float F[16])
{
    const float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __fdiv_rn(q[i], nq);
    quat_to_matrix(q, t, M);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) a = __fmaf_rn(proj[r * 4 + k], M[k * 4 + c], a);
            F[r * 4 + c] = a;
        }
}

// row r of final = proj . mtx, the k-ordered fma chain of pose_matrices (same bits).  The matrix-core transform wants row
// lane % 4 of final in each lane; selecting it from a full F[16] held in registers compiles to an indexed scratch array (four
// scratch loads), computing just that row from proj's row r does not.
__device__ __forceinline__ void final_row(const float pr[4], const float M[16], float Fr[4])
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) a = __fmaf_rn(pr[k], M[k * 4 + c], a);
        Fr[c] = a;
    }
}

// one vertex per lane on the matrix core: four v_mfma_f32_4x4x1_16b_f32 with A = row (lane%4) of final and
// B = p[k] (same lane mapping and k-ordered accumulation as xfm.hip / the oracle's fmaf chain).  Must be executed by all
// 64 lanes of a wave.
__device__ __forceinline__ f32x4 xfm_vertex_mfma(const float Fr[4] /* row lane % 4 of final */, float px, float py, float pz)
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(Fr[0], px, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(Fr[1], py, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(Fr[2], pz, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(Fr[3], 1.0f, acc, 0, 0, 0);
    return acc;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgnf(float x) { return (float)((x > 0.f) - (x < 0.f)); }

struct PixAcc {
    float dF[12];  // rows x,y,w of d loss / d final
    float dM2[4];  // d loss / d mtx[2][:]
    float L[4];    // rgb, depth, mask, edge loss sums (actual - background)
};

__device__ __forceinline__ void acc_vertex(PixAcc& A, const float* __restrict__ pos, int v, float gx, float gy, float gw)
{
    const float x = pos[(size_t)v * 3 + 0], y = pos[(size_t)v * 3 + 1], z = pos[(size_t)v * 3 + 2];
    A.dF[0] = __fmaf_rn(gx, x, A.dF[0]); A.dF[1] = __fmaf_rn(gx, y, A.dF[1]); A.dF[2] = __fmaf_rn(gx, z, A.dF[2]); A.dF[3] += gx;
    A.dF[4] = __fmaf_rn(gy, x, A.dF[4]); A.dF[5] = __fmaf_rn(gy, y, A.dF[5]); A.dF[6] = __fmaf_rn(gy, z, A.dF[6]); A.dF[7] += gy;
    A.dF[8] = __fmaf_rn(gw, x, A.dF[8]); A.dF[9] = __fmaf_rn(gw, y, A.dF[9]); A.dF[10] = __fmaf_rn(gw, z, A.dF[10]); A.dF[11] += gw;
}

// rows x, y, w of final . [p; 1], accumulated over k = 0..3 from zero like the matrix core does (xfm_vertex): same bits
__device__ __forceinline__ float4 clip_xyw(const float (&Fx)[4], const float (&Fy)[4], const float (&Fw)[4], float x, float y, float z)
{
    float4 o;
    o.x = __fmaf_rn(Fx[3], 1.0f, __fmaf_rn(Fx[2], z, __fmaf_rn(Fx[1], y, __fmaf_rn(Fx[0], x, 0.f))));
    o.y = __fmaf_rn(Fy[3], 1.0f, __fmaf_rn(Fy[2], z, __fmaf_rn(Fy[1], y, __fmaf_rn(Fy[0], x, 0.f))));
    o.z = 0.f;  // (unused by the colour role)
    o.w = __fmaf_rn(Fw[3], 1.0f, __fmaf_rn(Fw[2], z, __fmaf_rn(Fw[1], y, __fmaf_rn(Fw[0], x, 0.f))));
    return o;
}

__device__ __forceinline__ void acc_vertex_regs(PixAcc& A, float x, float y, float z, float gx, float gy, float gw)
{
    A.dF[0] = __fmaf_rn(gx, x, A.dF[0]); A.dF[1] = __fmaf_rn(gx, y, A.dF[1]); A.dF[2] = __fmaf_rn(gx, z, A.dF[2]); A.dF[3] += gx;
    A.dF[4] = __fmaf_rn(gy, x, A.dF[4]); A.dF[5] = __fmaf_rn(gy, y, A.dF[5]); A.dF[6] = __fmaf_rn(gy, z, A.dF[6]); A.dF[7] += gy;
    A.dF[8] = __fmaf_rn(gw, x, A.dF[8]); A.dF[9] = __fmaf_rn(gw, y, A.dF[9]); A.dF[10] = __fmaf_rn(gw, z, A.dF[10]); A.dF[11] += gw;
}


#ifndef SHADE_GRID
#define SHADE_GRID 512  // shade workgroups per role (slices x hypotheses): both roles resident at once (2 x 512 at 128 registers); with the flag scan paid once per workgroup 512 beats 384 / 640 / 768 / 1024 on cfg2-cfg5
#endif
#ifndef EDGE_GRID
#define EDGE_GRID 1792  // edge_kernel workgroups (hypotheses x slices)
#endif
#ifndef SHADE_MIN_WAVES
#define SHADE_MIN_WAVES 4  // waves per SIMD the shade kernel is compiled for (128 VGPRs)
#endif
#define SCAN_LIST 256      // ... tiles of one workgroup held in LDS at a time (a power of two)
#define QUAD 8             // one wave shades one 8x8 quadrant of a 16x16 tile
#define QH (QUAD + 2)      // quadrant + 1-pixel halo
#define PAIR_CAP 160       // >= 2*64 + 8 + 8 candidate antialias pairs per quadrant
#define WAVES_PER_TILE 4

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS operations of one wave execute in issue order; this only stops the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// decode candidate pair `desc` = pixel lane | kind << 6 of the quadrant at (qx,qy):
// kind 0: (p, right)  1: (p, up)  2: (left, p)  3: (down, p).  h0/h1 = halo indices of pixel0 / pixel1.
__device__ __forceinline__ void pair_decode(int desc, int& h0, int& h1, int& d)
{
    const int pl = desc & 63, kind = desc >> 6;
    const int hx = (pl % QUAD) + 1, hy = (pl / QUAD) + 1;
    d = kind & 1;
    const int x0 = kind == 2 ? hx - 1 : hx, y0 = kind == 3 ? hy - 1 : hy;
    h0 = y0 * QH + x0;
    h1 = d ? h0 + QH : h0 + 1;
}

// Antialias pair analysis for the fused engine (semantics = aa_eval_pair + aa_pair_backward of raster_math.h).
// Differences in mechanics only: the triangle record (3 vertex ids + 3 opposite-vertex ids, packed once at
// setup) comes with two 16-byte loads; the clip positions of the 3 vertices, of the 3 opposite vertices and the
// object-space positions are all requested in ONE dependent level; and because the backward is linear in
// d loss / d alpha, the pair's contribution to d loss / d final PER UNIT d alpha (12 numbers) is produced here,
// so the backward pass after the pixel phase is 12 FMAs with no memory access.
struct AAUnit {
    bool valid, clamped;
    float alpha;
    bool target0;   // the contribution lands on pixel0 (else pixel1)
    float C[12];    // d(final rows x,y,w) per unit d alpha (zero if clamped)
};

// silhouette flags (bit k = edge k) of a triangle cut by the eye plane; out of line and with its own loads (L2 hits): the rare path
// must not cost the mask role registers
__device__ __attribute__((noinline)) static int aa_sil_straddler(const float* __restrict__ P, int v0, int v1, int v2, int o0, int o1, int o2)
{
    const float4 p0 = ld4(P + (size_t)v0 * 4), p1 = ld4(P + (size_t)v1 * 4), p2 = ld4(P + (size_t)v2 * 4);
    const float D = aa_det3_xyw(p0, p1, p2);
    int m = 7;
    if (o0 >= 0) { const float4 q = ld4(P + (size_t)o0 * 4); if (q.w > 0.f && sign_bit(aa_det3_xyw(q, p1, p2)) != sign_bit(D)) m &= ~1; }
    if (o1 >= 0) { const float4 q = ld4(P + (size_t)o1 * 4); if (q.w > 0.f && sign_bit(aa_det3_xyw(q, p2, p0)) != sign_bit(D)) m &= ~2; }
    if (o2 >= 0) { const float4 q = ld4(P + (size_t)o2 * 4); if (q.w > 0.f && sign_bit(aa_det3_xyw(q, p0, p1)) != sign_bit(D)) m &= ~4; }
    return m;
}

__device__ __forceinline__ void aa_eval_unit(const float* __restrict__ P, const int4* __restrict__ rec, const float* __restrict__ pos,
                                             int H, int W, int px, int py, int d, int t0, int t1, AAUnit& o)
{
    o.valid = false; o.clamped = true; o.alpha = 0.f; o.target0 = true;
#pragma unroll
    for (int i = 0; i < 12; ++i) o.C[i] = 0.f;
    if (t0 == t1) return;
    const bool chosen1 = t0 < 0;  // exactly one side is covered on this path
    const int t = chosen1 ? t1 : t0;
    const int cx = chosen1 ? px + (d == 0) : px, cy = chosen1 ? py + (d == 1) : py;
    const float ds = chosen1 ? -1.0f : 1.0f;
    const int4 r0 = rec[t * 2 + 0], r1 = rec[t * 2 + 1];  // {v0,v1,v2,opp0} {opp1,opp2,-,-}
    const int vi[3] = {r0.x, r0.y, r0.z};
    const int ov[3] = {r0.w, r1.x, r1.y};
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    const float fx = (float)cx + 0.5f - hw, fy = (float)cy + 0.5f - hh;
    // (recomputing the six clip-space vertices from the object-space table and the hypothesis' matrix instead -- so that step_kernel
    // need not store clip at all -- was built and measured: 12 more uniform registers in this role, 34 spills, shade +1.1 us on
    // cfg2, and step_kernel did not get faster without its 24 MB of stores: they are off its critical path)
    float4 p[3], q[3];
    float ps[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        p[i] = ld4(P + (size_t)vi[i] * 4);
        q[i] = ld4(P + (size_t)(ov[i] >= 0 ? ov[i] : vi[i]) * 4);
        ps[i][0] = pos[(size_t)vi[i] * 3 + 0]; ps[i][1] = pos[(size_t)vi[i] * 3 + 1]; ps[i][2] = pos[(size_t)vi[i] * 3 + 2];
    }
    // (every component of the six vertices is wanted HERE: left alone the compiler narrows the 16-byte loads -- w first, for the
    // eye-plane test, x y z behind it, the opposite vertices behind that -- into three dependent round trips; cfg2 38.25 -> 37.83 us)
    asm volatile("" : "+v"(p[0].x), "+v"(p[0].y), "+v"(p[0].z), "+v"(p[0].w), "+v"(p[1].x), "+v"(p[1].y), "+v"(p[1].z), "+v"(p[1].w),
                      "+v"(p[2].x), "+v"(p[2].y), "+v"(p[2].z), "+v"(p[2].w));
    asm volatile("" : "+v"(q[0].x), "+v"(q[0].y), "+v"(q[0].z), "+v"(q[0].w), "+v"(q[1].x), "+v"(q[1].y), "+v"(q[1].z), "+v"(q[1].w),
                      "+v"(q[2].x), "+v"(q[2].y), "+v"(q[2].z), "+v"(q[2].w));
    asm volatile("" : "+v"(ps[0][0]), "+v"(ps[0][1]), "+v"(ps[0][2]), "+v"(ps[1][0]), "+v"(ps[1][1]), "+v"(ps[1][2]), "+v"(ps[2][0]), "+v"(ps[2][1]), "+v"(ps[2][2]));
    float x[3], y[3], ox[3], oy[3], iw[3];
    bool behind[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        behind[i] = !(p[i].w > 0.f);
        iw[i] = behind[i] ? 0.f : __fdiv_rn(1.0f, p[i].w);
        x[i] = __fmaf_rn(p[i].x * iw[i], hw, -fx);  // (unused for a vertex behind the eye plane)
        y[i] = __fmaf_rn(p[i].y * iw[i], hh, -fy);
    }
    const bool straddler = behind[0] || behind[1] || behind[2];
    if (behind[0] && behind[1] && behind[2]) return;
    bool sil[3];
    if (!straddler) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ox[k] = x[k]; oy[k] = y[k];
            if (ov[k] >= 0 && q[k].w > 0.f) {
                const float iwq = __fdiv_rn(1.0f, q[k].w);
                ox[k] = __fmaf_rn(q[k].x * iwq, hw, -fx);
                oy[k] = __fmaf_rn(q[k].y * iwq, hh, -fy);
            }
        }
        const float bb = (x[1] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (y[1] - y[0]);
        float aw[3];
        aw[0] = (x[1] - ox[0]) * (y[2] - oy[0]) - (x[2] - ox[0]) * (y[1] - oy[0]);
        aw[1] = (x[2] - ox[1]) * (y[0] - oy[1]) - (x[0] - ox[1]) * (y[2] - oy[1]);
        aw[2] = (x[0] - ox[2]) * (y[1] - oy[2]) - (x[1] - ox[2]) * (y[0] - oy[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) sil[k] = sign_bit(aw[k]) == sign_bit(bb);
    } else {
        // a triangle cut by the eye plane (oracle aa_eval_pair, raster_math.h aa_eval_pair): homogeneous orientation tests; only
        // the edges with both endpoints in front can be the crossed edge
        const int m = aa_sil_straddler(P, vi[0], vi[1], vi[2], ov[0], ov[1], ov[2]);
        sil[0] = (m & 1) != 0; sil[1] = (m & 2) != 0; sil[2] = (m & 4) != 0;
    }
    if (!(sil[0] || sil[1] || sil[2])) return;
    if (d) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { const float tmp = x[i]; x[i] = y[i]; y[i] = tmp; }
    }
    int best = -1;
    float rbest = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int ia = (k + 1) % 3, ib = (k + 2) % 3;
        if (behind[ia] || behind[ib]) continue;
        if (sign_bit(y[ia]) == sign_bit(y[ib])) continue;
        const float dx = x[ib] - x[ia], dy = y[ib] - y[ia];
        const float r = ds * __fdiv_rn(x[ia] * dy - y[ia] * dx, dy);
        if (best < 0 || r > rbest) { best = k; rbest = r; }
    }
    if (best < 0) return;
    const int ia = (best + 1) % 3, ib = (best + 2) % 3;
#define SEL3(arr, i) ((i) == 0 ? arr[0] : ((i) == 1 ? arr[1] : arr[2]))
    const float xa = SEL3(x, ia), ya = SEL3(y, ia), xb = SEL3(x, ib), yb = SEL3(y, ib);
    const bool silb = SEL3(sil, best);
    const float dx = xb - xa, dy = yb - ya;
    if (!(silb && fabsf(dy) >= fabsf(dx))) return;
    const float eps = 0.0625f;
    if (!(rbest > -eps && rbest < 1.0f + eps)) return;
    o.valid = true;
    o.clamped = !(rbest > 0.f && rbest < 1.f);
    const float dcc = rbest < 0.f ? 0.f : (rbest > 1.f ? 1.f : rbest);
    o.alpha = ds * (0.5f - dcc);
    o.target0 = o.alpha > 0.f;
    if (o.clamped) return;
    // unit backward (galpha = 1): same expressions as aa_pair_backward
    const float gr = -1.0f;
    const float D = yb - ya;
    const float r = __fdiv_rn(xa * yb - ya * xb, D);
    // (one reciprocal and four multiplications instead of the four divisions: measured in round 5, 38.0 -> 38.1 us -- the role does not
    // feel its divisions)
    const float g_xa = gr * yb / D, g_xb = gr * (-ya) / D;
    const float g_ya = gr * (r - xb) / D, g_yb = gr * (xa - r) / D;
    const float gX[2] = {d ? g_ya : g_xa, d ? g_yb : g_xb};
    const float gY[2] = {d ? g_xa : g_ya, d ? g_xb : g_yb};
    const float ix[2] = {d ? ya : xa, d ? yb : xb};
    const float iy[2] = {d ? xa : ya, d ? xb : yb};
    const float iwa = SEL3(iw, ia), iwb = SEL3(iw, ib);
    const float iwv[2] = {iwa, iwb};
    const float pa[3] = {SEL3(ps, ia)[0], SEL3(ps, ia)[1], SEL3(ps, ia)[2]};
    const float pb[3] = {SEL3(ps, ib)[0], SEL3(ps, ib)[1], SEL3(ps, ib)[2]};
#undef SEL3
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float gx = gX[i] * 0.5f * (float)W * iwv[i];
        const float gy = gY[i] * 0.5f * (float)H * iwv[i];
        const float gw = -(gX[i] * (ix[i] + fx) + gY[i] * (iy[i] + fy)) * iwv[i];
        const float* pp = i == 0 ? pa : pb;
        o.C[0] = __fmaf_rn(gx, pp[0], o.C[0]); o.C[1] = __fmaf_rn(gx, pp[1], o.C[1]); o.C[2] = __fmaf_rn(gx, pp[2], o.C[2]); o.C[3] += gx;
        o.C[4] = __fmaf_rn(gy, pp[0], o.C[4]); o.C[5] = __fmaf_rn(gy, pp[1], o.C[5]); o.C[6] = __fmaf_rn(gy, pp[2], o.C[6]); o.C[7] += gy;
        o.C[8] = __fmaf_rn(gw, pp[0], o.C[8]); o.C[9] = __fmaf_rn(gw, pp[1], o.C[9]); o.C[10] = __fmaf_rn(gw, pp[2], o.C[10]); o.C[11] += gw;
    }
}

// A hypothesis' tile-flag row (shade_kernel), scanned by ONE wave in pieces of 1 KB = 1024 flags: lane l loads 16 bytes -- the flags
// of 16 consecutive tiles -- so a piece is one coalesced request of 8 lines (lane-contiguous runs of 64 bytes were measured first:
// 16 strided dword loads per lane made the scan 5 us of the kernel, one L1 line per cycle and CU), turns them into a 16-bit mask,
// and the exclusive prefix of the lanes' counts comes from five ballots (no LDS).  Four pieces are in flight per trip.  The tiles
// whose rank is sl mod S are this slice's: those numbered [win, win + SCAN_LIST) among them go to list[] (ty << 8 | tx); g_active
// (nullable) receives the whole ordered list.  Returns the number of set flags of the row.  Kept out of line: inlined, its loop
// nest around the shading loop cost the kernel 34 spilled registers.
__device__ __attribute__((noinline)) static int tile_scan(const unsigned* __restrict__ frow, int n_dw, int S, int sl, float invS, int ntx, int win,
                                                          unsigned short* list, int* g_active)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    int rank_base = 0;
    for (int d0 = 0; d0 < n_dw; d0 += 4 * 256) {  // (wave-uniform; one trip up to 4096 tiles)
        uint4 dd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int di = d0 + u * 256 + lane * 4;
            dd[u] = di < n_dw ? *reinterpret_cast<const uint4*>(frow + di) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (d0 + u * 256 >= n_dw) break;  // (wave-uniform)
            auto nib = [](unsigned w) { return ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u) | ((w & 0xff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u); };
            unsigned fm = nib(dd[u].x) | (nib(dd[u].y) << 4) | (nib(dd[u].z) << 8) | (nib(dd[u].w) << 12);
            const int cnt = __popc(fm);  // <= 16
            int pre = 0, tot = 0;
#pragma unroll
            for (int bit = 0; bit < 5; ++bit) {
                const unsigned long long m = __ballot((cnt >> bit) & 1);
                pre += __popcll(m & below) << bit;
                tot += __popcll(m) << bit;
            }
            if (tot == 0) continue;  // (wave-uniform)
            int rank = rank_base + pre;
            const int fbase = (d0 + u * 256 + lane * 4) * 4;  // tile index of the lane's first flag
            while (fm) {  // (a few set flags on a few lanes: the object's tiles)
                const int kbit = __ffs(fm) - 1;
                fm &= fm - 1;
                const int q = (int)(((float)rank + 0.5f) * invS);  // rank / S, exact for rank < 2^16
                if (rank - q * S == sl) {
                    const int tile = fbase + kbit;
                    const int ty = tile / ntx, tx = tile - ty * ntx;
                    if (q >= win && q < win + SCAN_LIST) list[q - win] = (unsigned short)((ty << 8) | tx);
                    if (g_active) g_active[rank] = (ty << 16) | tx;
                }
                ++rank;
            }
            rank_base += tot;
        }
    }
    return rank_base;
}

// ROLE 0: colour + depth terms (per covered pixel).  ROLE 1: antialiased-coverage (mask) term (silhouette
// pairs).  The two roles only share the zbuf they read, so they are separate workgroups of ONE launch
// (blockIdx.z picks the role): they overlap on the chip, and each body keeps its own, smaller register
// footprint instead of the union of both.  Each role writes its own partial per quadrant.
// The edge term (extension) is a separate, texture-free kernel (edge_kernel below): in the edge build the colour role
// also writes the luminance and U = d lum / d final of every covered pixel.

// `pool`: per-wave LDS scratch owned by the kernel (one allocation shared by the roles, which are different
// workgroups): 12 x 64 floats for the mask role, 24 x 64 for the edge role.
// what a wave knows after scanning its hypothesis' tile flags (shade_kernel)
struct TileWork {
    int b, par, sl, S;         // hypothesis, iteration parity, slice and slices per hypothesis
    int n_flags, n_mine;       // active tiles of the hypothesis, and of this slice
    const unsigned* frow;      // the flag row (for further windows of the list)
    int n_dw;
    float invS;
    unsigned short* list;      // LDS, this wave's: tiles [win, win + SCAN_LIST) of the slice, ty << 8 | tx
};

template <int ROLE, bool WLUM /* edge build: the colour role also feeds the edge term */>
__device__ __forceinline__ void shade_body(const EngineDev& E, float* __restrict__ pool, const TileWork& tw)
{
    __shared__ int s_ids[WAVES_PER_TILE][QH * QH + 4];  // zbuf id + 1 (0 = background), -1 = outside the image
    __shared__ float s_m[WAVES_PER_TILE][64];           // antialias contributions received by each pixel
    __shared__ float s_gm[WAVES_PER_TILE][64];          // d loss / d mask of each pixel
    __shared__ unsigned short s_pairs[WAVES_PER_TILE][ROLE == 1 ? PAIR_CAP : 4];
    const ddx_engine_desc& d = E.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = d.H, W = d.W, V = d.V;
    const RasterScratch& L = E.L;
    const float* __restrict__ pos = E.spos;
    int* ids = s_ids[wave];
    const int b = tw.b, par = tw.par;
    // colour role: rows x, y, w of this hypothesis' final = proj . mtx (uniform: scalar loads)
    float Fx[4] = {0.f, 0.f, 0.f, 0.f}, Fy[4] = {0.f, 0.f, 0.f, 0.f}, Fw[4] = {0.f, 0.f, 0.f, 0.f};
    if (ROLE == 0) {
        const float* Fm = E.mats + ((size_t)par * d.B + b) * 32 + 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) { Fx[c] = *(Fm + c); Fy[c] = *(Fm + 4 + c); Fw[c] = *(Fm + 12 + c); }
    }
    // every lane accumulates its pixels' terms over ALL the tiles of the workgroup; one wave reduction, one fold over the four
    // waves and one partial row per (slice, role) at the end (update_head sums min(slices, tiles) rows per role)
    PixAcc A;
#pragma unroll
    for (int i = 0; i < 12; ++i) A.dF[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) A.dM2[i] = 0.f;
    A.L[0] = A.L[1] = A.L[2] = A.L[3] = 0.f;
    const float lrb = E.b.lr_mult[b];
    const float inv_b = __fdiv_rn(1.0f, (float)d.B_global);
    const int n_mine = tw.n_mine;
    for (int k = 0; k < n_mine; ++k) {
        if ((k & (SCAN_LIST - 1)) == 0 && k > 0) {  // further windows (a frame-filling object with few slices) on demand
            wave_lds_sync();
            tile_scan(tw.frow, tw.n_dw, tw.S, tw.sl, tw.invS, L.ntx, k, tw.list, nullptr);
            wave_lds_sync();
        }
        const int txy8 = tw.list[k & (SCAN_LIST - 1)];
        const int tcx = txy8 & 0xff, tcy = txy8 >> 8;
        const int qx = tcx * DDX_TILE + (wave & 1) * QUAD, qy = tcy * DDX_TILE + (wave >> 1) * QUAD;
        const float* __restrict__ P = E.clip + (size_t)b * V * 4;
        const unsigned long long* __restrict__ zb = L.zbuf + ((size_t)par * d.B + b) * L.zper;
        const int lx = lane % QUAD, ly = lane / QUAD;
        const int px = qx + lx, py = qy + ly;
        const int hidx = (ly + 1) * QH + lx + 1;
        // the observed segmentation of the own pixel only depends on the pixel: requested WITH the zbuf entries, not after
        // them (for a quadrant without silhouette pairs the chain was zbuf -> seg -> loss)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        const size_t pix = (size_t)py * W + px;
        if (px < W && py < H) { s0 = E.b.gt_seg[pix * 3 + 0]; s1 = E.b.gt_seg[pix * 3 + 1]; s2 = E.b.gt_seg[pix * 3 + 2]; }
        int id;
        if (ROLE == 0) {
            // the colour / depth role only needs its own pixel: one zbuf entry per lane, no LDS staging
            id = -1;
            // (requesting the NEXT tile's entry while this one is shaded -- one dependent level less per tile -- measured in round 5:
            // cfg2 shade 18.2 -> 19.0 us, 4.7 % coverage 38.3 -> 40.0: 13 spilled registers instead of 10, and the other waves of the SIMD
            // already cover the round trip)
            if (px < W && py < H) {
                const unsigned long long key = *(zb + zaddr(px, py, L.zwb));
                id = key == ~0ull ? 0 : (int)(unsigned)(key & 0xffffffffull) + 1;
            }
            if (__ballot(id > 0) == 0ull) continue;  // nothing drawn in this quadrant: only background terms
        } else {
            // ---- stage the 10x10 id halo of this quadrant (zbuf is all ones wherever nothing was drawn).  (Requesting the NEXT
            // tile's halo entries while this one is shaded -- for workgroups that walk many tiles: an object filling a good part of
            // the frame, where this role is the longer one -- was built and measured in round 3: +-0 at 4.7 % coverage, -5 % at 21 %:
            // the other waves of the SIMD already cover the round trip; the texel gathers' DRAM traffic is the limit there.)
            bool anycov = false;
#pragma unroll
            for (int e = lane; e < QH * QH; e += 64) {
                const int gx = qx - 1 + e % QH, gy = qy - 1 + e / QH;
                int v = -1;
                if (gx >= 0 && gy >= 0 && gx < W && gy < H) {
                    const unsigned long long key = *(zb + zaddr(gx, gy, L.zwb));
                    v = key == ~0ull ? 0 : (int)(unsigned)(key & 0xffffffffull) + 1;
                }
                ids[e] = v;
                anycov |= v > 0;
            }
            s_m[wave][lane] = 0.f;
            if (__ballot(anycov) == 0ull) continue;  // nothing drawn in or next to this quadrant: only background terms
            wave_lds_sync();
            id = ids[hidx];
        }
        if (ROLE == 0 && id > 0) {
            const int t = id - 1;
            // ONE record per covered triangle, fetched by id: object-space positions + uv (or colours) of its vertices.  The
            // clip-space vertices are recomputed from the hypothesis' matrix (uniform, in SGPRs) with the k-ordered fma chain
            // the matrix core ran in the transform kernel -- the same bits -- instead of being gathered: the chain
            // zbuf -> indices -> vertices -> texels loses a level, and 18 gathers become 4 loads of a static, shared table.
            const float4* R = E.crec + (size_t)t * (d.Th > 0 ? 4 : 5);
            const float4 r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3];
            float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(d.Th > 0)) r4 = R[4];
            const float x0 = r0.x, y0 = r0.y, z0 = r0.z, x1 = r0.w, y1 = r1.x, z1 = r1.y, x2 = r1.z, y2 = r1.w, z2 = r2.x;
            const float a0x = r2.y, a0y = r2.z, a1x = r2.w, a1y = r3.x, a2x = r3.y, a2y = r3.z;  // (textured)
            const float4 p0 = clip_xyw(Fx, Fy, Fw, x0, y0, z0), p1 = clip_xyw(Fx, Fy, Fw, x1, y1, z1), p2 = clip_xyw(Fx, Fy, Fw, x2, y2, z2);
            Bary bc;
            pixel_bary(p0, p1, p2, px, py, H, W, bc);
            const float u = clamp01(bc.u), v = clamp01(bc.v), w2 = (1.0f - u) - v;
            float gu = 0.f, gv = 0.f;
            // colour of the pixel and its derivatives w.r.t. the barycentrics (u, v), per channel; used by the rgb term
            // and -- in the edge build -- written out as luminance + unit gradient for edge_kernel
            if (d.use_rgb || WLUM) {
                float col[3], dcu[3], dcv[3];
                if (d.Th > 0) {
                    const float tu = __fmaf_rn(w2, a2x, __fmaf_rn(v, a1x, u * a0x));
                    const float tv = __fmaf_rn(w2, a2y, __fmaf_rn(v, a1y, u * a0y));
                    TexelSetup ts;
                    tex_setup(tu, tv, d.Th, d.Tw, ts);
                    // one 64-byte record = the whole 2x2 footprint (texq): (x0,y0) (x1,y0) (x0,y1) (x1,y1), rgb each
#ifdef DDX_EXP_NO_TEXEL  // (leave-out measurement: every sample reads the SAME record -- an L1 hit -- instead of its own)
                    const float4* Q = E.texq + ((size_t)(ts.y0 & 0) * d.Tw + (ts.x0 & 1)) * 4;
#else
                    const float4* Q = E.texq + ((size_t)ts.y0 * d.Tw + ts.x0) * 4;
#endif
                    const float4 q0 = Q[0], q1 = Q[1], q2 = Q[2];
                    const float t00[3] = {q0.x, q0.y, q0.z}, t10[3] = {q0.w, q1.x, q1.y}, t01[3] = {q1.z, q1.w, q2.x}, t11[3] = {q2.y, q2.z, q2.w};
                    const float ux = (a0x - a2x) * (float)d.Tw, uy = (a0y - a2y) * (float)d.Th;  // d(texel x, y) / du
                    const float vx = (a1x - a2x) * (float)d.Tw, vy = (a1y - a2y) * (float)d.Th;  // d(texel x, y) / dv
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float c00 = t00[c], c10 = t10[c], c01 = t01[c], c11 = t11[c];
                        const float a = __fmaf_rn(ts.fx, c10 - c00, c00);
                        const float bq = __fmaf_rn(ts.fx, c11 - c01, c01);
                        col[c] = __fmaf_rn(ts.fy, bq - a, a);
                        const float dX = __fmaf_rn(ts.fy, (c11 - c01) - (c10 - c00), c10 - c00);  // d col / d texel x
                        const float dY = __fmaf_rn(ts.fx, (c11 - c10) - (c01 - c00), c01 - c00);  // d col / d texel y
                        dcu[c] = dX * ux + dY * uy;
                        dcv[c] = dX * vx + dY * vy;
                    }
                } else {
                    const float k0[3] = {r2.y, r2.z, r2.w}, k1[3] = {r3.x, r3.y, r3.z}, k2[3] = {r3.w, r4.x, r4.y};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float c0 = k0[c], c1 = k1[c], c2 = k2[c];
                        col[c] = __fmaf_rn(w2, c2, __fmaf_rn(v, c1, u * c0));
                        dcu[c] = c0 - c2;
                        dcv[c] = c1 - c2;
                    }
                }
                if (d.use_rgb) {
                    const float k = d.w_rgb * lrb * inv_b / (3.0f * (float)H * (float)W);
                    // (requested here, behind the texel record: asked for with the zbuf entry -- one dependent level less -- the three
                    // values live across the whole chain and the role spills 14 registers instead of 10: 38.2 -> 39.0 us, measured)
                    const float gt[3] = {E.b.gt_rgb[pix * 3 + 0], E.b.gt_rgb[pix * 3 + 1], E.b.gt_rgb[pix * 3 + 2]}, sg[3] = {s0, s1, s2};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float diff = (col[c] - gt[c]) * sg[c];
                        A.L[0] += fabsf(diff) - fabsf(gt[c] * sg[c]);
                        const float g = k * sgnf(diff) * sg[c];
                        gu = __fmaf_rn(g, dcu[c], gu);
                        gv = __fmaf_rn(g, dcv[c], gv);
                    }
                }
                if (WLUM && E.lumbuf) {
                    // luminance and U = d lum / d final (rows x, y, w) per unit d loss / d lum: the backward of the edge
                    // term is linear in d loss / d lum, so edge_kernel only multiplies
                    const float third = 1.0f / 3.0f;
                    const float gu2 = third * ((dcu[0] + dcu[1]) + dcu[2]), gv2 = third * ((dcv[0] + dcv[1]) + dcv[2]);
                    float gx[3], gy[3], gw[3];
                    bary_backward(bc, gu2, gv2, gx, gy, gw, (d.compat & DDX_COMPAT_UNCLAMPED_BARY_GRAD) != 0);
                    PixAcc T;
#pragma unroll
                    for (int i = 0; i < 12; ++i) T.dF[i] = 0.f;
                    acc_vertex_regs(T, x0, y0, z0, gx[0], gy[0], gw[0]);
                    acc_vertex_regs(T, x1, y1, z1, gx[1], gy[1], gw[1]);
                    acc_vertex_regs(T, x2, y2, z2, gx[2], gy[2], gw[2]);
                    const size_t gp = (size_t)b * H * W + pix;
                    E.lumbuf[gp] = lum3(col[0], col[1], col[2]);
                    float4* ub = reinterpret_cast<float4*>(E.ubuf + gp * 12);
                    ub[0] = make_float4(T.dF[0], T.dF[1], T.dF[2], T.dF[3]);
                    ub[1] = make_float4(T.dF[4], T.dF[5], T.dF[6], T.dF[7]);
                    ub[2] = make_float4(T.dF[8], T.dF[9], T.dF[10], T.dF[11]);
                }
            }
            if (d.use_depth) {
                const float k = d.w_depth * lrb * inv_b / ((float)H * (float)W);
                const float* M = E.mats + ((size_t)par * d.B + b) * 32;
                const float m20 = *(M + 8), m21 = *(M + 9), m22 = *(M + 10), m23 = *(M + 11);
                const float gbx = __fmaf_rn(w2, x2, __fmaf_rn(v, x1, u * x0));
                const float gby = __fmaf_rn(w2, y2, __fmaf_rn(v, y1, u * y0));
                const float gbz = __fmaf_rn(w2, z2, __fmaf_rn(v, z1, u * z0));
                float zc = __fmaf_rn(m20, gbx, 0.f);
                zc = __fmaf_rn(m21, gby, zc);
                zc = __fmaf_rn(m22, gbz, zc);
                zc = __fmaf_rn(m23, 1.0f, zc);
                const float depth = -zc, dbg = -m23;
                const float gtd = E.b.gt_depth[pix];
                const float diff = (depth - gtd) * s0, dbase = (dbg - gtd) * s0;
                A.L[1] += fabsf(diff) - fabsf(dbase);
                const float g = k * sgnf(diff) * s0;  // d loss / d depth
                A.dM2[0] += -g * gbx; A.dM2[1] += -g * gby; A.dM2[2] += -g * gbz;
                A.dM2[3] += -g + k * sgnf(dbase) * s0;  // actual term and the subtracted background term
                gu += -g * (m20 * (x0 - x2) + m21 * (y0 - y2) + m22 * (z0 - z2));
                gv += -g * (m20 * (x1 - x2) + m21 * (y1 - y2) + m22 * (z1 - z2));
            }
            if (gu != 0.f || gv != 0.f) {
                float gx[3], gy[3], gw[3];
                bary_backward(bc, gu, gv, gx, gy, gw, (d.compat & DDX_COMPAT_UNCLAMPED_BARY_GRAD) != 0);
                acc_vertex_regs(A, x0, y0, z0, gx[0], gy[0], gw[0]);
                acc_vertex_regs(A, x1, y1, z1, gx[1], gy[1], gw[1]);
                acc_vertex_regs(A, x2, y2, z2, gx[2], gy[2], gw[2]);
            }
        }
        if (ROLE == 1) {
            // ---- antialias, pair-parallel: compact the candidate pairs (exactly one side covered, at least
            // one side in this quadrant) with ballots, then ONE lane per pair instead of 4 divergent
            // neighbour probes per pixel.  A pair deposits alpha*(c1-c0) into its target pixel (LDS add);
            // pairs whose target lies in another quadrant are evaluated there as well and dropped here.
            const float k = d.w_mask * lrb * inv_b / (3.0f * (float)H * (float)W);
            const bool inimg = id >= 0;
            const int idr = ids[hidx + 1], idu = ids[hidx + QH], idl = ids[hidx - 1], idd = ids[hidx - QH];
            const bool c0 = inimg && idr >= 0 && ((idr > 0) != (id > 0));
            const bool c1 = inimg && idu >= 0 && ((idu > 0) != (id > 0));
            const bool c2 = inimg && lx == 0 && idl >= 0 && ((idl > 0) != (id > 0));
            const bool c3 = inimg && ly == 0 && idd >= 0 && ((idd > 0) != (id > 0));
            const unsigned long long lt = (1ull << lane) - 1ull;
            const unsigned long long m0 = __ballot(c0), m1 = __ballot(c1), m2 = __ballot(c2), m3 = __ballot(c3);
            const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
            const int np = n0 + n1 + n2 + n3;
            if (c0) s_pairs[wave][__popcll(m0 & lt)] = (unsigned short)(lane | (0 << 6));
            if (c1) s_pairs[wave][n0 + __popcll(m1 & lt)] = (unsigned short)(lane | (1 << 6));
            if (c2) s_pairs[wave][n0 + n1 + __popcll(m2 & lt)] = (unsigned short)(lane | (2 << 6));
            if (c3) s_pairs[wave][n0 + n1 + n2 + __popcll(m3 & lt)] = (unsigned short)(lane | (3 << 6));
            wave_lds_sync();
            // forward: each lane owns pair `lane` (+64, ... in the rare quadrant with more than 64 pairs)
            int tl0 = -1;
            float cd0 = 0.f;
            for (int j0 = 0; j0 < np; j0 += 64) {
                const int j = j0 + lane;
                int tl = -1;
                float cd = 0.f;
                AAUnit pr;
                pr.valid = false;
                if (j < np) {
                    int h0, h1, dd;
                    pair_decode(s_pairs[wave][j], h0, h1, dd);
                    const int t0 = ids[h0] - 1, t1 = ids[h1] - 1;
                    aa_eval_unit(P, E.trirec, pos, H, W, qx - 1 + h0 % QH, qy - 1 + h0 / QH, dd, t0, t1, pr);
                    if (pr.valid) {
                        const int ht = pr.target0 ? h0 : h1;
                        const int tx = ht % QH - 1, ty = ht / QH - 1;
                        cd = (float)((t1 >= 0) - (t0 >= 0));
                        if (tx >= 0 && tx < QUAD && ty >= 0 && ty < QUAD) {
                            tl = ty * QUAD + tx;
                            atomicAdd(&s_m[wave][tl], pr.alpha * cd);
                        }
                    }
                }
                if (j0 == 0) {
                    tl0 = (pr.valid && !pr.clamped) ? tl : -1;
                    cd0 = cd;
#pragma unroll
                    for (int i = 0; i < 12; ++i) pool[i * 64 + lane] = pr.C[i];  // parked in LDS: 12 VGPRs less across the pixel phase
                }
            }
            wave_lds_sync();
            // pixel: mask value, loss term, d loss / d mask
            float gm = 0.f;
            if (inimg) {
                const float m = (float)(id > 0) + s_m[wave][lane];
                const float e0 = m - s0, e1 = m - s1, e2 = m - s2;
                A.L[2] += (fabsf(e0) - fabsf(s0)) + (fabsf(e1) - fabsf(s1)) + (fabsf(e2) - fabsf(s2));
                gm = k * (sgnf(e0) + sgnf(e1) + sgnf(e2));
            }
            s_gm[wave][lane] = gm;
            s_m[wave][lane] = 0.f;  // re-arm for the next tile
            wave_lds_sync();
            // backward: a pair whose target pixel is ours scales its unit contribution by d loss / d alpha
            for (int j0 = 0; j0 < np; j0 += 64) {
                int tl = tl0;
                float cd = cd0;
                float C[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) C[i] = j0 == 0 ? pool[i * 64 + lane] : 0.f;
                if (j0 > 0) {  // rare: re-evaluate the overflow pairs
                    const int j = j0 + lane;
                    tl = -1;
                    if (j < np) {
                        int h0, h1, dd;
                        pair_decode(s_pairs[wave][j], h0, h1, dd);
                        const int t0 = ids[h0] - 1, t1 = ids[h1] - 1;
                        AAUnit pr;
                        aa_eval_unit(P, E.trirec, pos, H, W, qx - 1 + h0 % QH, qy - 1 + h0 / QH, dd, t0, t1, pr);
                        if (pr.valid && !pr.clamped) {
                            const int ht = pr.target0 ? h0 : h1;
                            const int tx = ht % QH - 1, ty = ht / QH - 1;
                            cd = (float)((t1 >= 0) - (t0 >= 0));
                            if (tx >= 0 && tx < QUAD && ty >= 0 && ty < QUAD) tl = ty * QUAD + tx;
#pragma unroll
                            for (int i = 0; i < 12; ++i) C[i] = pr.C[i];
                        }
                    }
                }
                if (tl >= 0) {
                    const float ga = s_gm[wave][tl] * cd;
#pragma unroll
                    for (int i = 0; i < 12; ++i) A.dF[i] = __fmaf_rn(ga, C[i], A.dF[i]);
                }
            }
            wave_lds_sync();
        }
    }
    if (tw.sl < tw.n_flags) {  // (workgroup-uniform: every wave counted the same flags)
        // ---- wave reduction, then the four waves folded in a fixed order -> one partial row per (slice, role): bit-reproducible
        __shared__ float s_rows[WAVES_PER_TILE][NPART];
        float* part = E.partials + (((size_t)b * E.pslices + tw.sl) * NROLE + ROLE) * NPART;
        constexpr int NV = NVALS - 1;  // (the 20th value, the edge loss, belongs to edge_kernel)
        float vals[NVALS];
#pragma unroll
        for (int i = 0; i < 12; ++i) vals[i] = A.dF[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) vals[12 + i] = A.dM2[i];
        vals[16] = A.L[0]; vals[17] = A.L[1]; vals[18] = A.L[2]; vals[19] = A.L[3];
        float mine = 0.f, mine2 = 0.f;
        bool nz = false;
#pragma unroll
        for (int i = 0; i < NV; ++i) nz |= vals[i] != 0.f;
        if (__ballot(nz) != 0ull) {  // e.g. mask role on an interior quadrant: every term is exactly zero
            wave_sum_n_lastrow(vals);  // totals in lanes 48..63
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (lane == 48 + i) mine = vals[i];
#pragma unroll
            for (int i = 16; i < NV; ++i)
                if (lane == 32 + i) mine2 = vals[i];
        }
        if (lane >= 48) {  // lanes 48..63 hold values 0..15, lanes 48..55 the slots 16..23 (values 16..NV-1, then zero padding)
            s_rows[wave][lane - 48] = mine;
            if (lane < 56) s_rows[wave][lane - 32] = mine2;
        }
        __syncthreads();
        if (tid < NPART) part[tid] = (s_rows[0][tid] + s_rows[1][tid]) + (s_rows[2][tid] + s_rows[3][tid]);
    }
}

// Two builds of the same kernel: without and with the edge role, so that the register allocation (and scratch
// footprint) of the reference-loss configurations does not depend on the extension.
// THE TILE PASS INSIDE THE SHADING LAUNCH (EngineDev::big_inline, round 4; made placement-independent in round 5).  big_pass_kernel
// between step_kernel and shade_kernel costs a kernel boundary in every iteration -- 2.4 of cfg2's 43 us, measured by leaving it out
// -- and exits at once in nearly all of them: the dense meshes of the benchmark never have a LARGE triangle.  With big_inline the
// launch is dropped and the shading grid gets one more slab of workgroups IN FRONT (grid z = 0: the "workers", S per hypothesis): a
// worker reads its hypothesis' count of large triangles and leaves if it is zero; otherwise the S workers of the hypothesis split its
// large tiles (a wave per tile, big_tile_wave with a 16-triangle stage), and each adds one to the hypothesis' arrival counter -- behind
// an agent-scope RELEASE -- when its atomics have been performed.  A shading workgroup looks at the same count (one scalar load,
// requested before its flag scan) and, only if it is not zero, waits until all S workers have arrived, then executes an agent-scope
// ACQUIRE (both fences sit on this rare path only: a hypothesis with large triangles on a mesh where none were expected).
// LIVENESS: HIP promises nothing about dispatch order or workgroup -> XCD placement (MI355X_MICROARCH.md).  On this stack workgroups
// are observed to start in the order of their linear id, so a shading workgroup that runs implies that its workers -- smaller ids --
// have started, and the wait is short.  Nothing depends on that: the wait is BOUNDED (EngineDev::wait_ticks of the 100 MHz clock);
// a workgroup whose budget runs out sets ENGINE_FLAG_INLINE_TIMEOUT in the status block and goes on with what the depth buffer
// holds.  Every kernel of the run still terminates and leaves the buffers re-armed, its numbers are void, and the host
// (ddx_engine_run_check, called by whoever synchronises next) restores the parameters and optimiser state the run started from
// (EngineDev::run_snap) and repeats it with the tile pass as its own launch, for good.  Same triangles, same keys, same zbuf as the
// separate launch.  No heavy code in the shading path (the round-3 form -- the shading workgroups running the tile pass themselves
// through a call -- cost the kernel 13 spilled registers and 2.5 us whether or not a large triangle existed).
__device__ __forceinline__ void big_worker_wg(const EngineDev& E, int b, int sl, int S, int it_arg)
{
    S = min(S, E.big_workers);
    if (sl >= S) return;
    const RasterScratch& L = E.L;
    const int par = (it_arg >= 0 ? it_arg : E.st->it_next - 1) & 1;
    const size_t hb = (size_t)par * E.d.B + b;
    const int n_big = __builtin_amdgcn_readfirstlane(L.bigcount[hb]);
    if (n_big <= 0) return;  // (always, on the meshes this mode is chosen for)
    __shared__ BigStage<16> s_stage[WAVES_PER_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned char* tb = L.tile_big + hb * L.NTp;
    const int w = sl * WAVES_PER_TILE + wave, NW = S * WAVES_PER_TILE;  // this wave among the hypothesis' worker waves: tiles w, w + NW, ...
    for (int base = 0; base * NW < L.NT; base += 64) {
        const int tile = (base + lane) * NW + w;
        const unsigned long long m = __ballot(tile < L.NT && tb[tile] != 0);
        for (unsigned long long r = m; r; r &= r - 1) {  // (wave-uniform)
            const int k = __ffsll((long long)r) - 1;
            big_tile_wave<16>(s_stage[wave], E.clip + (size_t)b * E.d.V * 4, E.stri, L.snap + (size_t)b * E.d.V, L.biglist + (size_t)b * E.d.T,
                              min(n_big, E.d.T), L.zbuf + hb * L.zper, L.zwb, L.ntx, (base + k) * NW + w, E.d.H, E.d.W);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);  // this wave's atomics have been performed
    __syncthreads();
    if (tid == 0) {
        // (the depth keys went out as agent-scope atomics; the release covers whatever else a future worker may store)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the wait behind buffer_wbl2: MI355X_MICROARCH.md "Compiler hazard")
        __hip_atomic_fetch_add(L.bigarrive + hb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// a shading workgroup of a hypothesis that HAS large triangles: until its S workers have arrived, or until the budget is spent
// (then: the status flag, and on with whatever zbuf holds -- the run is repeated by the host, see above).  Out of line: rare path.
__device__ __attribute__((noinline)) static void big_wait(int* arrive, int S, unsigned budget, int* flags)
{
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool ok = true;
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < S) {
            __builtin_amdgcn_s_sleep(8);
            if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)budget) { ok = false; break; }
        }
        if (!ok) __hip_atomic_fetch_or(flags, ENGINE_FLAG_INLINE_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's L1 holds nothing older than the workers' arrival
    }
    __syncthreads();
}

template <bool EDGE>
__device__ __forceinline__ void shade_wg(const EngineDev& E, int b, int sl, int S, int z, int it_arg)
{
    // workgroup (b, s) takes tiles s, s+S, ... of hypothesis b's active tiles in ascending tile order.  There is no list kernel:
    // every WAVE scans the hypothesis' row of tile flags itself (tile_scan: bytes written by step_kernel, 1.2 KB at 640x480),
    // ranks the set flags with ballots and keeps the tiles whose rank is s mod S -- no atomics, no barrier.  The first wave of the
    // z = 0 workgroups also writes the ordered list and the count for the kernels that come later (edge_kernel, update_head).
    // grid (B, S, z): x = hypothesis, y = slice.  Workgroups are dispatched in linear-id order and land on CUs
    // in a fixed pattern of that id (XCD = id % 8, CU = f(id/8 % 32), measured): slice-major order sends the
    // working slices of every hypothesis first and the idle ones (slice >= n_tiles) last, so the tail of the launch
    // is made of workgroups that exit at once, and every CU sees the same mix of slices.
    // z = role (the mask role first: it takes the cold misses on zbuf).  (One workgroup running the mask role and then the colour
    // role of its tiles -- one scan, half the workgroups -- was measured: the two roles' tile loops simply add up, 21 -> 21 us.)
    __shared__ float s_pool[WAVES_PER_TILE][12 * 64];
    __shared__ unsigned short s_list[WAVES_PER_TILE][SCAN_LIST];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* pool = s_pool[wave];
    const RasterScratch& L = E.L;
    TileWork tw;
    tw.b = b;
    const int it_cur = it_arg >= 0 ? it_arg : E.st->it_next - 1;  // the iteration being drawn (it_next is stable during this launch)
    tw.par = it_cur & 1;
    if (z == 0 && sl == 0 && b == 0 && tid == 0) E.st->it = it_cur + 1;  // read by the next step_kernel / finish_kernel
    tw.sl = sl;
    tw.S = S;
    tw.frow = reinterpret_cast<const unsigned*>(L.tile_flag + ((size_t)tw.par * E.d.B + tw.b) * L.NTp);
    tw.n_dw = L.NTp >> 2;  // dwords of the row
    tw.invS = __frcp_rn((float)tw.S);
    tw.list = s_list[wave];
    const bool lister = z == 0 && wave == 0;
    const int wg_id = (z * S + sl) * E.d.B + b;
    STAMP(E, 1, wg_id, 0);
    // (inline tile pass: the hypothesis' count of large triangles, requested before the scan and looked at after it)
    const int n_big_list = E.big_inline ? __builtin_amdgcn_readfirstlane(L.bigcount[(size_t)tw.par * E.d.B + b]) : 0;
    tw.n_flags = tile_scan(tw.frow, tw.n_dw, tw.S, tw.sl, tw.invS, L.ntx, 0, tw.list, lister ? L.active + (size_t)tw.b * L.NT : nullptr);
    tw.n_mine = tw.n_flags > tw.sl ? (tw.n_flags - tw.sl + tw.S - 1) / tw.S : 0;
    wave_lds_sync();
    if (lister && tw.sl == 0 && lane == 0) L.b_count[tw.b] = tw.n_flags;
    if (n_big_list > 0) big_wait(L.bigarrive + (size_t)tw.par * E.d.B + b, min(S, E.big_workers), E.wait_ticks, &E.st->flags);  // (workgroup-uniform; rare)
    STAMP(E, 1, wg_id, 1);
    const int role = z == 0 ? E.roles[0] : E.roles[1];
    if (role == 0) shade_body<0, EDGE>(E, pool, tw);
    else shade_body<1, EDGE>(E, pool, tw);
    STAMP(E, 1, wg_id, 2);
    STAMP(E, 1, wg_id, 3);
    if (tid == 0 && E.trace && wg_id < TRACE_WG) E.trace[((size_t)TRACE_WG + wg_id) * 8 + 4] = ((unsigned long long)z << 32) | (unsigned)tw.n_mine;
}

// (HALF: the same kernel under a second name, for the half-batch launches of a two-stream run -- a profile lists them apart)
template <bool EDGE, bool HALF = false>
__global__ __launch_bounds__(256, SHADE_MIN_WAVES) void shade_kernel(EngineDev E, int it_arg)
{
    // grid (B, S, roles), or (B, S, 1 + roles) with the tile pass inside the launch: slab z = 0 = its workers
    int z = blockIdx.z;
    if (E.big_inline) {
        if (E.dbg_reverse) z = z + 1 == (int)gridDim.z ? 0 : z + 1;  // (tests: the workers get the LARGEST ids)
        if (z == 0) {
            big_worker_wg(E, E.b_off + blockIdx.x, blockIdx.y, gridDim.y, it_arg);
            return;
        }
        --z;
    }
    shade_wg<EDGE>(E, E.b_off + blockIdx.x, blockIdx.y, gridDim.y, z, it_arg);
}

// group form: grid (sum of the members' hypotheses, largest slice count, 2)
template <bool EDGE>
__global__ __launch_bounds__(256, SHADE_MIN_WAVES) void shade_group_kernel(const EngineDev* __restrict__ tab, GroupHdr G, int it_arg)
{
    const int o = group_find(G, blockIdx.x);
    const EngineDev& E = tab[G.idx[o]];
    if ((int)blockIdx.y >= E.s_shade) return;
    int z = blockIdx.z;
    if (E.big_inline) {  // (the same for every member of a launch: grid z = 3)
        if (E.dbg_reverse) z = z + 1 == (int)gridDim.z ? 0 : z + 1;
        if (z == 0) {
            big_worker_wg(E, (int)blockIdx.x - G.bpre[o], blockIdx.y, E.s_shade, it_arg);
            return;
        }
        --z;
    }
    if (z >= E.n_roles) return;
    shade_wg<EDGE>(E, (int)blockIdx.x - G.bpre[o], blockIdx.y, E.s_shade, z, it_arg);
}

// ---------------------------------------------------------------------------------------------
// Edge term (extension, no reference counterpart; definition: oracle/ddx_oracle.c orc_loss_edge), after shade_kernel.
// Owner computes: the wave of an 8x8 quadrant owns the loss terms AND the gradient of its 64 pixels.  d loss / d lum of
// pixel p needs the Sobel coefficients of the 3x3 loss terms around p, which need the luminance of THEIR 3x3: a 12x12
// luminance halo (from lumbuf where zbuf says covered, 0 elsewhere), a 10x10 block of loss terms, 64 owned pixels.
// Everything comes from buffers the colour role wrote (lum, U = d lum / d final): one level of independent loads, no
// texture fetch, no barycentrics -- the old edge role re-shaded the 100 halo pixels of every quadrant in two rounds.
#define EH (QUAD + 4)  // 12: luminance halo
#define ET (QUAD + 2)  // 10: loss terms
__device__ __forceinline__ void edge_wg(const EngineDev& E, int b, int sl, int S, int it_arg)
{
    __shared__ float s_l[WAVES_PER_TILE][EH * EH];
    __shared__ float s_cx[WAVES_PER_TILE][ET * ET + 4], s_cy[WAVES_PER_TILE][ET * ET + 4];
    const ddx_engine_desc& d = E.d;
    const RasterScratch& L = E.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = d.H, W = d.W;
    const int n_tiles = L.b_count[b];  // (count and ordered list: written by shade_kernel's scan)
    const int par = (it_arg >= 0 ? it_arg : E.st->it_next - 1) & 1;
    const unsigned long long* __restrict__ zb = L.zbuf + ((size_t)par * d.B + b) * L.zper;
    const float* __restrict__ lumb = E.lumbuf + (size_t)b * H * W;
    const float lrb = E.b.lr_mult[b];
    const float kc = d.w_edge * lrb * __fdiv_rn(1.0f, (float)d.B_global) / (2.0f * (float)H * (float)W) * 0.125f;
    // accumulated over the workgroup's tiles, reduced once at the end (as shade_kernel): 12 d loss / d final + the loss.  Per-lane
    // accumulators in LDS (13 registers held across the tile loop cost two waves per SIMD): slot [i][tid], conflict-free
    __shared__ float s_acc[13][256];
#pragma unroll
    for (int i = 0; i < 13; ++i) s_acc[i][tid] = 0.f;
    for (int k = sl; k < n_tiles; k += S) {
        const int txy = L.active[(size_t)b * L.NT + k];
        const int qx = (txy & 0xffff) * DDX_TILE + (wave & 1) * QUAD, qy = (txy >> 16) * DDX_TILE + (wave >> 1) * QUAD;
        // ---- everything this quadrant needs is requested up front: the 12x12 (zbuf, lum) halo in 3 rounds of lanes, the
        // observed-image gradients of the 10x10 loss terms in 2, U of the owned pixel
        const int lx = lane % QUAD, ly = lane / QUAD;
        const int px = qx + lx, py = qy + ly;
        const bool inimg = px < W && py < H;
        const size_t pix = (size_t)py * W + px;
        bool cov[3];
        float lm[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = r * 64 + lane;
            const int gx = qx - 2 + e % EH, gy = qy - 2 + e / EH;
            cov[r] = false; lm[r] = 0.f;
            if (e < EH * EH && gx >= 0 && gy >= 0 && gx < W && gy < H) {
                const size_t g = (size_t)gy * W + gx;
                cov[r] = zb[zaddr(gx, gy, L.zwb)] != ~0ull;
                lm[r] = lumb[g];  // garbage where nothing is drawn: masked below
            }
        }
        float2 ge[2];
        bool term[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = r * 64 + lane;
            const int gx = qx - 1 + e % ET, gy = qy - 1 + e / ET;
            term[r] = e < ET * ET && gx >= 0 && gy >= 0 && gx < W && gy < H;
            ge[r] = term[r] ? E.gtedge[(size_t)gy * W + gx] : make_float2(0.f, 0.f);
        }
        float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0, u2 = u0;
        const bool own_cov = inimg && zb[zaddr(px, py, L.zwb)] != ~0ull;
        if (inimg) {  // (read unconditionally inside the image: masked by own_cov below -- keeps the loads independent of zbuf)
            const float4* ub = reinterpret_cast<const float4*>(E.ubuf + ((size_t)b * H * W + pix) * 12);
            u0 = ub[0]; u1 = ub[1]; u2 = ub[2];
        }
        bool any = false;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = r * 64 + lane;
            if (e < EH * EH) s_l[wave][e] = cov[r] ? lm[r] : 0.f;
            any |= cov[r];
        }
        if (__ballot(any) == 0ull) {  // nothing drawn in or around this quadrant: only background terms
            wave_lds_sync();  // (s_l is rewritten by the next tile)
            continue;
        }
        wave_lds_sync();
        // ---- the 10x10 loss terms: Sobel of the rendered luminance against the observed image's gradients
        float own_loss = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = r * 64 + lane;
            float cx = 0.f, cy = 0.f;
            if (term[r]) {
                const int tx = e % ET, ty = e / ET;           // term coordinates in the 10x10 block
                const float* c = &s_l[wave][(ty + 1) * EH + (tx + 1)];  // its centre in the 12x12 halo
                float v[3][3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) v[dy][dx] = c[(dy - 1) * EH + (dx - 1)];
                float gx, gy;
                sobel3(v, gx, gy);
                const float ex = gx - ge[r].x, ey = gy - ge[r].y;
                cx = kc * sgnf(ex);
                cy = kc * sgnf(ey);
                // the loss itself is counted by the owner of the pixel only (terms of the outer ring belong to neighbours)
                if (tx >= 1 && tx <= QUAD && ty >= 1 && ty <= QUAD) own_loss += (fabsf(ex) + fabsf(ey)) - (fabsf(ge[r].x) + fabsf(ge[r].y));
            }
            if (e < ET * ET) { s_cx[wave][e] = cx; s_cy[wave][e] = cy; }
        }
        wave_lds_sync();
        // ---- owned pixel: d loss / d lum from the 3x3 terms around it, times U
        if (own_cov) {
            float g = 0.f;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    // term n = p - (dy, dx): p is its (dy, dx) neighbour, coefficient kx[dy][dx] = dx (2 - |dy|), ky = dy (2 - |dx|)
                    const int n = (ly + 1 - dy) * ET + (lx + 1 - dx);
                    g = __fmaf_rn(s_cx[wave][n], (float)(dx * (2 - (dy < 0 ? -dy : dy))), g);
                    g = __fmaf_rn(s_cy[wave][n], (float)(dy * (2 - (dx < 0 ? -dx : dx))), g);
                }
            const float uu[12] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w, u2.x, u2.y, u2.z, u2.w};
#pragma unroll
            for (int i = 0; i < 12; ++i) s_acc[i][tid] += g * uu[i];  // (own slot: plain read-modify-write)
        }
        // the quadrant's loss terms: lanes hold up to 2 terms each (own_loss already restricted to owned pixels)
        if (own_loss != 0.f) s_acc[12][tid] += own_loss;
        wave_lds_sync();  // (the LDS arrays are reused by the next tile)
    }
    if (sl < n_tiles) {  // (workgroup-uniform)
        __shared__ float s_rows[WAVES_PER_TILE][NPART];
        float* part = E.partials + (((size_t)b * E.pslices + sl) * NROLE + 2) * NPART;
        float mine = 0.f, mine2 = 0.f;
        float acc[13];
        bool nz = false;
#pragma unroll
        for (int i = 0; i < 13; ++i) { acc[i] = s_acc[i][tid]; nz |= acc[i] != 0.f; }
        if (__ballot(nz) != 0ull) {
            wave_sum_n_lastrow(acc);  // totals in lanes 48..63
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (lane == 48 + i) mine = acc[i];
            if (lane == 32 + 19) mine2 = acc[12];
        }
        if (lane >= 48) {  // slots 0..15 from lanes 48..63, slots 16..23 from lanes 48..55
            s_rows[wave][lane - 48] = mine;
            if (lane < 56) s_rows[wave][lane - 32] = mine2;
        }
        __syncthreads();
        if (tid < NPART) part[tid] = (s_rows[0][tid] + s_rows[1][tid]) + (s_rows[2][tid] + s_rows[3][tid]);
    }
}

template <bool HALF = false>
__global__ __launch_bounds__(256) void edge_kernel(EngineDev E, int it_arg) { edge_wg(E, E.b_off + blockIdx.x, blockIdx.y, gridDim.y, it_arg); }

// group form: grid (sum of the members' hypotheses, largest slice count); members without the edge term leave at once
__global__ __launch_bounds__(256) void edge_group_kernel(const EngineDev* __restrict__ tab, GroupHdr G, int it_arg)
{
    const int o = group_find(G, blockIdx.x);
    const EngineDev& E = tab[G.idx[o]];
    if (!E.d.use_edge || (int)blockIdx.y >= E.s_edge) return;
    edge_wg(E, (int)blockIdx.x - G.bpre[o], blockIdx.y, E.s_edge, it_arg);
}

// ---------------------------------------------------------------------------------------------
// The optimiser step of iteration j for hypothesis b: the head of step_kernel and all of finish_kernel.
// EVERY workgroup of the hypothesis runs it, redundantly (one kernel less in the iteration's chain; the partial rows are a few
// KB from L2): fixed-order sum of the hypothesis' partial rows (one per shade / edge slice and role), whole-frame constants +
// the background depth term sum |seg| |d_bg - gt| of the whole frame from the depth-sorted seg list with prefix sums in
// double (two 64-way searches on one wave + six loads), proj^T chain, quaternion chain, SGD/Adam (lane-parallel tail), loss
// log.  Workgroup `slice` of `n_slices` also re-arms its share of what iteration j dirtied (zbuf of the active tiles and their
// flags, parity j & 1), and slice 0 writes parameters, optimiser state and logs.  Returns (after a barrier) with
// snew[0..6] = the updated parameters and sc[16..31] = proj in LDS.
// The sum is grouped the same way whatever the workgroup size (8 buckets of rows q = bucket mod 8, folded pairwise), so the
// 64- and 256-thread variants of step_kernel produce the same bits.
#define UPD_SLICES 8  // workgroups per hypothesis of finish_kernel; fewer for large batches (upd_slices())

template <int NTH, bool SEL = false /* finish_kernel: the selection of ddx_engine_run_select is compiled in */>
__device__ __forceinline__ void update_head(const EngineDev& E, int b, int j, int slice, int n_slices, float* snew, float* sc)
{
    constexpr int NG = NTH / 32, NW = NTH / 64, NBK = 8 / NG;  // groups of 32 threads, waves, row buckets per thread
    constexpr int SPEC = 64 / 8;                               // rows per bucket requested before the tile count is known (64 rows in all)
    const ddx_engine_desc& d = E.d;
    const int B = d.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float red[4][NPART];
    __shared__ float sums[NPART];
    __shared__ float sG[16];
    __shared__ float sgrad[8];
    __shared__ float s_bg[2];
    const int NT = E.L.NT;
    const int cur = j & 1;
    const int jj = tid % 32, grp = tid / 32;  // thread jj < NVALS of group g sums value jj of the rows of its buckets
#ifdef DDX_TRACE_HEAD
#define HSTAMP(i) do { if (E.trace && tid == 0 && NTH == 256 && b * n_slices + slice < TRACE_WG) E.trace[((size_t)2 * TRACE_WG + b * n_slices + slice) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HSTAMP(i)
#endif
    HSTAMP(0);
    const int n_act = *(E.L.b_count + b);
    const int* tiles = E.L.active + (size_t)b * NT;
    // ---- everything that does not depend on this iteration's sums is REQUESTED here, before the first wait: the partial
    // rows (speculative: rows of slices beyond the tile count are stale and masked below), the first tile of the re-arm
    // share, the totals of the sorted seg list.  The head is a chain of dependent round trips; these would otherwise each add one.
    const int rmask = E.role_mask;
    constexpr int NR = NROLE;
    const int PS = E.pslices, nrow = PS * NR;
    const float* pbase = E.partials + (size_t)b * nrow * NPART + jj;
    auto row_ok = [&](int q) {  // row q = slice * NR + role: written by shade_kernel (roles 0, 1) / edge_kernel (role 2) when the slice has a tile
        const int r = q % NR, s = q / NR;
        return q < nrow && ((rmask >> r) & 1) && s < (r == 2 ? E.s_edge : E.s_shade);
    };
    float v0[NBK][SPEC];
#pragma unroll
    for (int i = 0; i < NBK; ++i)
#pragma unroll
        for (int u = 0; u < SPEC; ++u) {
            const int q = (grp + NG * i) + 8 * u;
            v0[i][u] = (jj < NVALS && row_ok(q)) ? *(pbase + (size_t)q * NPART) : 0.f;
        }
    const int txy_first = slice < NT ? *(tiles + slice) : 0;
    const int ns = d.use_depth ? E.nseg : 0;  // (the size of the sorted seg list is known to the host since setup)
    const double segWn = E.seg_W[ns], segGn = E.seg_G[ns];
    // ---- the scalars of the tail, one per thread, into LDS: params 0..6, lr_mult 7, lr 8, proj 16..31, adam 32..45
    float sc_val = 0.f;
    if (tid < 7) sc_val = *(E.params2 + ((size_t)cur * 7 + tid) * B + b);
    else if (tid == 7) sc_val = E.b.lr_mult[b];
    else if (tid == 8) sc_val = E.b.lr_sched[j];
    else if (tid >= 16 && tid < 32) sc_val = E.b.proj[tid - 16];
    else if (tid >= 32 && tid < 46) sc_val = *(E.adam + ((size_t)cur * 14 + (tid - 32)) * B + b);
    const float dbg = -*(E.mats + ((size_t)cur * B + b) * 32 + 11);
    // ---- partial sums, fixed order => bit-reproducible
    float acc[NBK];
    {
        const int nlive = min(PS, n_act) * NR;  // (slice s has a tile <=> s < n_act)
#pragma unroll
        for (int i = 0; i < NBK; ++i) {
            acc[i] = 0.f;
            const int q0 = grp + NG * i;
#pragma unroll
            for (int u = 0; u < SPEC; ++u) acc[i] += (q0 + 8 * u < nlive) ? v0[i][u] : 0.f;
            if (jj < NVALS)
                for (int q1 = q0 + 8 * SPEC; q1 < nlive; q1 += 64) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {  // 8 loads in flight (a load-add chain would pay one L2 round trip per row)
                        const int q = q1 + 8 * u;
                        v[u] = (q < nlive && row_ok(q)) ? *(pbase + (size_t)q * NPART) : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[i] += v[u];
                }
        }
    }
    HSTAMP(1);
    if (tid < 46) sc[tid] = sc_val;
    // ---- re-arm what iteration j dirtied (zbuf of its active tiles, their flags; parity j & 1) so that no pass needs a
    // memset: tile k of the hypothesis' list is re-armed by workgroup k % n_slices.  Independent of the sums.
    {
        unsigned long long* Z = E.L.zbuf + ((size_t)cur * B + b) * E.L.zper;
        unsigned char* flag = E.L.tile_flag + ((size_t)cur * B + b) * E.L.NTp;
        unsigned char* big = E.L.tile_big + ((size_t)cur * B + b) * E.L.NTp;
        for (int k = slice; k < n_act; k += n_slices) {  // (workgroup-uniform)
            const int txy = k == slice ? txy_first : *(tiles + k);
            const int tx = txy & 0xffff, ty = txy >> 16;
            if (tid == 0) {
                flag[ty * E.L.ntx + tx] = 0;
                big[ty * E.L.ntx + tx] = 0;
            }
#pragma unroll
            for (int p = tid; p < DDX_TILE * DDX_TILE; p += NTH) {
                const int zx = tx * DDX_TILE + p % DDX_TILE, zy = ty * DDX_TILE + p / DDX_TILE;
                if (zx < d.W && zy < d.H) Z[zaddr(zx, zy, E.L.zwb)] = ~0ull;
            }
        }
    }
    // buckets 2w and 2w+1 are folded first (for 256 threads they live in the two halves of wave w), then the four pairs
#pragma unroll
    for (int i = 0; i < NBK; ++i) acc[i] += __shfl_xor(acc[i], 32, 64);
    if (lane < NPART) {
        if (NBK == 1) red[wave][lane] = acc[0];
        else {
#pragma unroll
            for (int i = 0; i < NBK; ++i) red[i % 4][lane] = acc[i];  // (NTH = 64: thread group g in {0,1} holds buckets g, g+2, g+4, g+6 = pairs 0..3)
        }
    }
    HSTAMP(2);
    __syncthreads();
    HSTAMP(3);
    if (tid < NPART) sums[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    // ---- whole-frame background depth term: sum_i w_i |dbg - g_i| and sum_i w_i sgn(dbg - g_i) (w = |seg0|, g = observed depth)
    // from the sorted list: k1 = #(g < dbg), k2 = #(g <= dbg) by a 64-way search per round on the last wave (no barrier), then
    // prefix-sum differences in double
    if (d.use_depth && wave == NW - 1) {
        int lo1 = 0, hi1 = ns, lo2 = 0, hi2 = ns, k1 = -1, k2 = -1;
        while (k1 < 0 || k2 < 0) {  // (wave-uniform)
            const int span1 = hi1 - lo1, span2 = hi2 - lo2;
            const int step1 = (span1 + 63) / 64, step2 = (span2 + 63) / 64;
            const int i1 = lo1 + lane * step1, i2 = lo2 + lane * step2;
            const bool in1 = k1 < 0 && span1 > 0 && i1 < hi1, in2 = k2 < 0 && span2 > 0 && i2 < hi2;
            const float g1 = in1 ? E.seg_gd[i1] : 0.f, g2 = in2 ? E.seg_gd[i2] : 0.f;
            const int c1 = __popcll(__ballot(in1 && g1 < dbg)), c2 = __popcll(__ballot(in2 && g2 <= dbg));
            if (k1 < 0) {
                if (span1 <= 0 || c1 == 0) k1 = lo1;
                else if (step1 == 1) k1 = lo1 + c1;
                else { const int pv = lo1 + (c1 - 1) * step1; lo1 = pv + 1; hi1 = min(pv + step1, hi1); }
            }
            if (k2 < 0) {
                if (span2 <= 0 || c2 == 0) k2 = lo2;
                else if (step2 == 1) k2 = lo2 + c2;
                else { const int pv = lo2 + (c2 - 1) * step2; lo2 = pv + 1; hi2 = min(pv + step2, hi2); }
            }
        }
        if (lane == 0) {
            const double W1 = E.seg_W[k1], G1 = E.seg_G[k1], W2 = E.seg_W[k2], G2 = E.seg_G[k2], dd = (double)dbg;
            s_bg[0] = (float)((dd * W1 - G1) + ((segGn - G2) - dd * (segWn - W2)));
            s_bg[1] = (float)(W1 - (segWn - W2));
        }
    }
    __syncthreads();
    HSTAMP(4);
    const float bgsum = d.use_depth ? s_bg[0] : 0.f, bgder = d.use_depth ? s_bg[1] : 0.f;
    // ---- tail on wave 0, one lane per output where the work allows
    const bool writer = slice == 0;
    if (wave == 0) {
        const float npx = (float)d.H * (float)d.W;
        const float lrb = sc[7];
        // loss log: weighted, not LR-scaled (diffdope.py:558-560,576-578,604-608)
        if (writer && lane < 4 && (E.eval_grad ? E.eval_loss != nullptr : E.b.loss_log != nullptr)) {
            float v = 0.f;
            if (lane == 0 && d.use_rgb) v = d.w_rgb * ((float)(E.st->c_rgb + (double)sums[16]) / (3.0f * npx));
            if (lane == 1 && d.use_depth) v = d.w_depth * ((float)((double)bgsum + (double)sums[17]) / npx);
            if (lane == 2 && d.use_mask) v = d.w_mask * ((float)(E.st->c_mask + (double)sums[18]) / (3.0f * npx));
            if (lane == 3 && d.use_edge) v = d.w_edge * ((float)(E.st->c_edge + (double)sums[19]) / (2.0f * npx));
            if (E.eval_grad) E.eval_loss[(size_t)lane * B + b] = v;
            else E.b.loss_log[((size_t)j * 4 + lane) * B + b] = v;
        }
        // ---- arg-min over the hypotheses inside this launch (ddx_engine_run_select; get_argmin / get_pose, diffdope.py:1488-1513,
        // 1618-1632): the mean of the used loss rows exactly as select_best_kernel forms it, ties to the lowest index
        if (SEL && writer && E.sel_out) {
            float v = 0.f;
            if (lane == 0 && d.use_rgb) v = d.w_rgb * ((float)(E.st->c_rgb + (double)sums[16]) / (3.0f * npx));
            if (lane == 1 && d.use_depth) v = d.w_depth * ((float)((double)bgsum + (double)sums[17]) / npx);
            if (lane == 2 && d.use_mask) v = d.w_mask * ((float)(E.st->c_mask + (double)sums[18]) / (3.0f * npx));
            if (lane == 3 && d.use_edge) v = d.w_edge * ((float)(E.st->c_edge + (double)sums[19]) / (2.0f * npx));
            const float v0 = __shfl(v, 0, 64), v1 = __shfl(v, 1, 64), v2 = __shfl(v, 2, 64), v3 = __shfl(v, 3, 64);
            if (lane == 0) {
                float a = 0.f;
                int n_used = 0;
                if (d.use_rgb) { a += v0; ++n_used; }
                if (d.use_depth) { a += v1; ++n_used; }
                if (d.use_mask) { a += v2; ++n_used; }
                if (d.use_edge) { a += v3; ++n_used; }
                a = __fdiv_rn(a, (float)max(n_used, 1));
                if (a == a) {  // (a NaN loss never wins: select_best_kernel's comparisons)
                    unsigned u = __float_as_uint(a);
                    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                    atomicMin(&E.st->sel_key, ((unsigned long long)u << 32) | (unsigned)b);
                }
                __builtin_amdgcn_s_waitcnt(0);  // the key has been folded in before this workgroup counts itself
                const int arrived = __hip_atomic_fetch_add(&E.st->sel_arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (arrived == B - 1) {  // the last hypothesis: every key is in
                    const unsigned long long key = __hip_atomic_load(&E.st->sel_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int w = 0;
                    float best = INFINITY;
                    if (key != ~0ull) {
                        unsigned ub = (unsigned)(key >> 32);
                        ub = (ub & 0x80000000u) ? (ub & 0x7fffffffu) : ~ub;
                        best = __uint_as_float(ub);
                        w = (int)(unsigned)(key & 0xffffffffull);
                    }
                    // (a run whose in-launch tile pass timed out is void: NaN tells the reader of the row to call ddx_engine_run_check)
                    const int fl = __hip_atomic_load(&E.st->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    E.sel_out[1] = (float)(w + E.sel_lo);
                    const float* Mw = E.mats + ((size_t)cur * B + w) * 32;  // mtx of iteration j (written by the step_kernel that drew it)
                    for (int i = 0; i < 16; ++i) E.sel_out[2 + i] = Mw[i];
                    // (the loss LAST and behind the rest of the row: a host that polls the row's first word in mapped pinned memory reads a complete row)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(E.sel_out, fl ? __uint_as_float(0x7fc00000u) : best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(&E.st->sel_key, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&E.st->sel_arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // d loss / d mtx = proj^T . dFinal (+ direct depth row): lane = k*4 + j
        if (lane < 16) {
            const int k = lane >> 2, c = lane & 3;
            float a = 0.f;
            a = __fmaf_rn(sc[16 + 0 * 4 + k], sums[0 + c], a);   // dFinal row x
            a = __fmaf_rn(sc[16 + 1 * 4 + k], sums[4 + c], a);   // row y
            a = __fmaf_rn(sc[16 + 3 * 4 + k], sums[8 + c], a);   // row w (row z carries no gradient)
            if (d.use_depth && k == 2) {
                a += sums[12 + c];
                if (c == 3) a += -(d.w_depth * lrb / ((float)d.B_global * npx)) * bgder;  // whole-frame background term (depth_bg = -m23)
            }
            sG[lane] = a;
        }
        wave_lds_sync();
        if (lane == 0) {
            // quaternion chain (the reverse of diffdope.py:57-80 and :1091)
            const float* G = sG;
            const float nq = sqrtf(sc[0] * sc[0] + sc[1] * sc[1] + sc[2] * sc[2] + sc[3] * sc[3]);
            const float x = sc[0] / nq, y = sc[1] / nq, z = sc[2] / nq, w = sc[3] / nq;
            const float gx = G[1] * 2 * y + G[2] * 2 * z + G[4] * 2 * y + G[5] * (-4 * x) + G[6] * (-2 * w) + G[8] * 2 * z + G[9] * 2 * w + G[10] * (-4 * x);
            const float gy = G[0] * (-4 * y) + G[1] * 2 * x + G[2] * 2 * w + G[4] * 2 * x + G[6] * 2 * z + G[8] * (-2 * w) + G[9] * 2 * z + G[10] * (-4 * y);
            const float gz = G[0] * (-4 * z) + G[1] * (-2 * w) + G[2] * 2 * x + G[4] * 2 * w + G[5] * (-4 * z) + G[6] * 2 * y + G[8] * 2 * x + G[9] * 2 * y;
            const float gw = G[1] * (-2 * z) + G[2] * 2 * y + G[4] * 2 * z + G[6] * (-2 * x) + G[8] * (-2 * y) + G[9] * 2 * x;
            const float dot = gx * x + gy * y + gz * z + gw * w;
            sgrad[0] = (gx - x * dot) / nq; sgrad[1] = (gy - y * dot) / nq; sgrad[2] = (gz - z * dot) / nq; sgrad[3] = (gw - w * dot) / nq;
            sgrad[4] = G[3]; sgrad[5] = G[7]; sgrad[6] = G[11];
        }
        wave_lds_sync();
        // optimiser step: lane = parameter
        if (lane < 7 && E.eval_grad) {  // evaluation pass: hand out the gradient, leave every state as it is
            if (writer) E.eval_grad[(size_t)lane * B + b] = sgrad[lane];
            snew[lane] = sc[lane];
        } else if (lane < 7) {
            const float g = sgrad[lane], lr = sc[8];
            float pnew;
            if (d.optimizer == 0) {
                pnew = sc[lane] - lr * g;
            } else {
                const float b1 = d.adam_beta1, b2 = d.adam_beta2;
                const float c1 = 1.f - exp2f((float)(j + 1) * log2f(b1)), c2 = 1.f - exp2f((float)(j + 1) * log2f(b2));
                const float m1 = b1 * sc[32 + lane] + (1.f - b1) * g;
                const float m2 = b2 * sc[39 + lane] + (1.f - b2) * g * g;
                if (writer) {
                    E.adam[((size_t)(1 - cur) * 14 + lane) * B + b] = m1;
                    E.adam[((size_t)(1 - cur) * 14 + 7 + lane) * B + b] = m2;
                }
                pnew = sc[lane] - lr * (m1 / c1) / (sqrtf(m2 / c2) + d.adam_eps);
            }
            snew[lane] = pnew;
            if (writer) {
                E.params2[((size_t)(1 - cur) * 7 + lane) * B + b] = pnew;
                E.b.params[(size_t)lane * B + b] = pnew;
            }
        }
        if (writer && b == 0) {
            // status of iteration j (single lanes across kernel boundaries, no atomics)
            int tot = 0, out = 0;
            for (int i = lane; i < B; i += 64) {
                tot += E.L.b_count[i];
                out += E.inside[i] == 0;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { tot += __shfl_xor(tot, o, 64); out += __shfl_xor(out, o, 64); }
            if (lane == 0) {
                E.st->last_active = tot;
                E.st->outside = out;
                E.st->last_pairs = E.L.counters[3 + cur];
            }
        }
    }
    HSTAMP(5);
    __syncthreads();
    HSTAMP(6);
#undef HSTAMP
}

// Meshlet geometry of a step_kernel instantiation: TPL triangles per thread, two vertex slots per thread.
// MODE: the fragment variant of scatter_resolve (raster_dev.h).
//
// The chain of one iteration used to be four launches (transform+update, scatter, compaction, shade) with the clip-space
// vertices and their window snap travelling through HBM in between; here the optimiser step, the transform and the scatter
// rasteriser are one workgroup-local pipeline and the compaction is gone (shade_kernel scans the flags itself).
template <int TPL, int NTH, int MODE, bool TAB = false /* balanced shares: the slot's meshlets come from the table, and the launch may record their times */>
__device__ __forceinline__ void step_wg(const EngineDev& E, int b, int slot, int SL, int mode, int it_arg)
{
    constexpr int NTRI = TPL * NTH, NVC = 2 * NTH;
    const ddx_engine_desc& d = E.d;
    // workgroup `slot` of the SL of hypothesis b: the head is paid once per workgroup, and SL is chosen so that all workgroups are
    // resident (launch_step)
    const int B = d.B, V = d.V, M = E.n_meshlets;
    const int tid = threadIdx.x, lane = tid & 63;
    __shared__ float4 s_clip[NVC];
    __shared__ int2 s_snap[NVC];
    __shared__ float snew[8];  // the parameters this iteration is drawn with
    __shared__ float sc[64];   // update_head's scalars; 16..31 = proj
    const int wg_id = b * SL + slot;
#if DDX_HEAD_PRIO
    __builtin_amdgcn_s_setprio(DDX_HEAD_PRIO);
#endif
    STAMP(E, 0, wg_id, 0);
    STAMP_HW(E, wg_id);
    // (it_arg >= 0: the host names the iteration -- plain stream launches -- and the kernel starts without a dependent scalar load;
    // -1: replayed from a captured graph, whose arguments are frozen: the device counter says which iteration this is)
    const int it = it_arg >= 0 ? it_arg : E.st->it;
    const int par = it & 1;
    // ---- the first meshlet (static tables, fixed-size slots: nothing here depends on another load): requested before the head
    float4 vr[2];
    int2 tr[TPL];
    // workgroup `slot` draws the meshlets [slot npw, (slot + 1) npw) of the INTERLEAVED order (engine_setup: front / back of the
    // Morton curve alternate, so a workgroup's meshlets lie on opposite sides of the object and one of each pair is culled whatever
    // the pose: the launch used to wait for the workgroups whose meshlets were all visible).  (Handing meshlets out through a
    // per-hypothesis ticket counter instead was built and measured: 64 hot words of returning device-scope atomics made every
    // workgroup slower -- head 5.1 -> 9.4 us, cfg2 44.4 -> 54 us per iteration.)
    const int npw = (M + SL - 1) / SL;
    const int m_begin = slot * npw, m_end = min(M, m_begin + npw);
    const bool tabled = TAB && E.slot_table != 0;
    const int n_my = tabled ? (int)E.slot_cnt[slot] : max(0, m_end - m_begin), t_off = tabled ? (int)E.slot_off[slot] : 0;
    // (the slot's list goes to LDS once: indexing the kernel arguments inside the meshlet loop would put a scalar load and its wait,
    // which also waits for the LDS traffic, into every trip)
    __shared__ unsigned short s_items[TAB ? 64 : 1];
    if (tabled && tid < min(n_my, 64)) s_items[tid] = E.slot_items[t_off + tid];  // (visible after the barrier of the head / first-iteration branch)
    auto meshlet_of = [&](int k) { return tabled ? (int)s_items[k] : m_begin + k; };
    const int m0 = n_my > 0 ? (tabled ? (int)E.slot_items[t_off] : m_begin) : M - 1;  // (a slot without meshlets only shares the head's re-arm work)
#pragma unroll
    for (int u = 0; u < 2; ++u) vr[u] = E.mvert[(size_t)m0 * NVC + u * NTH + tid];
#pragma unroll
    for (int k = 0; k < TPL; ++k) tr[k] = E.mtri[(size_t)m0 * NTRI + k * NTH + tid];
    if (mode == STEP_NORMAL) {
        update_head<NTH, false>(E, b, it - 1, slot, SL, snew, sc);
    } else {
        // first iteration of a run: the parameters as the caller holds them (un-normalised)
        if (tid < 7) snew[tid] = E.b.params[(size_t)tid * B + b];
        if (tid >= 16 && tid < 32) sc[tid] = E.b.proj[tid - 16];
        if (mode == STEP_FIRST && slot == 0 && tid < 21) {  // what ddx_engine_run_check restores should this run have to be repeated (EngineDev::run_snap)
            E.run_snap[(size_t)tid * B + b] = tid < 7 ? E.b.params[(size_t)tid * B + b] : E.adam[((size_t)par * 14 + (tid - 7)) * B + b];
        }
        __syncthreads();
    }
    const bool writer = slot == 0;
    STAMP(E, 0, wg_id, 1);
    // ---- pose -> matrices (every lane, from LDS broadcasts: ~150 flops, cheaper than a dependent load)
    float Fr[4];
    {
        float q[4], t3[3], M4[16], pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = snew[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t3[i] = snew[4 + i];
#pragma unroll
        for (int k = 0; k < 4; ++k) pr[k] = sc[16 + (lane & 3) * 4 + k];  // row lane % 4 of proj
        {   // q / |q| (diffdope.py:1091) and [R|t] (diffdope.py:46-89)
            const float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = __fdiv_rn(q[i], nq);
            quat_to_matrix(q, t3, M4);
        }
        final_row(pr, M4, Fr);  // row lane % 4 of final = proj . mtx (torch.matmul at :195, k-ordered fma)
        if (writer && tid < 4) {  // lanes 0..3 hold rows 0..3 of final; each writes its row of both matrices
            float* dst = E.mats + ((size_t)par * B + b) * 32;
            float* logm = E.b.mtx_log ? E.b.mtx_log + ((size_t)it * B + b) * 16 : nullptr;
            const float Mr[4] = {tid == 0 ? M4[0] : (tid == 1 ? M4[4] : (tid == 2 ? M4[8] : M4[12])), tid == 0 ? M4[1] : (tid == 1 ? M4[5] : (tid == 2 ? M4[9] : M4[13])),
                                 tid == 0 ? M4[2] : (tid == 1 ? M4[6] : (tid == 2 ? M4[10] : M4[14])), tid == 0 ? M4[3] : (tid == 1 ? M4[7] : (tid == 2 ? M4[11] : M4[15]))};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dst[tid * 4 + c] = Mr[c];
                dst[16 + tid * 4 + c] = Fr[c];
                if (logm) logm[tid * 4 + c] = Mr[c];
            }
        }
    }
    // ---- is the whole object inside the view volume?  The 8 corners of its object-space bounding box through the same
    // transform (lanes 0..7 of every wave): w > 0, -w <= z <= w at every corner => at every vertex (see EngineDev::cull_sign)
    bool inside_all;
    {
        const int c = lane & 7;
        const f32x4 acc = xfm_vertex_mfma(Fr, (c & 1) ? E.bbox[3] : E.bbox[0], (c & 2) ? E.bbox[4] : E.bbox[1], (c & 4) ? E.bbox[5] : E.bbox[2]);
        const bool ok = acc.w > 0.f && acc.z >= -acc.w && acc.z <= acc.w;  // (NaN: false)
        inside_all = __ballot(lane < 8 && !ok) == 0ull;
    }
    const int cull = (E.cull_sign != 0 && inside_all) ? E.cull_sign : 0;
    if (writer) {
        if (mode != STEP_NORMAL && tid < 7) E.params2[((size_t)par * 7 + tid) * B + b] = snew[tid];
        if (tid == 0) {
            E.inside[b] = inside_all ? 1 : 0;
            E.L.bigcount[(size_t)(1 - par) * B + b] = 0;  // the other parity's list of large triangles: consumed, nobody reads it now
            E.L.bigarrive[(size_t)(1 - par) * B + b] = 0;
            if (b == 0) {
                E.L.counters[3 + (1 - par)] = 0;  // ... and its "a large triangle exists" word
                E.st->it_next = it + 1;           // (read by big_pass / shade / edge of this iteration)
                if (mode != STEP_NORMAL) E.st->it = it;
            }
        }
    }
    ScatterTarget tg;
    tg.P = reinterpret_cast<const float*>(s_clip);
    tg.Z = E.L.zbuf + ((size_t)par * B + b) * E.L.zper;
    tg.flag = E.L.tile_flag + ((size_t)par * B + b) * E.L.NTp;
    tg.big = E.L.tile_big + ((size_t)par * B + b) * E.L.NTp;
    tg.anybig = E.L.counters + 3 + par;
    tg.biglist = E.L.biglist + (size_t)b * d.T;
    tg.bigcount = E.L.bigcount + (size_t)par * B + b;
    tg.ntx = E.L.ntx; tg.nty = E.L.nty; tg.NT = E.L.NT; tg.zwb = E.L.zwb;
    tg.ndc = E.L.ndc;
    STAMP(E, 0, wg_id, 2);
#if DDX_HEAD_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    const bool rec = TAB && E.mcost_rec != 0;
    unsigned long long t_m = rec ? __builtin_amdgcn_s_memrealtime() : 0ull;
    for (int mk = 0; mk < (DDX_ABLATE >= 4 ? 0 : n_my); ++mk) {  // (workgroup-uniform)
        const int m = meshlet_of(mk);
        // ---- the meshlet's vertices on the matrix core -> LDS (clip + 1/256-pixel window snap); the owner of a vertex also
        // stores it for the antialias pass and the tile pass
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned bits = __float_as_uint(vr[u].w);
            const f32x4 acc = xfm_vertex_mfma(Fr, vr[u].x, vr[u].y, vr[u].z);
            const float4 c4 = make_float4(acc.x, acc.y, acc.z, acc.w);
            const int2 sn = snap_vertex(c4, d.H, d.W);
            s_clip[u * NTH + tid] = c4;
            s_snap[u * NTH + tid] = sn;
            if (bits != 0xffffffffu && (bits >> 31)) {
                const size_t g = (size_t)b * V + (bits & 0x7fffffffu);
                *reinterpret_cast<f32x4*>(E.clip + g * 4) = acc;
                E.L.snap[g] = sn;
            }
        }
        __syncthreads();
        // ---- the meshlet's triangles: vertices from LDS
        int t[TPL], i0[TPL], i1[TPL], i2[TPL];
        bool ok[TPL];
        int2 va[TPL], vb[TPL], vc[TPL];
#pragma unroll
        for (int k = 0; k < TPL; ++k) {
            ok[k] = tr[k].y >= 0;
            t[k] = ok[k] ? tr[k].y : d.T;
            i0[k] = tr[k].x & 1023; i1[k] = (tr[k].x >> 10) & 1023; i2[k] = (tr[k].x >> 20) & 1023;
            va[k] = s_snap[i0[k]]; vb[k] = s_snap[i1[k]]; vc[k] = s_snap[i2[k]];
        }
        // the next meshlet of this workgroup is requested now and lands while this one is rasterised
        if (mk + 1 < n_my) {
            const int mn = meshlet_of(mk + 1);
#pragma unroll
            for (int u = 0; u < 2; ++u) vr[u] = E.mvert[(size_t)mn * NVC + u * NTH + tid];
#pragma unroll
            for (int q = 0; q < TPL; ++q) tr[q] = E.mtri[(size_t)mn * NTRI + q * NTH + tid];
        }
#ifdef DDX_TRACE_HEAD
        if (mk == 0) STAMP(E, 0, wg_id, 3);
#else
        STAMP(E, 0, wg_id, mk == 0 ? 3 : 5);
#endif
#if DDX_ABLATE >= 3
        if ((va[0].x ^ vb[0].y ^ vc[TPL - 1].x ^ t[0] ^ i0[0]) == 0x12345677) tg.flag[0] = 1;  // (measurement build: the vertices are still transformed, stored and read back)
#else
        scatter_resolve<TPL, NTH, MODE>(tg, d.H, d.W, d.T, t, i0, i1, i2, ok, va, vb, vc, cull);
#endif
#ifdef DDX_TRACE_HEAD
        if (mk == 0) STAMP(E, 0, wg_id, 4);
#else
        STAMP(E, 0, wg_id, mk == 0 ? 4 : 6);
#endif
        __syncthreads();  // (s_clip / s_snap are rewritten by the next meshlet)
        if (rec) {  // (set-up's calibration launch only)
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (tid == 0) E.mcost[(size_t)b * M + m] = (unsigned short)min(65535ull, now - t_m);
            t_m = now;
        }
    }
    STAMP(E, 0, wg_id, 7);
}

#ifndef STEP_MIN_WAVES
#define STEP_MIN_WAVES 5  // waves per SIMD step_kernel is compiled for (<= 96 registers): 1280 resident 256-thread workgroups -- exactly the 20 slots x 64
                          // hypotheses of the headline launch.  (Rounds 2-3 compiled for 6 = 80 registers, which the plain variant met with 48 bytes of
                          // scratch per lane and the others not at all; at 5 nothing spills: cfg2 41.3 -> 40.8 us, cfg4 50.7 -> 48.9, cfg5 85.0 -> 82.3,
                          // everything else within +-1 %, 512-hypothesis batches unchanged)
#endif
// (the hybrid scatter variant -- close-ups -- needs 117 registers: 4 waves per SIMD, as it always ran)
#define STEP_WAVES(MODE) ((MODE) == 2 ? 4 : STEP_MIN_WAVES)
// grid (slots, B), or (B, slots) with E.step_xcd
template <int TPL, int NTH, int MODE, bool TAB = false, bool HALF = false>
__global__ __launch_bounds__(NTH, STEP_WAVES(MODE)) void step_kernel(EngineDev E, int mode, int it_arg)
{
    // (TAB: always the slot-major grid (B, slots))
    if (TAB) step_wg<TPL, NTH, MODE, true>(E, E.b_off + blockIdx.x, blockIdx.y, gridDim.y, mode, it_arg);
    else step_wg<TPL, NTH, MODE, false>(E, E.b_off + (E.step_xcd ? blockIdx.x : blockIdx.y), E.step_xcd ? blockIdx.y : blockIdx.x, E.step_xcd ? gridDim.y : gridDim.x, mode, it_arg);
}

// group form: grid (largest slot count, sum of the members' hypotheses)
template <int TPL, int NTH, int MODE>
__global__ __launch_bounds__(NTH, STEP_WAVES(MODE)) void step_group_kernel(const EngineDev* __restrict__ tab, GroupHdr G, int mode, int it_arg)
{
    const int o = group_find(G, blockIdx.y);
    if ((int)blockIdx.x >= G.sl[o]) return;
    step_wg<TPL, NTH, MODE>(tab[G.idx[o]], (int)blockIdx.y - G.bpre[o], blockIdx.x, G.sl[o], mode, it_arg);
}

// the optimiser step of the LAST iteration of a run (or of an evaluation pass): update_head alone, then both parities are clean
__device__ __forceinline__ void finish_wg(const EngineDev& E, int b, int slice, int n_slices, int it_arg)
{
    __shared__ float snew[8];
    __shared__ float sc[64];
    const int it = it_arg >= 0 ? it_arg : E.st->it, par = it & 1;
    update_head<256, true>(E, b, it - 1, slice, n_slices, snew, sc);
    if (slice == 0 && threadIdx.x == 0) {
        E.L.bigcount[(size_t)(1 - par) * E.d.B + b] = 0;
        E.L.bigarrive[(size_t)(1 - par) * E.d.B + b] = 0;
        if (b == 0) E.L.counters[3 + (1 - par)] = 0;
    }
}

__global__ __launch_bounds__(256) void finish_kernel(EngineDev E, int it_arg) { finish_wg(E, blockIdx.y, blockIdx.x, gridDim.x, it_arg); }

// group form: grid (slices, sum of the members' hypotheses)
__global__ __launch_bounds__(256) void finish_group_kernel(const EngineDev* __restrict__ tab, GroupHdr G, int it_arg)
{
    const int o = group_find(G, blockIdx.y);
    finish_wg(tab[G.idx[o]], (int)blockIdx.y - G.bpre[o], blockIdx.x, gridDim.x, it_arg);
}

// the tile pass for large / near-clipped triangles of the iteration being drawn (raster_dev.h big_pass_body); exits on one
// scalar load when the batch has none (always, for the 20k-50k-triangle meshes of the benchmark)
#define BIG_WAVES 16
#define BIG_GRID 256
__device__ __forceinline__ void big_wg(const EngineDev& E, int g, int Gn, int it_arg)
{
    const int par = (it_arg >= 0 ? it_arg : E.st->it_next - 1) & 1, B = E.d.B;
    if (E.L.counters[3 + par] == 0) return;
    unsigned long long n_done = 0;
    big_pass_body<BIG_WAVES>(E.clip, E.stri, E.L.snap, E.L.tile_big + (size_t)par * B * E.L.NTp, E.L.biglist, E.L.bigcount + (size_t)par * B,
                  E.L.zbuf + (size_t)par * B * E.L.zper, E.L.zper, E.L.zwb, E.L.ntx, E.L.NT, E.L.NTp, B, E.d.V, E.d.T, E.d.H, E.d.W, g, Gn, n_done);
}

__global__ __launch_bounds__(BIG_WAVES * 64) void big_pass_kernel(EngineDev E, int it_arg) { big_wg(E, blockIdx.x, gridDim.x, it_arg); }

// group form: grid (workgroups per member, members)
__global__ __launch_bounds__(BIG_WAVES * 64) void big_pass_group_kernel(const EngineDev* __restrict__ tab, GroupHdr G, int it_arg)
{
    big_wg(tab[G.idx[blockIdx.y]], blockIdx.x, gridDim.x, it_arg);
}

// ---------------------------------------------------------------------------------------------
enum { K_STEP, K_BIG, K_SHADE, K_EDGE, K_FINISH, K_COUNT };
static const char* const kKernelNames[K_COUNT] = {"step_kernel", "big_pass_kernel", "shade_kernel", "edge_kernel", "finish_kernel"};

// shading grid (B, S): S slices per hypothesis, SHADE_GRID workgroups per role in total (all resident at 4 waves/SIMD)
static dim3 shade_grid(const ddx_engine_desc& d)
{
    int grid = SHADE_GRID;
    if (const char* ov = getenv("DDX_SHADE_GRID")) grid = atoi(ov);  // (tuning)
    // SHADE_GRID workgroups per role when both roles run (2 x 512 = the 1024 the chip holds at 4 waves per SIMD); a launch with ONE
    // role -- no mask term, or the mask term alone -- gets all of them (round 4: cfg3's shading launch ran at 2 waves per SIMD)
    const int n_roles = ((d.use_rgb || d.use_depth || d.use_edge) ? 1 : 0) + (d.use_mask ? 1 : 0);
    if (n_roles <= 1) grid *= 2;
    int S = d.shade_slices > 0 ? d.shade_slices : grid / d.B;
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    return dim3(d.B, S);
}

// finish_kernel: the slices of a hypothesis only share its re-arm work
static int upd_slices(const ddx_engine_desc& d)
{
    int sl = UPD_SLICES;
    while (sl > 1 && (long long)d.B * sl > 1024) sl >>= 1;
    return sl;
}

// edge_kernel: 66 VGPRs, 7 waves/SIMD, so more resident workgroups than the shade kernel has
static dim3 edge_grid(const ddx_engine_desc& d)
{
    int S = d.edge_slices > 0 ? d.edge_slices : EDGE_GRID / d.B;
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    return dim3(d.B, S);
}

// a run that does not continue where the last one stopped (rewind): the optimiser state follows the iteration parity
static int run_prologue(ddx_engine* e, int it0, hipStream_t s)
{
    EngineDev& E = e->dev;
    if ((it0 & 1) != e->adam_parity) {
        const size_t half = (size_t)14 * E.d.B;
        DDX_HIP(hipMemcpyAsync(E.adam + (size_t)(it0 & 1) * half, E.adam + (size_t)e->adam_parity * half, half * sizeof(float),
                               hipMemcpyDeviceToDevice, s));
        e->adam_parity = it0 & 1;
    }
    return 0;
}

// step_kernel: mode STEP_FIRST draws iteration `it` from the caller's parameters; STEP_NORMAL steps the optimiser for iteration
// it - 1 and draws iteration it (it = -1: the device counter names it -- launches captured into a graph).
#define STEP_DISPATCH(CALL)                                            \
    do {                                                               \
        if (e->small_mesh) { CALL(1, 64, 1); }                         \
        else if (e->dev.scatter_mode == 3) { CALL(2, 256, 3); }        \
        else if (e->dev.scatter_mode == 2) { CALL(2, 256, 2); }        \
        else { CALL(2, 256, 0); }                                      \
    } while (0)

// workgroups of the engine's step_kernel variant the chip holds at once (registers, LDS and the SGPR rule included: the
// runtime's own occupancy answer)
static int step_capacity(ddx_engine* e)
{
    int per_cu = 0, cus = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
#define STEP_OCC(TPL, NTH, MODE)                                                                                                   \
    do {                                                                                                                           \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, step_kernel<TPL, NTH, MODE>, NTH, 0);                           \
    } while (0)
    STEP_DISPATCH(STEP_OCC);
#undef STEP_OCC
    if (per_cu < 1) per_cu = 4;
    return per_cu * cus;
}

// BALANCED SHARES.  The launch ends with its slowest workgroup, and per-workgroup stamps show two things a static equal split
// cannot know: a meshlet costs 0.5 us when the pose culls it and up to 6 us when it faces the camera, and the 4th / 5th workgroup
// dispatched to a CU run at about 0.8x / 0.5x the speed of the first three, from their first instruction on (it follows the
// dispatch order: with the slot-major grid, slot s holds the block ids [s B, (s + 1) B), i.e. the (s B / CUs)-th workgroup of
// each CU).  So the first step launch after a set-up records what each meshlet cost each hypothesis (mcost: s_memrealtime
// ticks of the workgroup that drew it), the host reads them back once -- the set-up has synchronised the stream anyway --,
// averages over the hypotheses, removes the speed of the slot that measured, and hands the meshlets to the slots longest first,
// each to the slot that would finish it earliest given its speed and its later start (LPT).  The table travels in the kernel
// arguments.  Which workgroup draws a meshlet does not change a bit of the result (atomicMin, owner stores, idempotent flags).
static double slot_speed(int slot, int B, int cus, double* head_ticks)
{
    const int rank = (int)(((long long)slot * B) / cus);
    static const double v3 = getenv("DDX_BAL_V3") ? atof(getenv("DDX_BAL_V3")) : 0.8, v4 = getenv("DDX_BAL_V4") ? atof(getenv("DDX_BAL_V4")) : 0.5,
                        v5 = getenv("DDX_BAL_V5") ? atof(getenv("DDX_BAL_V5")) : 0.4;  // (tuning)
    *head_ticks = rank <= 2 ? 0.0 : (rank == 3 ? 110.0 : 300.0);
    return rank <= 2 ? 1.0 : (rank == 3 ? v3 : (rank == 4 ? v4 : v5));
}

static int balance_slots(ddx_engine* e, int SL, hipStream_t s)
{
    EngineDev& E = e->dev;
    const int M = E.n_meshlets, B = E.d.B;
    std::vector<unsigned short> h((size_t)B * M);
    DDX_HIP(hipMemcpyAsync(h.data(), E.mcost, h.size() * sizeof(unsigned short), hipMemcpyDeviceToHost, s));
    DDX_HIP(hipStreamSynchronize(s));
    int cus = 256, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    const int npw = (M + SL - 1) / SL;
    std::vector<double> cost((size_t)M, 0.0);
    for (int m = 0; m < M; ++m) {
        double sum = 0.0;
        for (int b = 0; b < B; ++b) sum += (double)h[(size_t)b * M + m];
        double ht;
        cost[(size_t)m] = std::max(1.0, sum / B * slot_speed(m / npw, B, cus, &ht));  // (measured by slot m / npw of the equal split)
    }
    std::vector<int> order((size_t)M);
    for (int m = 0; m < M; ++m) order[(size_t)m] = m;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return cost[(size_t)a] > cost[(size_t)b2]; });
    std::vector<double> load((size_t)SL), speed((size_t)SL);
    std::vector<std::vector<int>> items((size_t)SL);
    for (int sl = 0; sl < SL; ++sl) { double ht; speed[(size_t)sl] = slot_speed(sl, B, cus, &ht); load[(size_t)sl] = ht; }
    for (int m : order) {
        int best = 0;
        double fbest = 1e300;
        for (int sl = 0; sl < SL; ++sl) {
            const double f = load[(size_t)sl] + cost[(size_t)m] / speed[(size_t)sl];
            if (f < fbest) { fbest = f; best = sl; }
        }
        load[(size_t)best] = fbest;
        items[(size_t)best].push_back(m);
    }
    int off = 0;
    for (int sl = 0; sl < SL; ++sl) {
        if (items[(size_t)sl].size() > 64) {  // (the kernel keeps a slot's list in 64 LDS entries: keep the equal split, and do not try again)
            E.slot_table = 0;
            e->balanced = true;
            return 0;
        }
        E.slot_off[sl] = (unsigned short)off;
        E.slot_cnt[sl] = (unsigned char)items[(size_t)sl].size();
        for (int m : items[(size_t)sl]) E.slot_items[off++] = (unsigned short)m;
    }
    E.slot_table = 1;
    e->balanced = true;
    if (getenv("DDX_DEBUG_BALANCE")) {
        fprintf(stderr, "ddx balance: M %d SL %d |", M, SL);
        for (int sl = 0; sl < SL; ++sl) fprintf(stderr, " %d:%d(%.0f)", sl, (int)E.slot_cnt[sl], load[(size_t)sl]);
        fprintf(stderr, "\n");
    }
    return 0;
}

// slots (workgroups) per hypothesis of step_kernel, and whether the slot table (balance_slots) applies to this engine
static int step_slots(ddx_engine* e, bool* can_balance)
{
    EngineDev& E = e->dev;
    // every workgroup the same number of meshlets, all of them resident -- a second round of workgroups would pay the head's chain
    // again -- and at least a few, because the slots also share the re-arm of the tiles the previous iteration dirtied (a slot
    // beyond the meshlet count does just that)
    if (e->step_resident <= 0) e->step_resident = step_capacity(e);
    int SL = std::max(1, std::min(E.n_meshlets, e->step_resident / E.d.B));
    SL = ddx_cdiv(E.n_meshlets, ddx_cdiv(E.n_meshlets, SL));  // (npw = ceil(M / SL) meshlets each; step_kernel computes the same npw)
    SL = std::max(SL, std::min(UPD_SLICES, std::max(1, e->step_resident / E.d.B)));
    // balanced shares (slot table): the first step launch after a set-up runs with equal shares and records what every meshlet cost
    // (worth it from a few meshlets per slot on -- the 51 200-triangle meshes: cfg3 +5 %, cfg50k64 +7 %; with two meshlets per slot
    // the longest-first hand-out has nothing to even out and the measured times are not additive enough: cfg2 -2 %, midpoly -7 %)
    *can_balance = e->balance && SL > 1 && SL <= 64 && E.n_meshlets <= 192 && E.n_meshlets >= e->balance_min_per_slot * SL;
    return SL;
}

// half: -1 = all hypotheses; 0 / 1 = the first / second half of them (two-stream runs: same slots, same bits, half the grid)
static int launch_step(ddx_engine* e, int mode, int it, hipStream_t s, int half = -1)
{
    RoctxRange rr("ddx.step_kernel");
    EngineDev& E = e->dev;
    const int nb = half < 0 ? E.d.B : E.d.B / 2;
    E.b_off = half > 0 ? nb : 0;
    bool can_balance = false;
    const int SL = step_slots(e, &can_balance);
    // (the table variant of the kernel is a separate instantiation with the slot-major grid -- a slot's place in the dispatch order
    // is then the same for every hypothesis --; the equal-share variant stays the code it was: with the table logic compiled into it
    // it ran 1.3-3 us slower on every workload that does not use it)
    bool capturing = false;
    {
        hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
        capturing = hipStreamIsCapturing(s, &cst) == hipSuccess && cst != hipStreamCaptureStatusNone;
    }
    // (the calibration reads its measurements back and synchronises the stream once per set-up -- ddx.h; never inside a capture)
    const bool calibrate = can_balance && !e->balanced && mode != STEP_NORMAL && !capturing;
    if (calibrate) { E.slot_table = 0; E.mcost_rec = 1; }
    const bool tab = can_balance && (calibrate || E.slot_table);
    const dim3 g = (tab || E.step_xcd) ? dim3(nb, SL) : dim3(SL, nb);
#define STEP_LAUNCH(TPL, NTH, MODE)                                                          \
    do {                                                                                     \
        if (half >= 0) {                                                                     \
            if (tab) step_kernel<TPL, NTH, MODE, true, true><<<g, NTH, 0, s>>>(E, mode, it); \
            else step_kernel<TPL, NTH, MODE, false, true><<<g, NTH, 0, s>>>(E, mode, it);    \
        } else if (tab) step_kernel<TPL, NTH, MODE, true><<<g, NTH, 0, s>>>(E, mode, it);    \
        else step_kernel<TPL, NTH, MODE, false><<<g, NTH, 0, s>>>(E, mode, it);              \
    } while (0)
    STEP_DISPATCH(STEP_LAUNCH);
#undef STEP_LAUNCH
    E.b_off = 0;
    DDX_LAUNCH_CHECK();
    if (calibrate) {
        E.mcost_rec = 0;
        if (int err = balance_slots(e, SL, s)) return err;
    }
    return 0;
}

// the rest of an iteration after its step_kernel: tile pass for large triangles, shading (+ edge term)
static int launch_rest(ddx_engine* e, int it, hipStream_t s, hipEvent_t* ev /* K_COUNT+1 events or null */, int half = -1)
{
    EngineDev& E = e->dev;
    const ddx_engine_desc& d = E.d;
    const int nb = half < 0 ? d.B : d.B / 2;
    E.b_off = half > 0 ? nb : 0;
    if (ev) DDX_HIP(hipEventRecord(ev[K_BIG], s));
    if (!E.big_inline) {
        RoctxRange rr("ddx.big_pass_kernel");
        big_pass_kernel<<<BIG_GRID, BIG_WAVES * 64, 0, s>>>(E, it);
    }
    if (ev) DDX_HIP(hipEventRecord(ev[K_SHADE], s));
    {
        RoctxRange rr("ddx.shade_kernel");
        const dim3 g(nb, E.s_shade, E.n_roles + (E.big_inline ? 1 : 0));  // (the slices fixed at creation: the partial rows are laid out for them)
        if (half >= 0) {
            if (d.use_edge) shade_kernel<true, true><<<g, 256, 0, s>>>(E, it);
            else shade_kernel<false, true><<<g, 256, 0, s>>>(E, it);
        } else if (d.use_edge) shade_kernel<true><<<g, 256, 0, s>>>(E, it);
        else shade_kernel<false><<<g, 256, 0, s>>>(E, it);
    }
    if (ev) DDX_HIP(hipEventRecord(ev[K_EDGE], s));
    if (d.use_edge) {
        RoctxRange rr("ddx.edge_kernel");
        if (half >= 0) edge_kernel<true><<<dim3(nb, E.s_edge), 256, 0, s>>>(E, it);
        else edge_kernel<false><<<dim3(nb, E.s_edge), 256, 0, s>>>(E, it);
    }
    E.b_off = 0;
    if (ev) DDX_HIP(hipEventRecord(ev[K_FINISH], s));
    DDX_LAUNCH_CHECK();
    return 0;
}

static int launch_finish(ddx_engine* e, int it /* the iteration after the last one drawn */, hipStream_t s)
{
    RoctxRange rr("ddx.finish_kernel");
    EngineDev& E = e->dev;
    const dim3 g(upd_slices(E.d), E.d.B);
    finish_kernel<<<g, 256, 0, s>>>(E, it);
    DDX_LAUNCH_CHECK();
    return 0;
}

static int run_iteration(ddx_engine* e, int it, hipStream_t s)  // one iteration after the first of a run (it = -1: for a graph)
{
    RoctxRange rr("ddx.iteration");
    if (int err = launch_step(e, STEP_NORMAL, it, s)) return err;
    return launch_rest(e, it, s, nullptr);
}

static int check_desc(const ddx_engine_desc* d)
{
    DDX_REQUIRE(d, DDX_E_NULL, "engine: NULL desc");
    DDX_REQUIRE(d->B >= 1 && d->B <= WORK_MAX_B && d->B_global >= d->B && d->V >= 3 && d->T >= 1 && d->H >= 1 && d->W >= 1 &&
                    d->H <= 4096 && d->W <= 4096 && d->max_iters >= 1,
                DDX_E_SHAPE, "engine: bad shape B=%d Bg=%d V=%d T=%d H=%d W=%d iters=%d", d->B, d->B_global, d->V, d->T, d->H, d->W, d->max_iters);
    DDX_REQUIRE((d->Th > 0) == (d->Tw > 0), DDX_E_SHAPE, "engine: Th/Tw must both be zero or positive");
    return 0;
}

extern "C" size_t ddx_engine_scratch_bytes(const ddx_engine_desc* desc)
{
    if (check_desc(desc)) return 0;
    EngineDev E;
    return engine_layout(E, *desc, nullptr);
}

extern "C" int ddx_engine_create(const ddx_engine_desc* desc, const ddx_engine_buffers* bufs, ddx_engine** out)
{
    if (int e = check_desc(desc)) return e;
    DDX_REQUIRE(bufs && out, DDX_E_NULL, "engine_create: NULL pointer");
    const ddx_engine_buffers& b = *bufs;
    DDX_REQUIRE(b.pos && b.tri && b.opp && b.proj && b.gt_seg && b.lr_mult && b.lr_sched && b.params && b.scratch, DDX_E_NULL,
                "engine_create: a required buffer is NULL");
    DDX_REQUIRE(!desc->use_rgb || (b.gt_rgb && ((desc->Th > 0) ? (b.uv && b.tex) : (b.vtx_color != nullptr))), DDX_E_NULL,
                "engine_create: rgb loss needs gt_rgb and (uv+tex | vtx_color)");
    DDX_REQUIRE(!desc->use_depth || b.gt_depth, DDX_E_NULL, "engine_create: depth loss needs gt_depth");
    DDX_REQUIRE(!desc->use_edge || (b.gt_rgb && ((desc->Th > 0) ? (b.uv && b.tex) : (b.vtx_color != nullptr))), DDX_E_NULL,
                "engine_create: edge loss needs gt_rgb and (uv+tex | vtx_color)");
    DDX_REQUIRE(((uintptr_t)b.scratch & 255) == 0, DDX_E_ALIGN, "engine_create: scratch must be 256-byte aligned");
    ddx_engine* e = new (std::nothrow) ddx_engine();
    DDX_REQUIRE(e, DDX_E_NULL, "engine_create: out of host memory");
    e->dev.d = *desc;
    e->dev.b = *bufs;
    e->dev.eval_grad = nullptr;
    e->dev.eval_loss = nullptr;
    e->dev.sel_out = nullptr;
    e->dev.sel_lo = 0;
    {
        EngineDev& E = e->dev;
        // shade roles: 0 colour/depth (also runs for the edge term: it produces the luminance + unit gradients that
        // edge_kernel consumes), 1 mask.  Partial slot 2 belongs to edge_kernel.
        const bool on[MAX_ROLES] = {desc->use_rgb || desc->use_depth || desc->use_edge, desc->use_mask != 0, false};
        E.n_roles = 0;
        E.role_mask = 0;
        for (int r = 0; r < MAX_ROLES; ++r) E.roles[r] = 0;
        // grid z order: the MASK role first.  Workgroups are dispatched in linear-id order, so the role at z = 0 takes the cold
        // misses on zbuf / observed images; the colour role is the long one (7 us per tile against 4) and now starts on lines
        // the mask role's halo reads already pulled into the XCD's L2 (cfg2 shade 22.0 -> 20.3 us)
        for (int r = MAX_ROLES - 1; r >= 0; --r)
            if (on[r]) {
                E.role_mask |= 1 << r;
                E.roles[E.n_roles++] = r;
            }
        if (E.n_roles == 0) {  // no term at all: run the mask role with weight 0 semantics (nothing to optimise)
            E.n_roles = 1;
            E.roles[0] = 1;
            E.role_mask = 2;
        }
        if (desc->use_edge) E.role_mask |= 4;  // slot 2
        E.st_role = E.roles[0];
        E.s_shade = (int)shade_grid(*desc).y;
        E.s_edge = desc->use_edge ? (int)edge_grid(*desc).y : 0;
        E.pslices = E.s_shade > E.s_edge ? E.s_shade : E.s_edge;
        E.cull_sign = 0;
        E.trace = nullptr;
        if (const char* ov = getenv("DDX_TRACE"))
            if (atoi(ov)) {
                DDX_HIP(hipMalloc(&E.trace, (size_t)3 * TRACE_WG * 8 * 8));
                DDX_HIP(hipMemset(E.trace, 0, (size_t)3 * TRACE_WG * 8 * 8));
            }
        E.step_xcd = 0;  // (launch_step switches to the slot-major grid when it balances the shares)
        if (const char* ov = getenv("DDX_STEP_XCD")) E.step_xcd = atoi(ov) != 0;
        E.slot_table = 0; E.mcost_rec = 0;
        // the tile pass inside the shading launch (big_worker_wg): hypothesis b's workgroups all on XCD b % 8.
        // Whether this engine uses it is decided by the set-up (engine_setup: are large triangles to be expected at all?)
        e->inline_ok = !desc->separate_big_pass && desc->B % 8 == 0;
        if (const char* ov = getenv("DDX_BIG_INLINE")) e->inline_env = atoi(ov) != 0;
        if (const char* ov = getenv("DDX_TWO_STREAMS")) e->two_streams = atoi(ov);
        if (const char* ov = getenv("DDX_TWO_MIN")) { e->two_min_iters = std::max(2, atoi(ov)); e->two_min_env = true; }
        E.big_inline = 0;
        E.wait_ticks = 2000000u;  // 20 ms of the 100 MHz clock: a tile pass that really runs takes 0.1-2 ms on the heaviest cases measured
        if (const char* ov = getenv("DDX_BIG_WAIT_US")) E.wait_ticks = (unsigned)std::max(1, atoi(ov)) * 100u;
        E.dbg_reverse = 0;
        if (const char* ov = getenv("DDX_DEBUG_REVERSE_SLABS")) E.dbg_reverse = atoi(ov) != 0;
        E.big_workers = 64;
        if (const char* ov = getenv("DDX_BIG_WORKERS")) E.big_workers = std::max(1, atoi(ov));
        E.scatter_mode = 0;
        E.n_meshlets = 0;
        for (int c = 0; c < 6; ++c) E.bbox[c] = 0.f;
    }
    e->small_mesh = mesh_is_small(*desc);
    if (const char* ov = getenv("DDX_STEP_RESIDENT")) e->step_resident = std::max(1, atoi(ov));
    if (const char* ov = getenv("DDX_STEP_BALANCE")) e->balance = atoi(ov);
    if (const char* ov = getenv("DDX_STEP_BALANCE_MIN")) e->balance_min_per_slot = std::max(1, atoi(ov));
    const size_t need = engine_layout(e->dev, *desc, b.scratch);
    if (b.scratch_bytes < need) {
        delete e;
        DDX_REQUIRE(false, DDX_E_SCRATCH, "engine_create: scratch %zu < required %zu bytes", b.scratch_bytes, need);
    }
    *out = e;
    return 0;
}

// sorted internal mesh: vertex attributes gathered into sorted order, triangle / opposite-vertex ids renumbered
__global__ void remap_vertices_kernel(EngineDev E)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= E.d.V) return;
    const int o = E.vold[n];
#pragma unroll
    for (int c = 0; c < 3; ++c) E.spos[(size_t)n * 3 + c] = E.b.pos[(size_t)o * 3 + c];
    if (E.b.uv) { E.suv[(size_t)n * 2 + 0] = E.b.uv[(size_t)o * 2 + 0]; E.suv[(size_t)n * 2 + 1] = E.b.uv[(size_t)o * 2 + 1]; }
    if (E.b.vtx_color)
#pragma unroll
        for (int c = 0; c < 3; ++c) E.scol[(size_t)n * 3 + c] = E.b.vtx_color[(size_t)o * 3 + c];
}

__global__ void remap_triangles_kernel(EngineDev E)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= E.d.T) return;
    const int V = E.d.V;
    auto rm = [&](int i) { return (i >= 0 && i < V) ? E.vnew[i] : i; };  // (out-of-range ids stay out of range: the triangle is ignored)
    const int v0 = rm(E.b.tri[t * 3 + 0]), v1 = rm(E.b.tri[t * 3 + 1]), v2 = rm(E.b.tri[t * 3 + 2]);
    E.stri[t * 3 + 0] = v0; E.stri[t * 3 + 1] = v1; E.stri[t * 3 + 2] = v2;
    E.trirec[t * 2 + 0] = make_int4(v0, v1, v2, rm(E.b.opp[t * 3 + 0]));
    E.trirec[t * 2 + 1] = make_int4(rm(E.b.opp[t * 3 + 1]), rm(E.b.opp[t * 3 + 2]), 0, 0);
    // colour-role record (remap_vertices_kernel ran before this launch)
    float r[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) r[i] = 0.f;
    const int vv[3] = {v0, v1, v2};
    const bool ok = (unsigned)v0 < (unsigned)V && (unsigned)v1 < (unsigned)V && (unsigned)v2 < (unsigned)V;
    if (ok) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int c = 0; c < 3; ++c) r[k * 3 + c] = E.spos[(size_t)vv[k] * 3 + c];
            if (E.d.Th > 0) {
                if (E.b.uv) { r[9 + k * 2] = E.suv[(size_t)vv[k] * 2]; r[10 + k * 2] = E.suv[(size_t)vv[k] * 2 + 1]; }
            } else if (E.b.vtx_color) {
#pragma unroll
                for (int c = 0; c < 3; ++c) r[9 + k * 3 + c] = E.scol[(size_t)vv[k] * 3 + c];
            }
        }
    }
    const int rs = E.d.Th > 0 ? 4 : 5;
    for (int i = 0; i < rs; ++i) E.crec[(size_t)t * rs + i] = make_float4(r[i * 4], r[i * 4 + 1], r[i * 4 + 2], r[i * 4 + 3]);
}

// texq (see EngineDev): one thread per texel
__global__ __launch_bounds__(256) void build_texq_kernel(EngineDev E)
{
    const int Th = E.d.Th, Tw = E.d.Tw;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)Th * Tw) return;
    const int y = (int)(i / Tw), x = (int)(i - (long long)y * Tw);
    const int x1 = x + 1 >= Tw ? 0 : x + 1, y1 = y + 1 >= Th ? 0 : y + 1;
    const float* T = E.b.tex;
    const float *a = T + ((size_t)y * Tw + x) * 3, *b = T + ((size_t)y * Tw + x1) * 3, *c = T + ((size_t)y1 * Tw + x) * 3,
                *dq = T + ((size_t)y1 * Tw + x1) * 3;
    float4* Q = E.texq + (size_t)i * 4;
    Q[0] = make_float4(a[0], a[1], a[2], b[0]);
    Q[1] = make_float4(b[1], b[2], c[0], c[1]);
    Q[2] = make_float4(c[2], dq[0], dq[1], dq[2]);
    Q[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}

static inline float __uint_as_float_host(unsigned u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static int engine_setup(ddx_engine* e, hipStream_t s)
{
    EngineDev& E = e->dev;
    DDX_HIP(hipMemsetAsync(E.st, 0, sizeof(EngineState), s));
    DDX_HIP(hipMemsetAsync(&E.st->sel_key, 0xFF, sizeof(unsigned long long), s));
    DDX_HIP(hipMemsetAsync(E.adam, 0, (size_t)2 * 14 * E.d.B * sizeof(float), s));
    DDX_HIP(hipMemsetAsync(E.L.counters, 0, E.L.zero_bytes, s));  // (both parities) kept zero by update_head afterwards
    DDX_HIP(hipMemsetAsync(E.L.zbuf, 0xFF, E.L.zbuf_bytes, s));   // (both parities) re-armed per active tile by update_head
    {
        const int n_chunks = ddx_cdiv((long long)E.d.H * E.d.W, SETUP_CHUNK);
        setup_part_kernel<<<n_chunks, 256, 0, s>>>(E);
        setup_scan_kernel<<<1, 256, 0, s>>>(E, n_chunks);
        if (E.b.gt_depth) setup_fill_kernel<<<n_chunks, 256, 0, s>>>(E);
    }
    if (!e->mesh_done && E.texq && E.b.tex) build_texq_kernel<<<ddx_cdiv((long long)E.d.Th * E.d.Tw, 256), 256, 0, s>>>(E);
    DDX_LAUNCH_CHECK();
    DDX_HIP(hipMemsetD32Async((hipDeviceptr_t)E.inside, 1, (size_t)E.d.B, s));
    // ---- internal sorted mesh (host, once per engine).  (1) Vertices renumbered in Morton order of their object-space
    // position: the vertex data of neighbouring triangles and pixels become neighbours in memory whatever order the mesh
    // file had.  (2) Processing order of the rasteriser: triangles sorted by the Morton code of their centroid -- the 64-bit
    // atomicMin stream is bound by distinct zbuf lines per instruction; in file order the 128 triangles of a wave may be a
    // long thin strip (or anything), in Morton order they are a compact patch.  Triangle ids are not renumbered.
    if (!e->mesh_done) {
        const int V = E.d.V, T = E.d.T;
        std::vector<float> hpos((size_t)V * 3);
        std::vector<int> htri((size_t)T * 3);
        DDX_HIP(hipMemcpyAsync(hpos.data(), E.b.pos, hpos.size() * sizeof(float), hipMemcpyDeviceToHost, s));
        DDX_HIP(hipMemcpyAsync(htri.data(), E.b.tri, htri.size() * sizeof(int), hipMemcpyDeviceToHost, s));
        DDX_HIP(hipStreamSynchronize(s));
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        for (int v = 0; v < V; ++v)
            for (int c = 0; c < 3; ++c) {
                const float x = hpos[(size_t)v * 3 + c];
                if (x == x) { lo[c] = x < lo[c] ? x : lo[c]; hi[c] = x > hi[c] ? x : hi[c]; }
            }
        auto spread = [](unsigned x) {  // 10 bits -> every third bit
            x &= 1023u;
            x = (x | (x << 16)) & 0x030000FFu;
            x = (x | (x << 8)) & 0x0300F00Fu;
            x = (x | (x << 4)) & 0x030C30C3u;
            x = (x | (x << 2)) & 0x09249249u;
            return x;
        };
        auto morton = [&](const float m[3]) {
            unsigned q[3];
            for (int c = 0; c < 3; ++c) {
                const float ext = hi[c] - lo[c];
                const float u = (ext > 0.f && m[c] == m[c]) ? (m[c] - lo[c]) / ext : 0.f;
                q[c] = (unsigned)(u < 0.f ? 0.f : (u > 1.f ? 1023.f : u * 1023.f));
            }
            return spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
        };
        std::vector<unsigned long long> keys((size_t)(V > T ? V : T));
        std::vector<int> vold((size_t)V), vnew((size_t)V);
        for (int v = 0; v < V; ++v) keys[(size_t)v] = ((unsigned long long)morton(&hpos[(size_t)v * 3]) << 32) | (unsigned)v;
        std::sort(keys.begin(), keys.begin() + V);
        for (int n = 0; n < V; ++n) {
            vold[(size_t)n] = (int)(unsigned)(keys[(size_t)n] & 0xffffffffull);
            vnew[(size_t)vold[(size_t)n]] = n;
        }
        for (int t = 0; t < T; ++t) {
            float m[3] = {0.f, 0.f, 0.f};
            for (int k = 0; k < 3; ++k) {
                const int v = htri[(size_t)t * 3 + k];
                for (int c = 0; c < 3; ++c) m[c] += (v >= 0 && v < V) ? hpos[(size_t)v * 3 + c] * (1.0f / 3.0f) : 0.f;
            }
            keys[(size_t)t] = ((unsigned long long)morton(m) << 32) | (unsigned)t;
        }
        std::sort(keys.begin(), keys.begin() + T);
        // ---- meshlets: the triangles in Morton order of their centroids (the fragments of a wave's triangles then share zbuf
        // lines: the 64-bit atomicMin stream is bound by distinct lines per instruction), cut greedily into runs of at most
        // mesh_ntri triangles over at most mesh_nvc distinct vertices.  Triangles with an index outside [0, V) are left out
        // (they draw nothing).  A vertex is OWNED by the first meshlet that uses it: that workgroup stores its clip / snap.
        {
            const int NTRI = E.mesh_ntri, NVC = E.mesh_nvc;
            auto rm = [&](int i) { return vnew[(size_t)i]; };
            std::vector<float4> hv;
            std::vector<int2> ht;
            std::vector<int> slot_of((size_t)V, -1), stamp((size_t)V, -1);
            std::vector<char> owned((size_t)V, 0);
            std::vector<float> spos((size_t)V * 3);
            for (int n = 0; n < V; ++n)
                for (int c = 0; c < 3; ++c) spos[(size_t)n * 3 + c] = hpos[(size_t)vold[(size_t)n] * 3 + c];
            int M = 0, nt = 0, nv = 0;
            auto open_meshlet = [&]() {
                hv.resize((size_t)(M + 1) * NVC, make_float4(0.f, 0.f, 0.f, __uint_as_float_host(0xffffffffu)));
                ht.resize((size_t)(M + 1) * NTRI, make_int2(0, -1));
                nt = 0; nv = 0;
            };
            open_meshlet();
            for (int i = 0; i < T; ++i) {
                const int t = (int)(unsigned)(keys[(size_t)i] & 0xffffffffull);
                const int o[3] = {htri[(size_t)t * 3 + 0], htri[(size_t)t * 3 + 1], htri[(size_t)t * 3 + 2]};
                if (o[0] < 0 || o[1] < 0 || o[2] < 0 || o[0] >= V || o[1] >= V || o[2] >= V) continue;
                const int v[3] = {rm(o[0]), rm(o[1]), rm(o[2])};
                int fresh = 0;
                for (int k = 0; k < 3; ++k) {
                    bool seen = stamp[(size_t)v[k]] == M;
                    for (int k2 = 0; k2 < k; ++k2) seen = seen || v[k2] == v[k];
                    fresh += !seen;
                }
                if (nt == NTRI || nv + fresh > NVC) {
                    ++M;
                    open_meshlet();
                }
                int l[3];
                for (int k = 0; k < 3; ++k) {
                    if (stamp[(size_t)v[k]] != M) {
                        stamp[(size_t)v[k]] = M;
                        slot_of[(size_t)v[k]] = nv;
                        unsigned bits = (unsigned)v[k];
                        if (!owned[(size_t)v[k]]) { owned[(size_t)v[k]] = 1; bits |= 0x80000000u; }
                        hv[(size_t)M * NVC + nv] = make_float4(spos[(size_t)v[k] * 3], spos[(size_t)v[k] * 3 + 1], spos[(size_t)v[k] * 3 + 2], __uint_as_float_host(bits));
                        ++nv;
                    }
                    l[k] = slot_of[(size_t)v[k]];
                }
                ht[(size_t)M * NTRI + nt] = make_int2(l[0] | (l[1] << 10) | (l[2] << 20), t);
                ++nt;
            }
            E.n_meshlets = M + 1;
            // ---- final vertex numbering (round 4): the vertices a meshlet OWNS become one contiguous run of ids, in the order of
            // its first slots, so that the owner's clip / snap stores of a wave are whole lines (lane t of the meshlet's workgroup
            // stores vertex base + t): with the Morton ids the owned vertices of a meshlet were interleaved with its neighbours'
            // and the stores were partial lines (WRITE_SIZE 40.7 MB per step launch on cfg2 against 22 MB of payload).  Slots are
            // permuted (owned first, first-use order kept), the triangles' local indices follow, and vold / vnew are composed
            // with the renumbering; vertices no triangle uses keep their Morton order behind all the others.
            {
                const int Mn = E.n_meshlets;
                std::vector<int> fin((size_t)V, -1);  // Morton id -> final id
                int next = 0;
                for (int m = 0; m < Mn; ++m) {
                    int perm[1024];  // old slot -> new slot (NVC <= 512)
                    int n_own = 0, n_all = 0;
                    for (int k = 0; k < NVC; ++k) {
                        const unsigned bits = [&] { unsigned u; memcpy(&u, &hv[(size_t)m * NVC + k].w, 4); return u; }();
                        if (bits == 0xffffffffu) break;
                        ++n_all;
                        if (bits >> 31) ++n_own;
                    }
                    int io = 0, ib = n_own;
                    for (int k = 0; k < n_all; ++k) {
                        unsigned bits; memcpy(&bits, &hv[(size_t)m * NVC + k].w, 4);
                        if (bits >> 31) { fin[(size_t)(bits & 0x7fffffffu)] = next + io; perm[k] = io++; }
                        else perm[k] = ib++;
                    }
                    next += n_own;
                    std::vector<float4> tmp(hv.begin() + (size_t)m * NVC, hv.begin() + (size_t)m * NVC + n_all);
                    for (int k = 0; k < n_all; ++k) hv[(size_t)m * NVC + perm[k]] = tmp[(size_t)k];
                    for (int k = 0; k < NTRI; ++k) {
                        int2& tr = ht[(size_t)m * NTRI + k];
                        if (tr.y < 0) continue;
                        const int l0 = tr.x & 1023, l1 = (tr.x >> 10) & 1023, l2 = (tr.x >> 20) & 1023;
                        tr.x = perm[l0] | (perm[l1] << 10) | (perm[l2] << 20);
                    }
                }
                for (int n = 0; n < V; ++n)
                    if (fin[(size_t)n] < 0) fin[(size_t)n] = next++;  // (unreferenced vertices)
                for (size_t i = 0; i < hv.size(); ++i) {
                    unsigned bits; memcpy(&bits, &hv[i].w, 4);
                    if (bits == 0xffffffffu) continue;
                    bits = (bits & 0x80000000u) | (unsigned)fin[(size_t)(bits & 0x7fffffffu)];
                    memcpy(&hv[i].w, &bits, 4);
                }
                std::vector<int> vold2((size_t)V);
                for (int n = 0; n < V; ++n) vold2[(size_t)fin[(size_t)n]] = vold[(size_t)n];
                vold.swap(vold2);
                for (int n = 0; n < V; ++n) vnew[(size_t)vold[(size_t)n]] = n;
            }
            // interleave the two ends of the Morton curve: 0, M-1, 1, M-2, ...  The bit-complement of a Morton code is the point
            // reflection through the centre of the bounding box, so consecutive meshlets of the new order are (roughly) antipodes
            // of the object: a workgroup's consecutive meshlets are one front-facing and one back-facing patch in any pose
            {
                const int Mn = E.n_meshlets;
                std::vector<float4> hv2(hv.size());
                std::vector<int2> ht2(ht.size());
                for (int i = 0; i < Mn; ++i) {
                    const int src = (i & 1) ? Mn - 1 - i / 2 : i / 2;
                    std::copy(hv.begin() + (size_t)src * NVC, hv.begin() + (size_t)(src + 1) * NVC, hv2.begin() + (size_t)i * NVC);
                    std::copy(ht.begin() + (size_t)src * NTRI, ht.begin() + (size_t)(src + 1) * NTRI, ht2.begin() + (size_t)i * NTRI);
                }
                hv.swap(hv2);
                ht.swap(ht2);
            }
            DDX_HIP(hipMemcpyAsync(E.mvert, hv.data(), hv.size() * sizeof(float4), hipMemcpyHostToDevice, s));
            DDX_HIP(hipMemcpyAsync(E.mtri, ht.data(), ht.size() * sizeof(int2), hipMemcpyHostToDevice, s));
            DDX_HIP(hipMemcpyAsync(E.vold, vold.data(), vold.size() * sizeof(int), hipMemcpyHostToDevice, s));
            DDX_HIP(hipMemcpyAsync(E.vnew, vnew.data(), vnew.size() * sizeof(int), hipMemcpyHostToDevice, s));
            remap_vertices_kernel<<<ddx_cdiv(V, 256), 256, 0, s>>>(E);
            remap_triangles_kernel<<<ddx_cdiv(T, 256), 256, 0, s>>>(E);
            DDX_LAUNCH_CHECK();
            DDX_HIP(hipStreamSynchronize(s));  // (the host vectors above are the copy sources)
        }
        // size of the mesh's triangles in object space (engine_setup's estimate of whether a hypothesis will have LARGE triangles)
        {
            double area = 0.0, emax = 0.0;
            for (int t = 0; t < T; ++t) {
                const int i0 = htri[(size_t)t * 3], i1 = htri[(size_t)t * 3 + 1], i2 = htri[(size_t)t * 3 + 2];
                if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= V || i1 >= V || i2 >= V) continue;
                double p[3][3];
                const int iv[3] = {i0, i1, i2};
                for (int k = 0; k < 3; ++k)
                    for (int c = 0; c < 3; ++c) p[k][c] = hpos[(size_t)iv[k] * 3 + c];
                double e1[3], e2[3], e3[3];
                for (int c = 0; c < 3; ++c) { e1[c] = p[1][c] - p[0][c]; e2[c] = p[2][c] - p[0][c]; e3[c] = p[2][c] - p[1][c]; }
                const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                const double a2 = std::sqrt(cx * cx + cy * cy + cz * cz);
                if (std::isfinite(a2)) area += 0.5 * a2;
                for (const double* ed : {e1, e2, e3}) {
                    const double l = std::sqrt(ed[0] * ed[0] + ed[1] * ed[1] + ed[2] * ed[2]);
                    if (std::isfinite(l) && l > emax) emax = l;
                }
            }
            e->mesh_area = area;
            e->mesh_max_edge = emax;
        }
        // object-space bounding box of ALL vertices (the view-volume test of the culling rule, EngineDev::cull_sign)
        bool finite = true;
        for (int v = 0; v < V; ++v)
            for (int c = 0; c < 3; ++c) finite = finite && std::isfinite(hpos[(size_t)v * 3 + c]);
        for (int c = 0; c < 3; ++c) { E.bbox[c] = finite ? lo[c] : 0.f; E.bbox[3 + c] = finite ? hi[c] : 0.f; }
        // ---- back-face culling (RasterScratch::cull_sign): only for a CLOSED, consistently oriented surface.  Vertices are
        // welded by position (bit-equal coordinates: uv seams duplicate vertices), triangles with two welded corners equal are
        // ignored (they have no area in any view), and every welded edge must then belong to exactly two triangles that run
        // through it in opposite directions.  Which snapped-area sign is a back face: a triangle p0 p1 p2 in camera space has
        // det [p0; p1; p2] > 0 exactly when its counter-clockwise normal points away from the camera; the projection maps that
        // determinant to the screen orientation times det A, A = the x, y, w rows of proj (their 4th column must be zero, as
        // in any pinhole projection); and counter-clockwise is outward when the signed volume is positive.
        E.cull_sign = 0;
        bool want = !E.d.no_backface_cull && finite;
        if (const char* ov = getenv("DDX_NO_CULL")) want = want && !atoi(ov);
        if (want) {
            std::vector<unsigned long long> vk((size_t)V);
            std::vector<int> canon((size_t)V), order((size_t)V);
            for (int v = 0; v < V; ++v) order[(size_t)v] = v;
            auto bits = [&](int v, int c) { unsigned u; float f = hpos[(size_t)v * 3 + c] + 0.0f; memcpy(&u, &f, 4); return u; };  // (+0: -0 == 0)
            std::sort(order.begin(), order.end(), [&](int a, int b) {
                for (int c = 0; c < 3; ++c) { const unsigned x = bits(a, c), y = bits(b, c); if (x != y) return x < y; }
                return a < b;
            });
            for (int i = 0; i < V; ++i) {
                const int v = order[(size_t)i];
                const bool same = i > 0 && bits(v, 0) == bits(order[(size_t)i - 1], 0) && bits(v, 1) == bits(order[(size_t)i - 1], 1) &&
                                  bits(v, 2) == bits(order[(size_t)i - 1], 2);
                canon[(size_t)v] = same ? canon[(size_t)order[(size_t)i - 1]] : v;
            }
            bool closed = true;
            double vol6 = 0.0;
            // shells (connected components over welded vertices) must all be oriented the same way: one inside-out shell next
            // to a regular one would lose its visible faces
            std::vector<int> shell((size_t)V);
            for (int i = 0; i < V; ++i) shell[(size_t)i] = i;
            auto root = [&](int v) {
                while (shell[(size_t)v] != v) { shell[(size_t)v] = shell[(size_t)shell[(size_t)v]]; v = shell[(size_t)v]; }
                return v;
            };
            std::vector<double> tvol((size_t)T, 0.0);
            std::vector<unsigned long long> ek;  // (lo << 32 | hi) << 1 | direction
            ek.reserve((size_t)T * 3);
            for (int t = 0; t < T && closed; ++t) {
                const int i0 = htri[(size_t)t * 3], i1 = htri[(size_t)t * 3 + 1], i2 = htri[(size_t)t * 3 + 2];
                if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= V || i1 >= V || i2 >= V) { closed = false; break; }
                const int c[3] = {canon[(size_t)i0], canon[(size_t)i1], canon[(size_t)i2]};
                if (c[0] == c[1] || c[1] == c[2] || c[0] == c[2]) continue;
                const double x0 = hpos[(size_t)i0 * 3], y0 = hpos[(size_t)i0 * 3 + 1], z0 = hpos[(size_t)i0 * 3 + 2];
                const double x1 = hpos[(size_t)i1 * 3], y1 = hpos[(size_t)i1 * 3 + 1], z1 = hpos[(size_t)i1 * 3 + 2];
                const double x2 = hpos[(size_t)i2 * 3], y2 = hpos[(size_t)i2 * 3 + 1], z2 = hpos[(size_t)i2 * 3 + 2];
                tvol[(size_t)t] = x0 * (y1 * z2 - z1 * y2) - y0 * (x1 * z2 - z1 * x2) + z0 * (x1 * y2 - y1 * x2);
                vol6 += tvol[(size_t)t];
                { const int ra = root(c[0]), rb = root(c[1]); if (ra != rb) shell[(size_t)ra] = rb; }
                { const int ra = root(c[1]), rb = root(c[2]); if (ra != rb) shell[(size_t)ra] = rb; }
                for (int k = 0; k < 3; ++k) {
                    const unsigned a = (unsigned)c[k], bq = (unsigned)c[(k + 1) % 3];
                    const unsigned lo = a < bq ? a : bq, hi = a < bq ? bq : a;
                    ek.push_back(((((unsigned long long)lo << 31) | hi) << 1) | (a < bq ? 1u : 0u));  // (V < 2^31)
                }
            }
            if (closed) {
                std::sort(ek.begin(), ek.end());
                closed = !ek.empty() && ek.size() % 2 == 0;
                for (size_t i = 0; closed && i < ek.size(); i += 2)
                    closed = (ek[i] >> 1) == (ek[i + 1] >> 1) && (ek[i] & 1) != (ek[i + 1] & 1) && (i + 2 >= ek.size() || (ek[i + 2] >> 1) != (ek[i] >> 1));
            }
            if (getenv("DDX_DEBUG_CULL")) fprintf(stderr, "ddx cull: edges paired %d\n", (int)closed);
            if (closed) {
                // every shell must enclose a volume of the sign of the whole -- and a real one: a flat two-sided patch (the same
                // triangles wound both ways) is "closed" with a volume that is round-off of either sign
                std::vector<double> svol((size_t)V, 0.0), sabs((size_t)V, 0.0);
                for (int t = 0; t < T; ++t)
                    if (tvol[(size_t)t] != 0.0) {
                        const size_t r = (size_t)root(canon[(size_t)htri[(size_t)t * 3]]);
                        svol[r] += tvol[(size_t)t];
                        sabs[r] += std::fabs(tvol[(size_t)t]);
                    }
                for (int v = 0; v < V && closed; ++v)
                    if (sabs[(size_t)v] > 0.0 && (!(std::fabs(svol[(size_t)v]) > 1e-9 * sabs[(size_t)v]) || (svol[(size_t)v] > 0.0) != (vol6 > 0.0))) closed = false;
            }
            float hp[16];
            DDX_HIP(hipMemcpyAsync(hp, E.b.proj, sizeof(hp), hipMemcpyDeviceToHost, s));
            DDX_HIP(hipStreamSynchronize(s));
            const double detA = (double)hp[0] * ((double)hp[5] * hp[14] - (double)hp[6] * hp[13]) - (double)hp[1] * ((double)hp[4] * hp[14] - (double)hp[6] * hp[12]) +
                                (double)hp[2] * ((double)hp[4] * hp[13] - (double)hp[5] * hp[12]);
            const bool pinhole = hp[3] == 0.f && hp[7] == 0.f && hp[15] == 0.f;
            if (getenv("DDX_DEBUG_CULL")) fprintf(stderr, "ddx cull: closed %d pinhole %d vol6 %g detA %g edges %zu\n", (int)closed, (int)pinhole, vol6, detA, ek.size());
            if (closed && pinhole && vol6 != 0.0 && detA != 0.0 && std::isfinite(vol6) && std::isfinite(detA))
                E.cull_sign = ((vol6 > 0.0) == (detA > 0.0)) ? 1 : -1;
        }
        e->mesh_done = true;
    }
    // ---- which scatter variant: expected covered centres per triangle, from the observed segmentation mask (the hypotheses
    // render the object at about the observed size; front and back faces both produce fragments).  Above about one centre per
    // triangle the fragment-exchange variant wins (cfg2 at half the distance: 42 -> 37 us, at a third: 115 -> 63 us); below it
    // costs occupancy (cfg3ref: 44 -> 52 us).  DDX_SCATTER_EXCHANGE=0/1/2 overrides (tuning).
    {
        EngineState hst;
        DDX_HIP(hipMemcpyAsync(&hst, E.st, sizeof(EngineState), hipMemcpyDeviceToHost, s));
        DDX_HIP(hipStreamSynchronize(s));
        // sorted seg list + prefix sums (see EngineDev::seg_gd)
        E.nseg = 0;
        std::vector<double> hW(1, 0.0), hG(1, 0.0);
        std::vector<float> hgd;
        if (E.d.use_depth) {
            const int n = hst.n_seg;
            std::vector<float2> hs((size_t)n);
            if (n > 0) DDX_HIP(hipMemcpyAsync(hs.data(), E.seglist, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost, s));
            DDX_HIP(hipStreamSynchronize(s));
            std::sort(hs.begin(), hs.end(), [](const float2& a, const float2& b) { return a.x < b.x; });
            hgd.resize((size_t)n); hW.resize((size_t)n + 1); hG.resize((size_t)n + 1);
            for (int i = 0; i < n; ++i) {
                const double w = std::fabs((double)hs[(size_t)i].y);
                hgd[(size_t)i] = hs[(size_t)i].x;
                hW[(size_t)i + 1] = hW[(size_t)i] + w;
                hG[(size_t)i + 1] = hG[(size_t)i] + w * (double)hs[(size_t)i].x;
            }
            E.nseg = n;
            if (n > 0) DDX_HIP(hipMemcpyAsync(E.seg_gd, hgd.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice, s));
        }
        DDX_HIP(hipMemcpyAsync(E.seg_W, hW.data(), hW.size() * sizeof(double), hipMemcpyHostToDevice, s));
        DDX_HIP(hipMemcpyAsync(E.seg_G, hG.data(), hG.size() * sizeof(double), hipMemcpyHostToDevice, s));
        DDX_HIP(hipStreamSynchronize(s));  // (host vectors are the copy sources)
        const double per_tri = 2.0 * (hst.c_mask / 3.0) / (double)std::max(E.d.T, 1);
        E.scatter_mode = per_tri > SCATTER_EXCHANGE_PER_TRI ? 2 : 0;
        // micro-polygon regime on a launch of three or more rounds of resident workgroups (cfg3: 12 800 workgroups, cfg50k64: 6 400):
        // the compacting variant (3) -- survivors of the bbox / area / back-face tests packed into full waves before the coverage
        // and fragment code
        if (!E.scatter_mode && (long long)E.n_meshlets * E.d.B >= 6000) E.scatter_mode = 3;
        if (const char* ov = getenv("DDX_SCATTER_EXCHANGE")) { const int v = atoi(ov); E.scatter_mode = v <= 0 ? 0 : (v >= 2 ? 3 : 2); }  // 0 plain, 1 hybrid, 2 compacting
        if (const char* ov = getenv("DDX_SCATTER_MODE")) { const int v = atoi(ov); E.scatter_mode = (v == 2 || v == 3) ? v : 0; }
        // ---- where the tile pass for LARGE triangles runs.  As its own launch between step_kernel and shade_kernel it costs a kernel
        // boundary in every iteration (2.4 of cfg2's 43 us) and exits at once when the batch has no such triangle -- the dense meshes of
        // the benchmark never have one.  So: when no large triangle is to be EXPECTED, the launch is dropped and shade_kernel runs the
        // pass itself for a hypothesis that turns out to have some (big_inline_pass: correct, slower than the launch).  Expected size:
        // the observed object covers n_px pixels and shows about a quarter of the mesh's area (a convex body projects to area / 4 on
        // average), so one unit of object-space length is about sqrt(4 n_px / area) pixels and the longest edge of the mesh spans
        // that many pixels times its length; a triangle goes to the tile pass when its bounding box holds more than 64 pixel centres.
        {
            const double n_px = hst.c_mask / 3.0;
            const double px_per_unit = (e->mesh_area > 0.0 && n_px > 0.0) ? std::sqrt(4.0 * n_px / e->mesh_area) : 1e30;
            const double edge_px = e->mesh_max_edge * px_per_unit;
            bool expect_none = edge_px < 6.0;  // (a bounding box of 36 centres where 64 are allowed: hypotheses start nearer than the object is)
            if (e->inline_env >= 0) expect_none = e->inline_env != 0;
            E.big_inline = (e->inline_ok && expect_none) ? 1 : 0;
            if (getenv("DDX_DEBUG_INLINE")) fprintf(stderr, "ddx: longest edge ~%.2f px, inline tile pass %d (allowed %d)\n", edge_px, E.big_inline, (int)e->inline_ok);
        }
        // grid z order of the shading launch (engine_create: mask role first).  On CLOSE-UPS of a textured object the other order is
        // 5 % faster (object at 7-21 % of the frame: 9.76 -> 10.25, 7.78 -> 8.15, 5.97 -> 6.28, 4.53 -> 4.75 k it/s) and 3-5 % slower
        // below (1.2 %, 4.7 %: the crossover lies at 12-14 tiles per shading workgroup) and on untextured large-triangle meshes.
        // The tile count of a hypothesis is about that of the observed object: 1.35 tiles per 256 observed pixels.
        if (E.n_roles == 2) {
            const double tiles_per_wg = 1.35 * (hst.c_mask / 3.0) / 256.0 / (double)std::max(E.s_shade, 1);
            bool colour_first = E.d.use_rgb && E.d.Th > 0 && tiles_per_wg > 12.0;
            if (const char* ov = getenv("DDX_COLOUR_FIRST")) colour_first = atoi(ov) != 0;
            const int first = colour_first ? 0 : 1;
            if (E.roles[0] != first && (E.roles[1] == first)) { std::swap(E.roles[0], E.roles[1]); E.st_role = E.roles[0]; }
        }
    }
    e->setup_done = true;
    ++e->setup_gen;
    return 0;
}

static int engine_run_impl(ddx_engine* e, int it0, int n, int use_graph, void* stream, float* sel_out, int sel_lo);

// the engine's own stream for the second chain of a two-stream run (engine_run_impl), created -- and its queue woken -- by the
// set-up of an engine that may use it: a stream's first submission costs milliseconds
__global__ void side_spin_kernel(unsigned ticks)  // one wave busy for `ticks` of the 100 MHz clock
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}

static bool two_streams_possible(const ddx_engine* e)
{
    const EngineDev& E = e->dev;
    if (!(e->two_streams > 0 && !E.d.single_stream && E.big_inline && E.d.B >= 32 && E.d.B % 16 == 0)) return false;
    // Two half launches side by side win while a launch is short enough for its fixed parts (prologue, boundary, tail) to matter:
    // by (meshlets x hypotheses) of the step launch, tools/two_stream_threshold.py at 100 iterations -- 2 560 (cfg2) -6 %, 3 200
    // -13 %, 3 776 (cfg4) -4.6 %, 5 120 (cfg2 / cfg5 with 128 hypotheses) -6 %, 6 400 (cfg50k64) -2 %, 7 552 (cfg4 with 128) +4.4 %,
    // 12 800 (cfg3 / cfg3ref / cfg50k64 with 128) +1.5 % / -1 % / 0.
    return e->two_min_env || (long long)E.n_meshlets * E.d.B < 7000;
}

static int two_streams_min_iters(const ddx_engine* e)
{
    // (fork + join cost 25-30 us per run: even at 14-16 iterations for the short launches; cfg50k64 +10 us at 20, -31 at 48)
    if (e->two_min_env) return e->two_min_iters;
    return (long long)e->dev.n_meshlets * e->dev.d.B < 6000 ? e->two_min_iters : std::max(e->two_min_iters, 48);
}

// Do kernels of `cand` run BESIDE kernels of `s`?  HIP multiplexes its streams onto a few hardware queues, and two streams that
// share one take turns: a two-chain run is then not 5 % faster but 1.7x SLOWER than one chain (measured: a bench process with an
// RCCL process group, whose streams had shifted the assignment -- 73 instead of 43 us per iteration; GPU_MAX_HW_QUEUES=8 likewise).
// Nothing in the API tells; so: one 30-us single-wave kernel on each, started together, timed with events -- side by side they
// end 38-41 us after the fork (the second stream starts 8-10 us late: its event wait), in turn after 70-85.  Synchronises both
// streams (set-up time).
// THE REGISTRY (round 5): the answer belongs to the pair (device, caller stream), not to an engine -- bop.refine_frame builds its
// engines anew for every frame, and each used to create up to six streams and time three pairs of spin kernels at its set-up.
// One entry per pair for the life of the process: the stream that was found to run beside the caller's (or none), handed to every
// engine that meets that caller stream; engines on different caller streams (one object per stream) keep different second streams.
struct SideEntry { int device; hipStream_t caller, side; bool ok; };
static std::mutex g_side_mu;
static std::vector<SideEntry> g_side;
static hipEvent_t g_side_ev[3] = {nullptr, nullptr, nullptr};  // (timing events of the probe; used under g_side_mu)

static int side_runs_beside(hipStream_t s, hipStream_t cand, bool* ok)
{
    if (!g_side_ev[0])
        for (int i = 0; i < 3; ++i) DDX_HIP(hipEventCreate(&g_side_ev[i]));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        DDX_HIP(hipEventRecord(g_side_ev[0], s));
        DDX_HIP(hipStreamWaitEvent(cand, g_side_ev[0], 0));
        side_spin_kernel<<<1, 64, 0, s>>>(3000u);
        side_spin_kernel<<<1, 64, 0, cand>>>(3000u);
        DDX_HIP(hipEventRecord(g_side_ev[1], s));
        DDX_HIP(hipEventRecord(g_side_ev[2], cand));
        DDX_HIP(hipEventSynchronize(g_side_ev[1]));
        DDX_HIP(hipEventSynchronize(g_side_ev[2]));
        float t1 = 0.f, t2 = 0.f;
        DDX_HIP(hipEventElapsedTime(&t1, g_side_ev[0], g_side_ev[1]));
        DDX_HIP(hipEventElapsedTime(&t2, g_side_ev[0], g_side_ev[2]));
        best = std::min(best, std::max(t1, t2));  // (the best of three: an unrelated launch in between must not fail a good pair)
    }
    *ok = best < 0.050f;  // ms
    if (getenv("DDX_DEBUG_INLINE")) fprintf(stderr, "ddx: second stream %p beside %p: %.1f us for two 30-us kernels -> %s\n", (void*)cand, (void*)s, best * 1e3f, *ok ? "beside" : "in turn");
    return 0;
}

// the second stream for caller stream `s`: looked up in the registry; searched for at the first call for that pair (up to 6
// candidates until one runs beside `s`; none: runs on `s` keep one chain)
static int ensure_side_stream(ddx_engine* e, hipStream_t s, bool* usable)
{
    *usable = false;
    if (e->two_streams <= 0) return 0;
    int dev = 0;
    DDX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_side_mu);
    const SideEntry* hit = nullptr;
    for (const auto& en : g_side)
        if (en.device == dev && en.caller == s) { hit = &en; break; }
    if (!hit) {
        SideEntry en{dev, s, nullptr, false};
        std::vector<hipStream_t> rejected;
        int err = 0;
        bool ok = false;
        for (int attempt = 0; attempt < 6 && !ok && !err; ++attempt) {
            hipStream_t cand = nullptr;
            const hipError_t ce = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking);
            if (ce != hipSuccess) {
                ddx_set_error("hipStreamCreateWithFlags failed: %s (%s:%d)", hipGetErrorString(ce), __FILE__, __LINE__);
                err = (int)ce;
                break;
            }
            err = side_runs_beside(s, cand, &ok);
            if (!err && ok) en.side = cand;
            else rejected.push_back(cand);  // (kept until the search is over: a destroyed stream's queue slot would be handed out again)
        }
        for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
        if (err) return err;
        en.ok = en.side != nullptr;
        g_side.push_back(en);
        hit = &g_side.back();
    }
    e->probe_outcome = hit->ok ? 1 : 0;
    if (!hit->ok) return 0;
    e->side = hit->side;  // (the registry's: never destroyed by an engine)
    if (!e->ev_fork) {
        DDX_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        DDX_HIP(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    }
    *usable = true;
    return 0;
}

extern "C" int ddx_engine_run(ddx_engine* e, int it0, int n, int use_graph, void* stream)
{
    return engine_run_impl(e, it0, n, use_graph, stream, nullptr, 0);
}

extern "C" int ddx_engine_run_select(ddx_engine* e, int it0, int n, int use_graph, int lo, float* out18, void* stream)
{
    DDX_REQUIRE(out18, DDX_E_NULL, "engine_run_select: NULL out18");
    DDX_REQUIRE(n >= 1, DDX_E_SHAPE, "engine_run_select: n = %d (the selection rides on the run's last kernel)", n);
    return engine_run_impl(e, it0, n, use_graph, stream, out18, lo);
}

static int engine_run_impl(ddx_engine* e, int it0, int n, int use_graph, void* stream, float* sel_out, int sel_lo)
{
    DDX_REQUIRE(e, DDX_E_NULL, "engine_run: NULL engine");
    e->fwd_cached_it = -1;
    DDX_REQUIRE(it0 >= 0 && n >= 0 && it0 + n <= e->dev.d.max_iters, DDX_E_SHAPE, "engine_run: iterations [%d,%d) exceed max_iters=%d", it0,
                it0 + n, e->dev.d.max_iters);
    hipStream_t s = (hipStream_t)stream;
    if (!e->setup_done) {
        if (int err = engine_setup(e, s)) return err;
        if (two_streams_possible(e)) {
            bool usable = false;  // (the set-up has just synchronised: the place for the probe)
            if (int err = ensure_side_stream(e, s, &usable)) return err;
        }
    }
    if (n == 0) return 0;
    RoctxRange rr("ddx_engine_run");
    if (int err = run_prologue(e, it0, s)) return err;
    // iteration it0 is drawn from the caller's parameters; each later step_kernel first steps the optimiser for the iteration
    // before it; finish_kernel steps it for the last one
    // Long runs: the iterations after the first as two chains of half-batch launches, one on the caller's stream and one on a
    // stream of the engine's own, forked from and joined to the caller's by events -- one chain's kernel boundaries and launch
    // prologues are covered by the other chain's work (cfg2: 40.0 -> 37.8 us per iteration; fork + join cost 25-30 us per run: even
    // at 14-16 iterations, 1.5 % ahead at 20, 4.7 % at 48, 6 % at 100 -- tools/two_stream_threshold.py --: hence two_min_iters).
    // Every hypothesis runs the slots, slices and sums it runs in the full launches -- the same bits --; the words the halves share
    // are the status counters (rewritten by finish_kernel after the join) and the tile pass's global "a large triangle exists"
    // word, which only the separate big_pass_kernel reads: hence big_inline only.
    // (Forking before the first iteration as well measured 7 us worse per run.)
    bool capturing = false;
    {
        hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
        capturing = hipStreamIsCapturing(s, &cst) == hipSuccess && cst != hipStreamCaptureStatusNone;
    }
    bool two = two_streams_possible(e) && !use_graph && !capturing && n >= two_streams_min_iters(e) && !e->dev.trace;
    if (two)  // (a caller stream this engine has not met yet is probed here, once: synchronises it)
        if (int err = ensure_side_stream(e, s, &two)) return err;
    if (int err = launch_step(e, STEP_FIRST, it0, s)) return err;
    if (int err = launch_rest(e, it0, s, nullptr)) return err;
    if (use_graph && !e->exec && n > 1) {
        hipStream_t cs;
        DDX_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        DDX_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int err = 0;
        e->graph_chunk = std::max(1, std::min(use_graph, 64));  // use_graph = iterations per captured graph
        for (int k = 0; k < e->graph_chunk && !err; ++k) err = run_iteration(e, -1, cs);
        hipError_t ce = hipStreamEndCapture(cs, &e->graph);
        if (err || ce != hipSuccess) {
            (void)hipStreamDestroy(cs);
            if (err) return err;
            DDX_HIP(ce);
        }
        DDX_HIP(hipGraphInstantiate(&e->exec, e->graph, nullptr, nullptr, 0));
        DDX_HIP(hipStreamDestroy(cs));
    }
    if (two) {
        DDX_HIP(hipEventRecord(e->ev_fork, s));
        DDX_HIP(hipStreamWaitEvent(e->side, e->ev_fork, 0));
        int err = 0;
        for (int i = 1; i < n && !err; ++i) {
            err = launch_step(e, STEP_NORMAL, it0 + i, s, 0);
            if (!err) err = launch_step(e, STEP_NORMAL, it0 + i, e->side, 1);
            if (!err) err = launch_rest(e, it0 + i, s, nullptr, 0);
            if (!err) err = launch_rest(e, it0 + i, e->side, nullptr, 1);
        }
        // (joined on the error path too: whatever did go out on the engine's stream stays ordered before the caller's next work)
        DDX_HIP(hipEventRecord(e->ev_join, e->side));
        DDX_HIP(hipStreamWaitEvent(s, e->ev_join, 0));
        if (err) return err;
    }
    for (int i = two ? n : 1; i < n;) {
        if (use_graph && e->exec && i + e->graph_chunk <= n) {
            DDX_HIP(hipGraphLaunch(e->exec, s));
            i += e->graph_chunk;
        } else {
            if (int err = run_iteration(e, it0 + i, s)) return err;
            ++i;
        }
    }
    e->dev.sel_out = sel_out;  // (finish_kernel's by-value copy of the arguments carries it)
    e->dev.sel_lo = sel_lo;
    const int ferr = launch_finish(e, it0 + n, s);
    e->dev.sel_out = nullptr;
    if (ferr) return ferr;
    e->adam_parity = (it0 + n) & 1;
    e->last.kind = 1; e->last.it0 = it0; e->last.n = n; e->last.use_graph = use_graph; e->last.sel_out = sel_out; e->last.sel_lo = sel_lo;
    return 0;
}

// The host half of the bounded wait (big_wait): synchronises `stream`, reads the engine's flag word, and if a shading workgroup of
// the last run gave up waiting for the in-launch tile pass: clears the flag, moves the tile pass of this engine into its own launch
// for good, puts parameters and optimiser state back to what the run started from (run_snap, written by the run's first launch) and
// runs it again -- same iterations, same selection target -- then synchronises once more.  Returns 0 (nothing to do), 1 (the run
// was repeated), or an error code.  An evaluation pass that timed out is simply repeated by its caller after this has returned 1.
static int read_flags(ddx_engine* e, hipStream_t s, int* flags)
{
    DDX_HIP(hipStreamSynchronize(s));
    DDX_HIP(hipMemcpy(flags, &e->dev.st->flags, sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

static int disable_inline(ddx_engine* e, hipStream_t s)
{
    const int zero = 0;
    DDX_HIP(hipMemcpy(&e->dev.st->flags, &zero, sizeof(int), hipMemcpyHostToDevice));
    e->inline_ok = false;
    e->dev.big_inline = 0;
    // a captured graph holds launches of the in-launch form: captured again by the next run that asks for one
    if (e->exec) { (void)hipGraphExecDestroy(e->exec); e->exec = nullptr; }
    if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
    ++e->setup_gen;  // (a group compares generations: its table row of this member is stale now)
    (void)s;
    return 0;
}

static int restore_run_start(ddx_engine* e, int it0, hipStream_t s)
{
    EngineDev& E = e->dev;
    const size_t B = (size_t)E.d.B;
    const int par = it0 & 1;
    DDX_HIP(hipMemcpyAsync(E.b.params, E.run_snap, 7 * B * sizeof(float), hipMemcpyDeviceToDevice, s));
    DDX_HIP(hipMemcpyAsync(E.adam + (size_t)par * 14 * B, E.run_snap + 7 * B, 14 * B * sizeof(float), hipMemcpyDeviceToDevice, s));
    e->adam_parity = par;
    return 0;
}

extern "C" int ddx_engine_run_check(ddx_engine* e, void* stream)
{
    DDX_REQUIRE(e, DDX_E_NULL, "engine_run_check: NULL engine");
    hipStream_t s = (hipStream_t)stream;
    if (!e->setup_done) return 0;
    int flags = 0;
    if (int err = read_flags(e, s, &flags)) return err;
    if (!(flags & ENGINE_FLAG_INLINE_TIMEOUT)) {
        e->last.kind = 0;  // (checked and clean: nothing to repeat, whatever times out later)
        return 0;
    }
    if (getenv("DDX_DEBUG_INLINE")) fprintf(stderr, "ddx: a bounded in-kernel wait ran out (flags %d): the tile pass becomes its own launch, the run is repeated\n", flags);
    if (flags & ENGINE_FLAG_INLINE_TIMEOUT)
        if (int err = disable_inline(e, s)) return err;
    // (the flag is sticky: it may stem from the last run or from an evaluation behind it.  Repeating the run is right in both cases --
    // an evaluation changes nothing the run's result depends on -- and the caller repeats its evaluation when this returns 1)
    if (e->last.kind == 1) {
        if (int err = restore_run_start(e, e->last.it0, s)) return err;
        if (int err = engine_run_impl(e, e->last.it0, e->last.n, e->last.use_graph, stream, e->last.sel_out, e->last.sel_lo)) return err;
        DDX_HIP(hipStreamSynchronize(s));
    }
    return 1;
}

// ---------------------------------------------------------------------------------------------
// get_argmin / get_pose (diffdope.py:1488-1513,1618-1632) for the local hypotheses, on the device: mean over the used
// loss rows of iteration `it`, arg-min (ties -> lowest index), and the winner's row for the [world,18] exchange table:
// (loss, global index, 4x4 pose).  One workgroup; replaces ~15 tiny framework launches and 3 host syncs.
__global__ __launch_bounds__(256) void select_best_kernel(const float* __restrict__ loss_rows, int row_mask, int B,
                                                          const float* __restrict__ mtx, int lo, float* __restrict__ out18)
{
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const int tid = threadIdx.x;
    const int n_used = __popc(row_mask & 15);
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int b = tid; b < B; b += 256) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((row_mask >> r) & 1) a += loss_rows[(size_t)r * B + b];
        a = __fdiv_rn(a, (float)max(n_used, 1));
        if (a < best || (a == best && b < bi)) { best = a; bi = b; }
    }
    s_v[tid] = best;
    s_i[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float v = s_v[tid + o];
            const int i = s_i[tid + o];
            if (v < s_v[tid] || (v == s_v[tid] && i < s_i[tid])) { s_v[tid] = v; s_i[tid] = i; }
        }
        __syncthreads();
    }
    const int w = s_i[0] < B ? s_i[0] : 0;  // (all-NaN losses: index 0, as torch.argmin would not help either)
    if (tid == 0) { out18[0] = s_v[0]; out18[1] = (float)(w + lo); }
    if (tid < 16) out18[2 + tid] = mtx[(size_t)w * 16 + tid];
}

extern "C" int ddx_select_best(const float* loss_rows, int row_mask, int B, const float* mtx, int lo, float* out18, void* stream)
{
    DDX_REQUIRE(loss_rows && mtx && out18, DDX_E_NULL, "select_best: NULL pointer");
    DDX_REQUIRE(B >= 1 && (row_mask & 15) != 0, DDX_E_SHAPE, "select_best: B=%d row_mask=%d", B, row_mask);
    select_best_kernel<<<1, 256, 0, (hipStream_t)stream>>>(loss_rows, row_mask, B, mtx, lo, out18);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_engine_eval(ddx_engine* e, int it, float* grad_out, float* loss_out, void* stream)
{
    DDX_REQUIRE(e && grad_out, DDX_E_NULL, "engine_eval: NULL pointer");
    DDX_REQUIRE(it >= 0 && it < e->dev.d.max_iters, DDX_E_SHAPE, "engine_eval: iteration %d outside [0,%d)", it, e->dev.d.max_iters);
    e->fwd_cached_it = -1;
    hipStream_t s = (hipStream_t)stream;
    if (!e->setup_done)
        if (int err = engine_setup(e, s)) return err;
    if (int err = run_prologue(e, it, s)) return err;
    // (an evaluation always takes the tile pass as its own launch: nothing in it can time out, so ddx_engine_run_check only ever has
    // runs to repeat; same bits either way)
    const int keep_inline = e->dev.big_inline;
    e->dev.big_inline = 0;
    int err = launch_step(e, STEP_EVAL, it, s);
    if (!err) err = launch_rest(e, it, s, nullptr);
    e->dev.big_inline = keep_inline;
    e->dev.eval_grad = grad_out;  // (finish_kernel in evaluation mode: hands out gradient and losses, steps nothing)
    e->dev.eval_loss = loss_out;
    if (!err) err = launch_finish(e, it + 1, s);
    e->dev.eval_grad = nullptr;
    e->dev.eval_loss = nullptr;
    return err;
}

// SURVEY 8(b2) names for the fused pass: forward (losses) and backward (d loss / d params).  Forward and analytic backward
// are ONE pass in this engine (the pixel's gradient is known as soon as it is shaded), so both run ddx_engine_eval.
extern "C" int ddx_render_loss_fwd(ddx_engine* e, int it, float* loss_out, void* stream)
{
    DDX_REQUIRE(e && loss_out, DDX_E_NULL, "render_loss_fwd: NULL pointer");
    const int err = ddx_engine_eval(e, it, e->dev.eval_tmp, loss_out, stream);
    if (!err) e->fwd_cached_it = it;  // the gradient of this very pass is kept for the backward call of the pair
    return err;
}

extern "C" int ddx_render_loss_bwd(ddx_engine* e, int it, float* grad_out, void* stream)
{
    DDX_REQUIRE(e && grad_out, DDX_E_NULL, "render_loss_bwd: NULL pointer");
    if (e->fwd_cached_it == it) {  // forward of the same pair ran last: its pass already produced the gradient
        e->fwd_cached_it = -1;
        DDX_HIP(hipMemcpyAsync(grad_out, e->dev.eval_tmp, (size_t)7 * e->dev.d.B * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return 0;
    }
    return ddx_engine_eval(e, it, grad_out, nullptr, stream);
}

// stand-alone optimiser steps over n floats (the engine's own loop has them fused into step_kernel)
__global__ void sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g, float lr, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = p[i] - lr * g[i];
}

__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 float lr, float b1, float b2, float eps, int step, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // (the expressions of update_head, so a caller stepping with this gets the engine's trajectory bit for bit)
    const float c1 = 1.f - exp2f((float)step * log2f(b1)), c2 = 1.f - exp2f((float)step * log2f(b2));
    const float gi = g[i];
    const float m1 = b1 * m[i] + (1.f - b1) * gi;
    const float m2 = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = m1; v[i] = m2;
    p[i] = p[i] - lr * (m1 / c1) / (sqrtf(m2 / c2) + eps);
}

extern "C" int ddx_sgd_step(float* params, const float* grad, float lr, int n, void* stream)
{
    DDX_REQUIRE(params && grad, DDX_E_NULL, "sgd_step: NULL pointer");
    DDX_REQUIRE(n >= 0, DDX_E_SHAPE, "sgd_step: n = %d", n);
    if (n == 0) return 0;
    sgd_step_kernel<<<ddx_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(params, grad, lr, n);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" int ddx_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                             float eps, int step, int n, void* stream)
{
    DDX_REQUIRE(params && grad && exp_avg && exp_avg_sq, DDX_E_NULL, "adam_step: NULL pointer");
    DDX_REQUIRE(n >= 0 && step >= 1, DDX_E_SHAPE, "adam_step: n = %d, step = %d (steps count from 1)", n, step);
    if (n == 0) return 0;
    adam_step_kernel<<<ddx_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(params, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, n);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" const int32_t* ddx_engine_status_ptr(ddx_engine* e) { return e ? &e->dev.st->overflow : nullptr; }

extern "C" int ddx_engine_new_observation(ddx_engine* e)
{
    DDX_REQUIRE(e, DDX_E_NULL, "engine_new_observation: NULL engine");
    e->setup_done = false;  // the next run / eval redoes the observation half of the setup (frame constants, seg list, optimiser state)
    e->balanced = false;    // ... and the next run measures the meshlets under the new poses
    e->dev.slot_table = 0;
    // a captured graph holds the kernels' by-value arguments of the OLD observation (size of the segmentation list, scatter
    // variant): it is captured again by the next run that asks for one
    if (e->exec) { (void)hipGraphExecDestroy(e->exec); e->exec = nullptr; }
    if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
    e->adam_parity = 0;
    e->fwd_cached_it = -1;
    e->last.kind = 0;
    return 0;
}

extern "C" int ddx_engine_cull_sign(ddx_engine* e) { return (e && e->setup_done) ? e->dev.cull_sign : 0; }

// the outcome of the two-stream probe (ensure_side_stream): 1 = a stream of the engine's runs beside the caller stream it met last,
// long runs go out as two half-batch chains; 0 = probed, no such stream (or switched off): one chain; -1 = never probed (the engine
// is not eligible: launches that fill the chip, tile pass as its own launch, B not a multiple of 16)
extern "C" int ddx_engine_two_chains(ddx_engine* e)
{
    return e ? e->probe_outcome : -1;
}

extern "C" int ddx_engine_profile(ddx_engine* e, int it0, int iters, float* ms_out, const char** names_out, int max_k, void* stream)
{
    DDX_REQUIRE(e && ms_out, DDX_E_NULL, "engine_profile: NULL pointer");
    DDX_REQUIRE(it0 >= 0 && iters >= 1 && it0 + iters <= e->dev.d.max_iters && max_k >= K_COUNT, DDX_E_SHAPE, "engine_profile: bad range");
    e->fwd_cached_it = -1;
    hipStream_t s = (hipStream_t)stream;
    if (!e->setup_done)
        if (int err = engine_setup(e, s)) return err;
    if (int err = run_prologue(e, it0, s)) return err;
    hipEvent_t ev[K_COUNT + 1];
    for (auto& x : ev) DDX_HIP(hipEventCreate(&x));
    for (int k = 0; k < K_COUNT; ++k) ms_out[k] = 0.f;
    // iteration it0 is drawn by the first-iteration form of step_kernel (no optimiser step in it): timed only when it is the
    // only one; the later iterations are the steady state
    int timed = 0;
    for (int i = 0; i < iters; ++i) {
        DDX_HIP(hipEventRecord(ev[K_STEP], s));
        if (int err = launch_step(e, i == 0 ? STEP_EVAL : STEP_NORMAL, it0 + i, s)) return err;
        if (int err = launch_rest(e, it0 + i, s, ev)) return err;
        DDX_HIP(hipEventSynchronize(ev[K_FINISH]));
        if (i == 0 && iters > 1) continue;
        ++timed;
        for (int k = K_STEP; k < K_FINISH; ++k) {
            float ms = 0.f;
            DDX_HIP(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
            ms_out[k] += ms;
        }
    }
    DDX_HIP(hipEventRecord(ev[K_FINISH], s));
    if (int err = launch_finish(e, it0 + iters, s)) return err;
    DDX_HIP(hipEventRecord(ev[K_COUNT], s));
    DDX_HIP(hipEventSynchronize(ev[K_COUNT]));
    e->adam_parity = (it0 + iters) & 1;
    e->last.kind = 0;  // (a profile steps the optimiser without a snapshot: nothing to repeat)
    for (int k = K_STEP; k < K_FINISH; ++k) ms_out[k] /= (float)timed;
    if (!e->dev.d.use_edge) ms_out[K_EDGE] = 0.f;  // (not launched)
    if (e->dev.big_inline) ms_out[K_BIG] = 0.f;    // (not launched: the tile pass rides in the shading launch; the interval is two event records)
    DDX_HIP(hipEventElapsedTime(&ms_out[K_FINISH], ev[K_FINISH], ev[K_COUNT]));  // once per run, not per iteration
    for (int k = 0; k < K_COUNT; ++k)
        if (names_out) names_out[k] = kKernelNames[k];
    for (auto& x : ev) (void)hipEventDestroy(x);
    return K_COUNT;
}

// ---------------------------------------------------------------------------------------------
// Engine groups (see GroupHdr): the members advance in lock step, one launch of each kernel per iteration for all of them.
struct ddx_engine_group {
    std::vector<ddx_engine*> members;
    EngineDev* d_tab = nullptr;        // device table of the members' EngineDev
    std::vector<EngineDev> h_tab;
    std::vector<unsigned> gen_up;      // set-up generation of each member when its row was uploaded (0 = never)
    bool uploaded = false;
    bool big_inline = false;  // decided with the table upload
    int last_it0 = 0, last_n = 0;  // the last run, for ddx_engine_group_run_check (last_n = 0: none, or seen clean)
};

extern "C" int ddx_engine_group_create(ddx_engine** engines, int n, ddx_engine_group** out)
{
    DDX_REQUIRE(engines && out, DDX_E_NULL, "engine_group_create: NULL pointer");
    DDX_REQUIRE(n >= 1 && n <= GROUP_MAX, DDX_E_SHAPE, "engine_group_create: %d members (1..%d)", n, GROUP_MAX);
    for (int i = 0; i < n; ++i) {
        DDX_REQUIRE(engines[i], DDX_E_NULL, "engine_group_create: member %d is NULL", i);
        DDX_REQUIRE(engines[i]->dev.d.max_iters == engines[0]->dev.d.max_iters, DDX_E_SHAPE, "engine_group_create: members differ in max_iters");
        for (int j = 0; j < i; ++j) DDX_REQUIRE(engines[j] != engines[i], DDX_E_SHAPE, "engine_group_create: member %d listed twice", i);
    }
    {   // the hypotheses of all members share one grid dimension (y of the step / finish launches: at most 65535)
        long long btot = 0;
        for (int i = 0; i < n; ++i) btot += engines[i]->dev.d.B;
        DDX_REQUIRE(btot <= 65535, DDX_E_SHAPE, "engine_group_create: %lld hypotheses in all (at most 65535 per group)", btot);
    }
    ddx_engine_group* g = new (std::nothrow) ddx_engine_group();
    DDX_REQUIRE(g, DDX_E_NULL, "engine_group_create: out of host memory");
    g->members.assign(engines, engines + n);
    g->h_tab.resize((size_t)n);
    g->gen_up.assign((size_t)n, 0u);
    if (hipMalloc(&g->d_tab, (size_t)n * sizeof(EngineDev)) != hipSuccess) {
        delete g;
        DDX_REQUIRE(false, DDX_E_NULL, "engine_group_create: hipMalloc of the member table failed");
    }
    *out = g;
    return 0;
}

extern "C" void ddx_engine_group_destroy(ddx_engine_group* g)
{
    if (!g) return;
    if (g->d_tab) (void)hipFree(g->d_tab);
    delete g;
}

// the step_kernel variant of a member: members of one variant share a launch
static int step_variant(const ddx_engine* e) { return e->small_mesh ? 1 : (e->dev.scatter_mode == 3 ? 3 : (e->dev.scatter_mode == 2 ? 2 : 0)); }

static int group_step(ddx_engine_group* g, int mode, int it, hipStream_t s)
{
    const int n = (int)g->members.size();
    long long Btot = 0;
    for (auto* e : g->members) Btot += e->dev.d.B;
    for (int variant = 0; variant < 4; ++variant) {
        GroupHdr H;
        H.n = 0;
        H.bpre[0] = 0;
        int slmax = 1;
        for (int i = 0; i < n; ++i) {
            ddx_engine* e = g->members[i];
            if (step_variant(e) != variant) continue;
            if (e->step_resident <= 0) e->step_resident = step_capacity(e);
            // (slots as launch_step chooses them, with the hypotheses of the whole group sharing the chip)
            int SL = std::max(1, std::min(e->dev.n_meshlets, (int)(e->step_resident / Btot)));
            SL = ddx_cdiv(e->dev.n_meshlets, ddx_cdiv(e->dev.n_meshlets, SL));
            SL = std::max(SL, std::min(UPD_SLICES, std::max(1, (int)(e->step_resident / Btot))));
            H.idx[H.n] = i;
            H.sl[H.n] = SL;
            H.bpre[H.n + 1] = H.bpre[H.n] + e->dev.d.B;
            slmax = std::max(slmax, SL);
            ++H.n;
        }
        if (H.n == 0) continue;
        const dim3 grid(slmax, H.bpre[H.n]);
        if (variant == 1) step_group_kernel<1, 64, 1><<<grid, 64, 0, s>>>(g->d_tab, H, mode, it);
        else if (variant == 3) step_group_kernel<2, 256, 3><<<grid, 256, 0, s>>>(g->d_tab, H, mode, it);
        else if (variant == 2) step_group_kernel<2, 256, 2><<<grid, 256, 0, s>>>(g->d_tab, H, mode, it);
        else step_group_kernel<2, 256, 0><<<grid, 256, 0, s>>>(g->d_tab, H, mode, it);
    }
    DDX_LAUNCH_CHECK();
    return 0;
}

static GroupHdr group_all(const ddx_engine_group* g)
{
    GroupHdr H;
    H.n = (int)g->members.size();
    H.bpre[0] = 0;
    for (int i = 0; i < H.n; ++i) {
        H.idx[i] = i;
        H.sl[i] = 0;
        H.bpre[i + 1] = H.bpre[i] + g->members[(size_t)i]->dev.d.B;
    }
    return H;
}

static int group_rest(ddx_engine_group* g, int it, hipStream_t s)
{
    const GroupHdr H = group_all(g);
    int smax = 1, semax = 0;
    bool edge = false;
    for (auto* e : g->members) {
        smax = std::max(smax, e->dev.s_shade);
        if (e->dev.d.use_edge) { edge = true; semax = std::max(semax, e->dev.s_edge); }
    }
    if (!g->big_inline) big_pass_group_kernel<<<dim3(BIG_GRID, H.n), BIG_WAVES * 64, 0, s>>>(g->d_tab, H, it);
    const dim3 gs(H.bpre[H.n], smax, g->big_inline ? 3 : 2);
    if (edge) shade_group_kernel<true><<<gs, 256, 0, s>>>(g->d_tab, H, it);
    else shade_group_kernel<false><<<gs, 256, 0, s>>>(g->d_tab, H, it);
    if (edge) edge_group_kernel<<<dim3(H.bpre[H.n], semax), 256, 0, s>>>(g->d_tab, H, it);
    DDX_LAUNCH_CHECK();
    return 0;
}

// Iterations [it0, it0 + n) of every member (rows of each member's own lr_sched / loss_log / mtx_log); asynchronous on `stream`.
// The result of a member is bit for bit what ddx_engine_run gives it alone.
extern "C" int ddx_engine_group_run(ddx_engine_group* g, int it0, int n, void* stream)
{
    DDX_REQUIRE(g, DDX_E_NULL, "engine_group_run: NULL group");
    hipStream_t s = (hipStream_t)stream;
    const int max_iters = g->members[0]->dev.d.max_iters;
    DDX_REQUIRE(it0 >= 0 && n >= 0 && it0 + n <= max_iters, DDX_E_SHAPE, "engine_group_run: iterations [%d,%d) exceed max_iters=%d", it0, it0 + n, max_iters);
    bool fresh = !g->uploaded;
    for (size_t i = 0; i < g->members.size(); ++i) {
        ddx_engine* e = g->members[i];
        e->fwd_cached_it = -1;
        if (!e->setup_done)
            if (int err = engine_setup(e, s)) return err;
        // a member whose set-up ran since its row was uploaded -- here, or in a run / evaluation / profile of its own after a
        // new observation -- has a stale row (size of the seg list, scatter variant, order of the roles, culling sign, box)
        fresh = fresh || e->setup_gen != g->gen_up[i];
    }
    if (n == 0) return 0;
    RoctxRange rr("ddx_engine_group_run");
    for (auto* e : g->members)
        if (int err = run_prologue(e, it0, s)) return err;
    if (fresh) {  // (set-up fills fields of EngineDev: meshlet count, culling sign, scatter variant, bounding box, seg list size)
        int smax = 1, btot = 0;
        bool edge = false, all_inline = true;
        for (auto* e : g->members) {
            smax = std::max(smax, e->dev.s_shade);
            btot += e->dev.d.B;
            edge = edge || e->dev.d.use_edge;
            all_inline = all_inline && e->dev.big_inline;
        }
        g->big_inline = all_inline;  // (every member a multiple of 8 hypotheses: so is every prefix, and hypothesis x of the launch sits on XCD x % 8)
        (void)smax; (void)btot; (void)edge;
        if (getenv("DDX_DEBUG_INLINE")) fprintf(stderr, "ddx group: inline tile pass %d\n", (int)g->big_inline);
        for (size_t i = 0; i < g->members.size(); ++i) {
            g->h_tab[i] = g->members[i]->dev;
            g->h_tab[i].eval_grad = nullptr;
            g->h_tab[i].eval_loss = nullptr;
            g->h_tab[i].sel_out = nullptr;
            g->h_tab[i].big_inline = g->big_inline ? 1 : 0;
            g->h_tab[i].slot_table = 0;  // (a group's grid has its own dispatch order: equal shares)
            g->h_tab[i].mcost_rec = 0;
            g->gen_up[i] = g->members[i]->setup_gen;
        }
        DDX_HIP(hipMemcpyAsync(g->d_tab, g->h_tab.data(), g->h_tab.size() * sizeof(EngineDev), hipMemcpyHostToDevice, s));
        DDX_HIP(hipStreamSynchronize(s));  // (pageable source)
        g->uploaded = true;
    }
    if (int err = group_step(g, STEP_FIRST, it0, s)) return err;
    if (int err = group_rest(g, it0, s)) return err;
    for (int i = 1; i < n; ++i) {
        if (int err = group_step(g, STEP_NORMAL, it0 + i, s)) return err;
        if (int err = group_rest(g, it0 + i, s)) return err;
    }
    {
        const GroupHdr H = group_all(g);
        finish_group_kernel<<<dim3(UPD_SLICES, H.bpre[H.n]), 256, 0, s>>>(g->d_tab, H, it0 + n);
        DDX_LAUNCH_CHECK();
    }
    for (auto* e : g->members) { e->adam_parity = (it0 + n) & 1; e->last.kind = 0; }
    g->last_it0 = it0;
    g->last_n = n;
    return 0;
}

// ddx_engine_run_check for a group: any member's flag repeats the group's last run with the tile pass of EVERY member in its own launch
extern "C" int ddx_engine_group_run_check(ddx_engine_group* g, void* stream)
{
    DDX_REQUIRE(g, DDX_E_NULL, "engine_group_run_check: NULL group");
    hipStream_t s = (hipStream_t)stream;
    bool any = false;
    for (auto* e : g->members) {
        if (!e->setup_done) continue;
        int flags = 0;
        if (int err = read_flags(e, s, &flags)) return err;
        any = any || (flags & ENGINE_FLAG_INLINE_TIMEOUT);
    }
    if (!any) { g->last_n = 0; return 0; }
    for (auto* e : g->members) {
        if (int err = disable_inline(e, s)) return err;
        if (g->last_n > 0)
            if (int err = restore_run_start(e, g->last_it0, s)) return err;
    }
    g->uploaded = false;
    if (g->last_n > 0) {
        if (int err = ddx_engine_group_run(g, g->last_it0, g->last_n, stream)) return err;
        DDX_HIP(hipStreamSynchronize(s));
    }
    return 1;
}

// a member's observation changed (ddx_engine_new_observation): its table row is uploaded again by the next run
extern "C" int ddx_engine_group_invalidate(ddx_engine_group* g)
{
    DDX_REQUIRE(g, DDX_E_NULL, "engine_group_invalidate: NULL group");
    g->uploaded = false;
    return 0;
}

// DDX_TRACE=1 builds of the engine object: copies the stamp buffer to the host ([3][TRACE_WG][8] uint64); returns the number of
// uint64 written, 0 when tracing is off (measurement tool, tools/trace_kernels.py)
extern "C" int ddx_engine_trace_read(ddx_engine* e, unsigned long long* out, int max_n)
{
    if (!e || !e->dev.trace || !out) return 0;
    const int n = std::min(max_n, 3 * TRACE_WG * 8);
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    if (hipMemcpy(out, e->dev.trace, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

extern "C" void ddx_engine_destroy(ddx_engine* e)
{
    if (!e) return;
    if (e->dev.trace) (void)hipFree(e->dev.trace);
    if (e->exec) (void)hipGraphExecDestroy(e->exec);
    if (e->graph) (void)hipGraphDestroy(e->graph);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    delete e;
}

#ifdef DDX_EXPERIMENTS  // variant builds only (tools/build_variant.py): measurement code that is not part of the product
#include "tools/experiments/stagger_exp.inc"
#include "tools/experiments/role_kernels.inc"
#endif
