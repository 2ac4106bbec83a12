/*
 * ddx.h -- C ABI of libddx.so, the MI355X (gfx950) native library behind diffdope_amd.
 *
 * This is the drop-in boundary for the render-and-compare hot path of NVlabs/diff-dope
 * (SURVEY.md section 8b).  It replaces
 *   - the reference's own plugin `renderutils_plugin` (diffdope/c_src/torch_bindings.cpp:279-284,
 *     loaded by diffdope/ops.py:83-96):  xfm_fwd / xfm_bwd / xfm_bwd_full / xfm_bwd_mtx;
 *   - the nvdiffrast entry points the reference calls (diffdope/diffdope.py:147,198,214,221,1312):
 *     rasterize / interpolate / texture(linear) / antialias, forward and backward;
 *   - the body of DiffDope.run_optimization (diffdope/diffdope.py:1656-1714) as one fused engine.
 *
 * Conventions
 *   - plain C: raw device pointers + sizes, no torch/ATen types anywhere;
 *   - the CALLER allocates every buffer (outputs, gradients, scratch); the library never owns
 *     tensor memory.  Scratch sizes are queried with the *_scratch_bytes functions;
 *   - every call is asynchronous on the hipStream_t passed as `stream` (a void*; pass
 *     torch.cuda.current_stream().cuda_stream) and is hipGraph-capturable;
 *   - return value: 0 = ok, negative = bad argument (DDX_E_*), positive = hipError_t;
 *     ddx_last_error() returns a thread-local message for the last failing call;
 *   - all floating point is fp32, indices int32, tensors dense row-major unless a batch stride
 *     (in elements, 0 = broadcast one copy over the batch) is given.
 */
#ifndef DDX_H
#define DDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDX_VERSION 100

#define DDX_E_NULL (-1)     /* a required pointer is NULL */
#define DDX_E_SHAPE (-2)    /* a size is out of range */
#define DDX_E_SCRATCH (-3)  /* scratch buffer too small */
#define DDX_E_ALIGN (-4)    /* pointer not 16-byte aligned */

int ddx_version(void);
const char* ddx_last_error(void);

/* Compatibility switches for the arithmetic this build restates from nvdiffrast's published behaviour (DESIGN.md section 2,
 * "deviations"): process-wide for the op-level entry points (read at call time), per engine through ddx_engine_desc.compat.
 * ddx_set_compat returns the previous flags.
 *   DDX_COMPAT_UNCLAMPED_BARY_GRAD (deviation D2): the rasterize backward differentiates the UNCLAMPED barycentric expression,
 *   as nvdiffrast does, instead of the forward's true derivative (a component saturated by the [0,1] clamp passes nothing). */
#define DDX_COMPAT_UNCLAMPED_BARY_GRAD 1
int ddx_set_compat(int flags);

/* ---------------------------------------------------------------------------------------------
 * xfm: batched 4x4 transform of points / vectors.
 * Replaces torch_bindings.cpp:142-175 (xfm_fwd), :177-203 (xfm_bwd), :242-277 (xfm_bwd_mtx),
 * :205-239 (xfm_bwd_full) and the kernels of c_src/mesh.cu:22-214.
 *   points  [Bp,N,3] with batch stride points_bstride (0 => broadcast, Bp==1)
 *   matrix  [B,4,4]
 *   out     [B,N,4] (is_points) or [B,N,3] (vectors):  out[b,n,r] = sum_c M[b,r,c] p[n,c] (+ M[b,r,3])
 *   dout    same shape as out
 *   dpoints [B,N,3] dense (the reference returns a dense [B,N,3] even for broadcast points)
 *   dmatrix [B,4,4], fully written by the call (zero rows/cols where the op has no dependence)
 * variant: bit 0: 0 = MFMA (v_mfma_f32_4x4x1_16b_f32), 1 = plain VALU fma (for A/B measurements); bit 1 (value 2), d_matrix
 * kernels only: deterministic reduction -- one workgroup per hypothesis in a fixed order instead of one per 4096 points
 * joined by fp32 atomicAdd (the default, whose sum order over workgroups depends on scheduling).
 * ------------------------------------------------------------------------------------------- */
int ddx_xfm_fwd(const float* points, long long points_bstride, const float* matrix, int B, int N,
                int is_points, float* out, int variant, void* stream);
int ddx_xfm_bwd_points(const float* matrix, int B, int N, int is_points, const float* dout,
                       float* dpoints, int variant, void* stream);
int ddx_xfm_bwd_mtx(const float* points, long long points_bstride, int B, int N, int is_points,
                    const float* dout, float* dmatrix, int variant, void* stream);
int ddx_xfm_bwd_full(const float* points, long long points_bstride, const float* matrix, int B, int N,
                     int is_points, const float* dout, float* dpoints, float* dmatrix, int variant,
                     void* stream);

/* matrix_batch_44_from_position_quat (diffdope.py:46-89) and its backward as one kernel each: q [B,4] xyzw (used as given: the
 * caller normalises, diffdope.py:1091), p [B,3] -> mtx [B,4,4] row-major; dmtx [B,4,4] -> dq [B,4], dp [B,3], fully written. */
int ddx_pose_matrix_fwd(const float* q, const float* p, int B, float* mtx, void* stream);
int ddx_pose_matrix_bwd(const float* q, const float* dmtx, int B, float* dq, float* dp, void* stream);
/* Object3D.forward's pose head (diffdope.py:1085-1098): the seven parameters, each a [B] array of its own, to quat [B,4] = (qx, qy,
 * qz, qw) / |.| and trans [B,3], one kernel each way; the backward writes d params [7,B] (rows qx qy qz qw x y z; dquat or dtrans
 * NULL = zeros). */
int ddx_pose_pack_fwd(const float* qx, const float* qy, const float* qz, const float* qw, const float* x, const float* y, const float* z,
                      int B, float* quat, float* trans, void* stream);
int ddx_pose_pack_bwd(const float* qx, const float* qy, const float* qz, const float* qw, const float* dquat, const float* dtrans, int B,
                      float* dparams, void* stream);

/* ---------------------------------------------------------------------------------------------
 * rasterize: replaces dr.rasterize(glctx, pos, tri, resolution) at diffdope.py:198-200 and its
 * backward.  Software rasteriser: per-vertex 1/256-pixel snap, per-triangle scatter with a 64-bit
 * (depth key, id) atomicMin for triangles of a few pixels, a tile pass for large ones (raster.hip).
 *   pos   [B,V,4] clip space;  tri [T,3];  rast [B,H,W,4] = (u, v, z/w, tri_id+1), 0 on background
 * H, W <= 4096.  Scratch holds no lists, so it cannot overflow: size = ddx_rasterize_scratch_bytes.
 * ------------------------------------------------------------------------------------------- */
size_t ddx_rasterize_scratch_bytes(int B, int V, int T, int H, int W);
int ddx_rasterize_fwd(const float* pos, const int32_t* tri, int B, int V, int T, int H, int W,
                      void* scratch, size_t scratch_bytes, float* rast, void* stream);
/* The same, for callers whose consumers of `rast` are this library's fused passes (the *_rows entry points below): row_range
 * [B,2] int32 (device) receives the first / last pixel row of each hypothesis' active 16x16 tiles (first > last: it draws nothing);
 * with emit_all = 0 only those rows (+ a margin of 8) of `rast` are written -- the rest of the buffer is NOT defined and must
 * not be read by anything but the *_rows passes given the same row_range.  emit_all = 1: the whole frame, as ddx_rasterize_fwd. */
int ddx_rasterize_fwd_rows(const float* pos, const int32_t* tri, int B, int V, int T, int H, int W,
                           void* scratch, size_t scratch_bytes, float* rast, int32_t* row_range, int emit_all, void* stream);
/* ... for a caller that keeps `scratch` from call to call (the iterations of a loop): zbuf_clean != 0 states that the depth buffer
 * inside it is all-empty, as every call of this library's rasteriser leaves it (the pass that reads it puts back what it finds),
 * and the clear of 8 bytes per pixel and hypothesis is left out.  Pass 0 on a scratch's first use, after a change of (B, V, T, H, W)
 * and after a call that returned an error.  Same rast and row_range either way. */
int ddx_rasterize_fwd_rows_clean(const float* pos, const int32_t* tri, int B, int V, int T, int H, int W, void* scratch,
                                 size_t scratch_bytes, float* rast, int32_t* row_range, int emit_all, int zbuf_clean, void* stream);
/* drast [B,H,W,4] (channels 0,1 used) -> dpos [B,V,4], fully written (zeroed then accumulated). */
int ddx_rasterize_bwd(const float* pos, const int32_t* tri, int B, int V, int T, int H, int W,
                      const float* rast, const float* drast, float* dpos, void* stream);

/* ---------------------------------------------------------------------------------------------
 * interpolate: replaces dr.interpolate(attr, rast, tri) (diffdope.py:143-153 and :203,:212,:218,:230).
 * The pixel-derivative outputs (diff_attrs="all") are never consumed by diff-dope and are not produced.
 *   attr [Ba,Va,A] with batch stride attr_bstride (0 => broadcast); out [B,H,W,A]
 *   dattr (nullable) same layout as attr, accumulated over pixels (and over b when broadcast),
 *   fully written; drast [B,H,W,4] fully written (channels 2,3 zero).
 * ------------------------------------------------------------------------------------------- */
int ddx_interpolate_fwd(const float* attr, long long attr_bstride, int Va, int A, const float* rast,
                        const int32_t* tri, int T, int B, int H, int W, float* out, void* stream);
int ddx_interpolate_bwd(const float* attr, long long attr_bstride, int Va, int A, const float* rast,
                        const int32_t* tri, int T, int B, int H, int W, const float* dout, float* dattr,
                        float* drast, void* stream);

/* ---------------------------------------------------------------------------------------------
 * texture: replaces dr.texture(tex, uv, filter_mode="linear") (boundary wrap), diffdope.py:221-226.
 *   tex [Bt,Th,Tw,C] with batch stride tex_bstride (0 => one shared texture -- the reference
 *   replicates it B times, diffdope.py:875-893; this build does not need that);
 *   uv [B,H,W,2]; out [B,H,W,C]; duv [B,H,W,2] fully written; dtex nullable, same layout as tex,
 *   fully written.
 * ------------------------------------------------------------------------------------------- */
int ddx_texture_linear_fwd(const float* tex, long long tex_bstride, int Th, int Tw, int C, const float* uv,
                           int B, int H, int W, float* out, void* stream);
int ddx_texture_linear_bwd(const float* tex, long long tex_bstride, int Th, int Tw, int C, const float* uv,
                           int B, int H, int W, const float* dout, float* duv, float* dtex, int Bt,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * antialias: replaces dr.antialias(color, rast, pos, tri) at diffdope.py:214.
 * The edge topology (nvdiffrast rebuilds its hash on every call) is built ONCE per mesh on the host:
 *   ddx_topology_build(tri_host [T,3], T, opp_host [T,3]): opp[t,k] = vertex opposite to edge k
 *   (joining vertices (k+1)%3,(k+2)%3) in the lowest-indexed other triangle sharing it, or -1.
 *   color/out/dcolor [B,H,W,C]; dpos [B,V,4] fully written.
 * ------------------------------------------------------------------------------------------- */
int ddx_topology_build(const int32_t* tri_host, int T, int32_t* opp_host);
int ddx_antialias_fwd(const float* color, int C, const float* rast, const float* pos, const int32_t* tri,
                      const int32_t* opp, int B, int V, int T, int H, int W, float* out, void* stream);
int ddx_antialias_bwd(const float* color, int C, const float* rast, const float* pos, const int32_t* tri,
                      const int32_t* opp, int B, int V, int T, int H, int W, const float* dout,
                      float* dcolor, float* dpos, void* stream);

/* ---------------------------------------------------------------------------------------------
 * gbuffer: everything of render_texture_batch (diffdope.py:203-231) between dr.rasterize and dr.antialias in ONE pass each way:
 * interpolate(pos) -> xfm_points(., mtx) -> depth (:203-209); interpolate(uv) -> texture linear (or interpolate(vtx_color)) ->
 * * clamp(rast[..., 3], 0, 1) -> rgb (:218-231); interpolate(ones) -> cover, the input of antialias (:212).  One copy of the
 * mesh attributes and of the texture serves every hypothesis.
 *   rast [B,H,W,4] from ddx_rasterize_fwd; mtx [B,4,4]; pos [V,3]; tri [T,3]; uv [V,2] + tex [Th,Tw,3], or vtx_color [V,3]
 *   rgb [B,H,W,3], depth [B,H,W] (background: -mtx[2][3]), cover [B,H,W,3]
 * Backward: drgb / ddepth (either may be NULL) -> dclip [B,V,4] (x, y, w; what ddx_rasterize_bwd makes of the (u, v) gradients
 * the separate ops would hand it) and dmtx [B,4,4] (row 2), both fully written; clip [B,V,4] = the positions rasterized.
 * No gradient with respect to pos / uv / tex / vtx_color is produced (use the separate ops for that).
 * ------------------------------------------------------------------------------------------- */
int ddx_gbuffer_fwd(const float* rast, const float* mtx, const float* pos, const int32_t* tri, const float* uv, const float* tex,
                    int Th, int Tw, const float* vtx_color, int B, int V, int T, int H, int W, float* rgb, float* depth, float* cover,
                    void* stream);
int ddx_gbuffer_bwd(const float* rast, const float* clip, const float* mtx, const float* pos, const int32_t* tri, const float* uv,
                    const float* tex, int Th, int Tw, const float* vtx_color, int B, int V, int T, int H, int W, const float* drgb,
                    const float* ddepth, float* dclip, float* dmtx, void* stream);
/* ... restricted to the rows a hypothesis draws into (row_range [B,2] from ddx_rasterize_fwd_rows; NULL = the whole frame): a pixel
 * outside them is background by construction -- the forward writes rgb = cover = 0, depth = -mtx[2][3] there without reading
 * `rast`, the backward only adds its d depth to d mtx[2][3] -- so 64 hypotheses of an object that covers a sixth of the rows
 * stop reading five sixths of `rast`, d rgb and d mask.  Same outputs, bit for bit, as the unrestricted calls. */
int ddx_gbuffer_fwd_rows(const float* rast, const float* mtx, const float* pos, const int32_t* tri, const float* uv, const float* tex,
                         int Th, int Tw, const float* vtx_color, int B, int V, int T, int H, int W, const int32_t* row_range, float* rgb,
                         float* depth, float* cover, void* stream);
int ddx_gbuffer_bwd_rows(const float* rast, const float* clip, const float* mtx, const float* pos, const int32_t* tri, const float* uv,
                         const float* tex, int Th, int Tw, const float* vtx_color, int B, int V, int T, int H, int W,
                         const int32_t* row_range, const float* drgb, const float* ddepth, float* dclip, float* dmtx, void* stream);

/* The silhouette of render_texture_batch -- dr.antialias applied to the interpolation of a tensor of ones (diffdope.py:212-214) --
 * without a colour operand: the colour IS the coverage (recomputed from rast), so the forward blends IN PLACE on the `cover` image
 * ddx_gbuffer_fwd wrote (mask [B,H,W,3]: coverage on entry, antialiased silhouette on return; no copy of the frame) and the
 * backward produces dpos [B,V,4] (fully written) from dmask alone -- coverage has no gradient path, so there is no d colour.
 * Bit-identical to ddx_antialias_fwd / _bwd called with that coverage image as colour. */
int ddx_silhouette_fwd(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T, int H, int W,
                       float* mask, void* stream);
int ddx_silhouette_bwd(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T, int H, int W,
                       const float* dmask, float* dpos, void* stream);
/* ... restricted to the blocks of rows that can hold a silhouette pair (row_range as above; NULL = the whole frame) */
int ddx_silhouette_fwd_rows(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T, int H, int W,
                            const int32_t* row_range, float* mask, void* stream);
int ddx_silhouette_bwd_rows(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T, int H, int W,
                            const int32_t* row_range, const float* dmask, float* dpos, void* stream);
/* ... with the silhouette kept as ONE channel.  The three channels the reference carries (interpolate of a [T,3] tensor of ones,
 * diffdope.py:212) are one number per pixel; at 64 x 640x480 the two extra copies are 157 MB written by the g-buffer pass and read and
 * written again by every consumer.  `channels` = 1 or 3 (3: the functions above).  ddx_gbuffer_fwd_rows_c: cover [B,H,W,cover_channels];
 * rgb and depth may each be NULL there (an output nobody reads: without rgb neither the texture nor the vertex colours are read).  ddx_silhouette_bwd_rows_c
 * with one channel takes d mask [B,H,W,1] = the sum of the three channel gradients. */
int ddx_gbuffer_fwd_rows_c(const float* rast, const float* mtx, const float* pos, const int32_t* tri, const float* uv, const float* tex,
                           int Th, int Tw, const float* vtx_color, int B, int V, int T, int H, int W, const int32_t* row_range, float* rgb,
                           float* depth, float* cover, int cover_channels, void* stream);
int ddx_silhouette_fwd_rows_c(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T, int H, int W,
                              const int32_t* row_range, float* mask, int channels, void* stream);
int ddx_silhouette_bwd_rows_c(const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B, int V, int T, int H, int W,
                              const int32_t* row_range, const float* dmask, int channels, float* dpos, void* stream);

/* Image-space part of the built-in losses for the op-by-op path (diffdope.py:547-613: l1_rgb_with_mask :547-562,
 * l1_depth_with_mask :565-580, l1_mask :583-613): out[b] = mean_i |(x[b,i] - y[i]) * m[i * m_stride]|, with the observed image y [N]
 * and mask m shared by the B hypotheses (m NULL = no mask; m_stride 3 reads channel 0 of a [H,W,3] mask for a [H,W] depth image).
 * partial: caller scratch [B,128] floats.  Backward: dx[b,i] = sign(.) * m * gout[b] / N.  Fixed reduction order. */
int ddx_masked_l1_fwd(const float* x, const float* y, const float* m, int m_stride, int B, long long N, float* partial, float* out,
                      void* stream);
int ddx_masked_l1_bwd(const float* x, const float* y, const float* m, int m_stride, const float* gout, int B, long long N, float* dx,
                      void* stream);
/* ... for an x of ONE channel against an observed image and mask of three (l1_mask, diffdope.py:583-613, on the single copy of the
 * silhouette that ddx_gbuffer_fwd_rows_c / ddx_silhouette_*_rows_c keep): x [B,P], y and m [P,3] (m NULL = no mask);
 * out[b] = mean over the 3 P terms |(x[b,i] - y[i,c]) * m[i,c]|; dx[b,i] = sum over c of sign(.) * m[i,c] * gout[b] / (3 P). */
int ddx_masked_l1_bc3_fwd(const float* x, const float* y, const float* m, int B, long long P, float* partial, float* out, void* stream);
int ddx_masked_l1_bc3_bwd(const float* x, const float* y, const float* m, const float* gout, int B, long long P, float* dx, void* stream);
/* ... and the batch-weighted sum of the rows in the same launches: what a built-in loss returns, (out * learning_rates).mean() * weight
 * (diffdope.py:534-544 dist_batch_lr, :562, :580, :613), is sum_b out[b] * bw[b] with bw = learning_rates * weight / B.  bc3 = 0: the
 * operands of ddx_masked_l1_fwd; bc3 != 0: those of ddx_masked_l1_bc3_fwd (N = P).  out [B]: the same bits as those calls;
 * sum_out [1].  Backward: gout [B] (d out) or NULL, gsum [1] (d sum) or NULL -- not both NULL --, one launch. */
int ddx_masked_l1_fwd_sum(const float* x, const float* y, const float* m, int m_stride, int bc3, int B, long long N, const float* bw,
                          float* partial, float* out, float* sum_out, void* stream);
int ddx_masked_l1_bwd_sum(const float* x, const float* y, const float* m, int m_stride, int bc3, const float* gout, const float* gsum,
                          const float* bw, int B, long long N, float* dx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused refinement engine: the body of DiffDope.run_optimization (diffdope.py:1656-1714) for the
 * built-in losses (l1_rgb_with_mask / l1_depth_with_mask / l1_mask, diffdope.py:547-613):
 * pose -> matrices -> vertex transform -> tile binning/raster -> shade + loss + analytic backward
 * -> d loss / d(q,t) -> optimiser step, B hypotheses at a time, no G-buffer ever written to HBM.
 * use_edge / w_edge add this build's EXTENSION term (the reference has no edge loss): L1 between the
 * 3x3 Sobel/8 gradients (zero padding) of the luminance (r+g+b)/3 of the rendered colour image and of
 * gt_rgb * gt_seg, mean over pixels and the two components (oracle/ddx_oracle.c:orc_loss_edge).
 * ------------------------------------------------------------------------------------------- */
typedef struct ddx_engine_desc {
    int32_t B;        /* hypotheses on this device */
    int32_t B_global; /* hypotheses of the whole job (the batch mean of diffdope.py:562 divides by this) */
    int32_t V, T, H, W;
    int32_t Th, Tw;   /* texture size; Th == 0 => per-vertex colour path (diffdope.py:230) */
    int32_t use_rgb, use_depth, use_mask;
    float w_rgb, w_depth, w_mask;
    int32_t optimizer; /* 0 = SGD (reference, diffdope.py:1363), 1 = Adam */
    float adam_beta1, adam_beta2, adam_eps;
    int32_t max_iters; /* rows available in lr_sched / loss_log / mtx_log */
    int32_t use_edge;  /* EXTENSION (no reference counterpart): Sobel-gradient L1 term, needs gt_rgb and a colour source */
    float w_edge;
    /* Slices (workgroups) per hypothesis of the shading / edge launches; 0 = chosen from B (512 / B and 1792 / B, at most
     * 64).  Each slice contributes one partial row per wave to the fixed-order sum of a hypothesis' gradient, so the last
     * bits of that sum depend on the slice count.  A job sharded over GPUs (B < B_global) that must reproduce the unsharded
     * run BIT FOR BIT passes the unsharded run's counts here (tests/test_gpu_engine.py::test_engine_shard_invariance);
     * with 0 the shards agree with the unsharded run to fp32 rounding of that sum. */
    int32_t shade_slices, edge_slices;
    /* != 0: draw both faces of every triangle, always -- dr.rasterize's rule (diffdope/diffdope.py:198-200), what the Python layer
     * passes unless asked for cull_backfaces=True, and what bench.py and the full-size parity tests run.  A C caller that wants the
     * reference's results sets this field to 1: a zero-initialised descriptor asks for deviation D5.  0: when the mesh is a closed, consistently oriented surface
     * (checked once, vertices welded by position) the rasteriser skips back-facing triangles of every hypothesis that lies
     * entirely inside the view volume -- they are hidden behind front faces there, so this changes nothing in exact
     * arithmetic (deviation D5) and halves the fragment work.  Environment DDX_NO_CULL=1 also disables. */
    int32_t no_backface_cull;
    int32_t compat;   /* DDX_COMPAT_* bits for this engine (0 = this build's documented behaviour) */
    /* Where the tile pass for LARGE / near-clipped triangles runs.  0 (default): the set-up estimates the longest edge of the mesh in
     * pixels from the observed object's size; where no large triangle is to be expected (dense meshes) and B is a multiple of 8,
     * the launch between the rasterising and the shading kernel is dropped and worker workgroups in the first slab of the shading
     * launch run the pass for a hypothesis that has such triangles after all.  The shading workgroups of such a hypothesis wait
     * for its workers behind agent-scope release / acquire fences and for a BOUNDED time (20 ms; DDX_BIG_WAIT_US): nothing is
     * assumed about dispatch order or workgroup placement.  A wait that runs out sets bit 0 of status word 7, the run still
     * terminates, its numbers are void, and ddx_engine_run_check repeats it with the separate launch.
     * 1: always the separate launch.  Environment DDX_BIG_INLINE=0 / 1 overrides the estimate.  Same results either way, bit
     * for bit. */
    int32_t separate_big_pass;
    /* 0 (default): ddx_engine_run / ddx_engine_run_select may issue a run of 16 or more iterations (48 or more for larger step
     * launches; never above 7 000 meshlet-hypothesis pairs, where one launch fills the chip; no graph replay, no capture in
     * progress, tile pass inside the shading launch, B a multiple of 16) as two chains of half-batch launches: one on the
     * caller's stream, one on a second stream of the library, forked from the caller's stream after the first iteration and joined to
     * it before the run's last kernel -- so everything the call enqueues is still ordered on `stream` as far as the caller can
     * see, and the results are the same bit for bit.  The second stream comes from a process-wide registry keyed by (device,
     * caller stream): the first engine that needs one for a caller stream checks with two 30-us kernels that a candidate really
     * runs beside it (streams that share a hardware queue take turns), tries up to six, and records the answer -- one chain if
     * none can be had -- for every later engine on that stream (synchronises the caller's stream, once per process and stream).  1: one chain on the
     * caller's stream only.  Environment DDX_TWO_STREAMS=0
     * has the same effect for every engine of the process; DDX_TWO_MIN=n forks every eligible run of n or more iterations. */
    int32_t single_stream;
} ddx_engine_desc;

typedef struct ddx_engine_buffers {
    /* mesh (one copy, shared by all hypotheses) */
    const float* pos;        /* [V,3] */
    const int32_t* tri;      /* [T,3] */
    const int32_t* opp;      /* [T,3] from ddx_topology_build */
    const float* uv;         /* [V,2] or NULL */
    const float* tex;        /* [Th,Tw,3] or NULL */
    const float* vtx_color;  /* [V,3] or NULL */
    const float* proj;       /* [4,4] */
    /* observed images (one copy), stored bottom-up like the reference holds them (diffdope.py:1131) */
    const float* gt_rgb;     /* [H,W,3] or NULL */
    const float* gt_depth;   /* [H,W]   or NULL */
    const float* gt_seg;     /* [H,W,3] */
    const float* lr_mult;    /* [B] per-hypothesis loss multipliers (diffdope.py:1368-1375) */
    const float* lr_sched;   /* [max_iters] optimiser lr per iteration (diffdope.py:1657-1664) */
    float* params;           /* [7,B] qx,qy,qz,qw,x,y,z -- read and updated in place */
    float* loss_log;         /* [max_iters,4,B] weighted, un-LR'd per-hypothesis losses (rgb,depth,mask,edge) */
    float* mtx_log;          /* [max_iters,B,16] pose matrix used by each iteration's forward */
    void* scratch;
    size_t scratch_bytes;
} ddx_engine_buffers;

typedef struct ddx_engine ddx_engine;

/* The first run of an engine performs its one-time setup on the run's stream (frame constants of the observed images, the
 * internal spatially sorted copy of the mesh, per-triangle records) and reads the number of observed-mask pixels back to
 * choose the rasteriser's fragment variant (expected covered pixel centres per triangle); the result of the optimisation
 * does not depend on that choice (bit-identical, tests/test_gpu_engine.py).  Environment: DDX_SCATTER_EXCHANGE=0|1
 * overrides the choice (tuning only). */
size_t ddx_engine_scratch_bytes(const ddx_engine_desc* desc);
int ddx_engine_create(const ddx_engine_desc* desc, const ddx_engine_buffers* bufs, ddx_engine** out);
/* Run iterations [it0, it0+n) (rows of lr_sched / loss_log / mtx_log).  use_graph = k > 0 replays a captured hipGraph of
 * k iterations (k <= 64, fixed at the first graph run of the engine; a remainder of fewer than k iterations is launched
 * kernel by kernel).  Measured on MI355X: k = 1 is 9 % slower than plain stream launches, k = 20 equal (+-1 %).
 * Asynchronous, with ONE exception: the first run / eval / profile after ddx_engine_create or ddx_engine_new_observation does the
 * set-up, which reads small tables back (frame constants, segmentation list; on meshes with four or more meshlets per workgroup
 * also the per-meshlet times of its first launch, once) and therefore synchronises `stream`; that call must not be made while
 * `stream` is being captured (the meshlet calibration is skipped inside a capture rather than breaking it). */
int ddx_engine_run(ddx_engine* e, int it0, int n, int use_graph, void* stream);
/* ddx_engine_run (n >= 1) with the selection of the best local hypothesis folded into the run's LAST kernel -- get_argmin / get_pose
 * (diffdope/diffdope.py:1488-1513, 1618-1632) for iteration it0 + n - 1 without a further launch: out18 (18 floats; device memory
 * or mapped pinned host memory) receives what ddx_select_best writes for that iteration's rows -- (mean over the engine's enabled
 * loss terms of the winner's weighted losses, lo + its local index, its 4x4 pose row-major), ties to the lowest index, a NaN
 * loss never wins.  The row is written by the run's last kernel, out18[0] LAST and behind the rest of the row: with out18 in mapped
 * pinned host memory the end of a run costs one kernel and one synchronisation, and a caller may instead poll out18[0] (set to a
 * sentinel before the call: a loss is never negative) -- the re-arm work of that last kernel may then still be in flight, so the
 * NEXT call on the engine still orders itself behind it through `stream`. */
int ddx_engine_run_select(ddx_engine* e, int it0, int n, int use_graph, int lo, float* out18, void* stream);
/* The host half of every bounded in-kernel wait (at present: the tile pass inside the shading launch, separate_big_pass above).
 * Synchronises `stream` and reads status word 7 (bit 0: the in-launch tile pass).  0: the work enqueued so far is valid.  1: a wait of the last ddx_engine_run /
 * ddx_engine_run_select ran out -- the engine has been switched to the separate tile-pass launch for good, parameters and
 * optimiser state have been put back to what that run started from (a snapshot its first kernel takes) and the run has been
 * repeated, same iterations, same out18, and has finished: its results are valid now, bit for bit those of an engine created
 * with separate_big_pass = 1.  An evaluation pass (ddx_engine_eval / ddx_render_loss_*) enqueued since must be repeated by the
 * caller.  Call it wherever the results of a run are consumed and before the next run is enqueued (a second run behind a void
 * one starts from void parameters); the Python RefineEngine does both.  ddx_engine_run_select also signals the condition in
 * band: out18[0] is NaN (never otherwise) when the run was void.  Negative / positive > 1: error codes as everywhere. */
int ddx_engine_run_check(ddx_engine* e, void* stream);
/* Evaluation pass without an optimiser step (for callers that bring their own optimiser): renders the hypotheses
 * at the CURRENT contents of `params`, writes d loss / d params to grad_out [7,B] and the weighted, un-LR'd
 * per-hypothesis losses (rgb, depth, mask, edge) to loss_out [4,B] (may be NULL); parameters, optimiser state and logs
 * are left untouched.  loss = sum_k sum_b lr_mult[b] * loss_out[k,b] / B_global  (diffdope.py:534-613);
 * `it` only selects the mtx_log row the pose matrices are logged to. */
int ddx_engine_eval(ddx_engine* e, int it, float* grad_out, float* loss_out, void* stream);
/* The fused pass under the names of a forward / backward pair (what a torch.autograd.Function around the loss of
 * diffdope.py:1707-1711 binds): ddx_render_loss_fwd writes the weighted per-hypothesis losses [4,B], ddx_render_loss_bwd
 * d loss / d params [7,B] (the 7 nn.Parameters of diffdope.py:1019-1026).  Forward and analytic backward are ONE pass in
 * this engine: ddx_render_loss_fwd runs it (= ddx_engine_eval) and keeps the gradient, so that a ddx_render_loss_bwd for the same
 * iteration that follows it, with no other pass of the engine and no change of `params` in between, is a 7 B-float copy; a
 * ddx_render_loss_bwd without that forward runs the pass itself.  A caller that wants both at once calls ddx_engine_eval. */
int ddx_render_loss_fwd(ddx_engine* e, int it, float* loss_out, void* stream);
int ddx_render_loss_bwd(ddx_engine* e, int it, float* grad_out, void* stream);
/* Stand-alone optimiser steps over n floats (params [7,B] -> n = 7 B), for callers that run the evaluation pass and step
 * themselves: p -= lr g (torch.optim.SGD as diffdope.py:1642-1644 uses it; lr from the schedule of :1657-1664), and Adam
 * (torch.optim.Adam update, bias correction with step counted from 1; the engine's own fused update uses the same
 * expressions).  ddx_engine_run has both fused into its last kernel. */
int ddx_sgd_step(float* params, const float* grad, float lr, int n, void* stream);
int ddx_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                  int step, int n, void* stream);
/* get_argmin / get_pose (diffdope.py:1488-1513,1618-1632) for the local hypotheses, on the device:
 * loss_rows [4,B] (one row of loss_log), row_mask bit r = loss row r takes part in the mean, mtx [B,16],
 * lo = global index of the first local hypothesis; out18 = (mean loss of the winner, its global index, its 4x4
 * pose row-major): this rank's row of the [world,18] table that ONE all_reduce(SUM) exchanges. */
int ddx_select_best(const float* loss_rows, int row_mask, int B, const float* mtx, int lo, float* out18, void* stream);
/* device int32[8] inside scratch ([7]: bit 0 = a bounded wait of the in-launch tile pass ran out since the last
 * ddx_engine_run_check -- every number produced since is void until that call has returned): [0] reserved (always 0), [1] large triangles (tile-pass) of the last
 * iteration, [2] active tiles of the last iteration, [3] internal, [4] pixels with seg != 0, [5] last iteration drawn + 1,
 * [6] hypotheses of the last iteration whose object-space bounding box had a corner outside the view volume (w <= 0 or
 * |z| > w): their triangles with a vertex at w <= 0 took the near-plane clipping path and their back faces were drawn -- 0 in any
 * sane refinement */
const int32_t* ddx_engine_status_ptr(ddx_engine* e);
/* 0: the engine draws both faces (mesh not closed, projection not a pinhole, no_backface_cull, or not set up yet: the
 * decision is taken by the first run / eval); +1 / -1: triangles whose snapped screen area has this sign are culled as back
 * faces in hypotheses that lie inside the view volume. */
int ddx_engine_cull_sign(ddx_engine* e);
/* Outcome of the set-up's two-stream probe (single_stream above): 1 = long runs of this engine go out as two half-batch chains
 * (a stream of the engine's was measured to run beside the caller's); 0 = probed and refused (streams sharing a hardware queue
 * take turns: one chain); -1 = not eligible or not probed yet. */
int ddx_engine_two_chains(ddx_engine* e);
/* The same object in a new frame (tracking): the caller has overwritten the contents of gt_rgb / gt_depth / gt_seg, params, lr_mult
 * and / or lr_sched IN PLACE (same buffers, same shapes); mesh, texture and projection are unchanged.  The next run / eval redoes
 * the observation half of the set-up only (frame constants, sorted segmentation list, optimiser state, iteration 0) and keeps the
 * mesh half (sorted copies, triangle and texel records, closedness analysis): ~1 ms instead of ~4-8 ms per frame. */
int ddx_engine_new_observation(ddx_engine* e);
/* per-kernel launch durations of the last ddx_engine_profile call are returned in ms (host array
 * of `n_kernels`), measured with hipEvents on `stream`; returns the number of kernels. */
int ddx_engine_profile(ddx_engine* e, int it0, int iters, float* ms_out, const char** names_out, int max_k,
                       void* stream);
void ddx_engine_destroy(ddx_engine* e);

/* ---------------------------------------------------------------------------------------------
 * Engine groups: several engines advance in lock step with ONE launch of each kernel per iteration for all of them -- the
 * objects of one frame (examples/run_bop_scene.py:48-89 loops over them one after the other; BASELINE config 5 puts 4 objects x
 * 64 hypotheses on a GPU).  A 64-hypothesis launch is latency-bound and fills a fraction of the chip; the members of a group
 * share every grid.  Each member keeps its own buffers, scratch, schedule rows and launch geometry, and ends with bit for bit the
 * result ddx_engine_run would give it alone.  Members may differ in mesh, texture, frame size, loss set and batch size; they
 * must agree on max_iters.  The group borrows the engines (destroy the group first).  At most 32 members and 65535 hypotheses in all.
 * ------------------------------------------------------------------------------------------- */
typedef struct ddx_engine_group ddx_engine_group;
int ddx_engine_group_create(ddx_engine** engines, int n, ddx_engine_group** out);
/* iterations [it0, it0 + n) of every member; asynchronous on `stream` (plain stream launches) */
int ddx_engine_group_run(ddx_engine_group* g, int it0, int n, void* stream);
/* ddx_engine_run_check for a group: 1 = a member's wait ran out; every member now runs the separate tile-pass launch, all have
 * been put back to the start of the group's last run, and that run has been repeated and has finished */
int ddx_engine_group_run_check(ddx_engine_group* g, void* stream);
/* after ddx_engine_new_observation on a member nothing is needed, also when that member was then run, evaluated or profiled on
 * its own (every set-up bumps a generation counter the group compares); call this if a member was re-created in place */
int ddx_engine_group_invalidate(ddx_engine_group* g);
void ddx_engine_group_destroy(ddx_engine_group* g);
/* measurement hook (DDX_TRACE=1 in the environment when the engine is created): per-workgroup phase stamps, see
 * tools/trace_kernels.py; returns the number of uint64 written, 0 when tracing is off */
int ddx_engine_trace_read(ddx_engine* e, unsigned long long* out, int max_n);

#ifdef __cplusplus
}
#endif
#endif /* DDX_H */
