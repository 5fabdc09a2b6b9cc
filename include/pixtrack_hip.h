/*
 * pixtrack_hip.h -- C ABI of the MI355X (gfx950) hot path of the pixtrack tracker.
 *
 * The reference (GiantAI/pixtrack) has no C ABI: its per-frame path is Python that
 * calls three third-party engines (pixloc's optimizer and UNet in PyTorch-CUDA, and
 * instant-ngp's Testbed through pyngp).  Each entry point below replaces ONE of the
 * Python-level interfaces that path goes through; the citation names the reference
 * line where pixtrack crosses that interface.  The Python host side
 * (the pixtrack_amd package) binds these with ctypes and keeps the reference's own
 * class/method names on top (INTEGRATION.md shows the stub a maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host (small records such as
 *     poses, cameras and configuration are read on the host and passed as kernel arguments);
 *   - the caller owns every buffer; the library keeps no reference after return
 *     (work is enqueued on `stream`; buffers must stay alive until it completes);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - return value: 0 = enqueued OK; negative = PXT_E_* (argument / HIP error).
 *     Algorithmic failure of the optimiser (fewer than 10 valid points) is NOT an
 *     error: it is reported in the `failed` field of the output record, exactly like
 *     pixloc's `(T, failed)` return.
 *   - poses are 12 floats: row-major R (9) then t (3)   [pixloc Pose._data];
 *   - cameras are 10 floats: w,h,fx,fy,cx,cy,k1,k2,p1,p2 with the pixel-centre
 *     origin of pixloc Camera (cx,cy already shifted by -0.5), plus `ndist` in
 *     {0,2,4} saying how many distortion terms are live.
 *   - dense feature maps are HWC float32: map[(y*W + x)*cstride + c]; channels
 *     [0,C) are the descriptor, channel C is the confidence; cstride % 4 == 0 and
 *     cstride >= C+1.  Sparse reference features use the same record per point.
 */
#ifndef PIXTRACK_HIP_H
#define PIXTRACK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PXT_OK 0
#define PXT_E_ARG -1     /* bad argument (null pointer, size, alignment) */
#define PXT_E_HIP -2     /* a HIP runtime call failed (pxt_last_error() has text) */
#define PXT_E_TIMEOUT -3 /* in-kernel spin bound exceeded (reported via status words) */
#define PXT_E_STATE -4   /* context misuse */

#define PXT_MAX_LEVELS 8
#define PXT_LM_LOG_STRIDE 20 /* floats per logged iteration, see pxt_lm_refine */

/* Library / device info ---------------------------------------------------- */
int pxt_version(void);               /* ABI version (11), bumps on any signature change or added entry point */
const char* pxt_last_error(void);    /* text of the last PXT_E_HIP on this thread */
int pxt_device_cus(int* n_cus_host); /* multiprocessor count of the current device */

/* -------------------------------------------------------------------------
 * Levenberg-Marquardt feature-metric pose refinement.
 *
 * Replaces, in one persistent launch and with no host round trip per iteration:
 *   for level in coarse..fine:  T, failed = opt[level].run(p3d, F_ref, F_q, T,
 *                                            camera.scale(s), W_ref_query=...)
 * i.e. pixloc BaseRefiner.refine_pose_using_features -> LearnedOptimizer._run,
 * reached from pixtrack/localization/pixloc_pose_refiners.py:260-262, with
 * pixtrack's early-stop-every-iteration rule
 * (pixtrack/optimizers/pixtrack_optimizer.py:5-18) and the per-iteration
 * masked-mean cost that DebugTracker logs (pixtrack/localization/tracker.py:32-46).
 * A single-level call (n_levels = 1) is exactly one `opt.run(...)`.
 * ---------------------------------------------------------------------- */
typedef struct {
  const float* fmap;   /* query map of this level, HWC, [h][w][cstride]; descriptor
                          channels already L2-normalised over C (A.4) */
  const float* fref;   /* reference observations [n_points][cstride]; descriptor
                          L2-normalised; channel C = reference confidence */
  int32_t h, w, C, cstride;
  float cam[10];       /* query camera ALREADY scaled to this level */
  int32_t ndist;
  float lambda[6];     /* damping 10^(lo + sigmoid(const)*(hi-lo)) of this level */
} pxt_lm_level;

typedef struct {
  int32_t num_iters;        /* per-level cap (pixloc_tracker_r9.py:47 -> 150) */
  int32_t pad;              /* interpolation border (pixloc_tracker_r9.py:48 -> 1) */
  int32_t loss;             /* 0 squared, 1 huber, 2 barron(alpha) */
  float loss_alpha;         /* barron alpha (0 = Cauchy-like) */
  float loss_scale;         /* pre/post scale `a` of pixloc scaled_loss */
  float grad_stop;          /* ||g|| < grad_stop            -> stop */
  float dt_stop, dR_stop;   /* dt < dt_stop AND dR(deg) < dR_stop -> stop */
  int32_t min_valid;        /* failed |= n_valid < min_valid (10) */
  int32_t n_workgroups;     /* persistent grid size; 0 = library default */
  int32_t spin_limit;       /* polls before an inter-workgroup wait gives up with PXT_E_TIMEOUT; 0 = library default
                               (2^22, seconds).  Tests force a time-out with 1. */
  int32_t path;             /* 0 = automatic: a level whose points fit the grid's lane groups keeps every point in
                               registers for the whole level (one round); 2 = always the general several-rounds path
                               (A/B, tests: both give the same poses to fp32 reduction order) */
} pxt_lm_conf;

/* Output record (device, floats):
 *   out[0..11]  T_refined (12)        out[12] failed (0/1)
 *   out[13]     status (0 ok, PXT_E_TIMEOUT as float if a spin bound tripped)
 *   out[14]     total iterations      out[15] 1.0, stored last with system-scope release
 *               (a host polling `out` in pinned memory may read the record once it sees it)
 *   out[16 + l] iterations run at level l (execution order), l < n_levels
 * Log (device, optional, may be NULL): log[(l*num_iters + i)*PXT_LM_LOG_STRIDE + k]
 *   k=0 masked-mean cost BEFORE the update of iteration i (tracker.py:40-41)
 *   k=1 number of valid points   k=2 dR (deg) of the step   k=3 dt of the step
 *   k=4 ||g||                    k=5 1 if the solve fell back from Cholesky to LU
 *   k=8..19 pose after the update (12)
 */
int pxt_lm_refine(const float* p3d, const uint8_t* point_mask /* may be NULL */,
                  int32_t n_points, const pxt_lm_level* levels_host, int32_t n_levels,
                  const float* T_init_host /* 12 floats */, const pxt_lm_conf* conf_host,
                  float* out /* device, 16+PXT_MAX_LEVELS */, float* log /* device or NULL */,
                  void* workspace /* device, pxt_lm_workspace_bytes() */, void* stream);

/* pxt_lm_refine whose last act is the NEXT frame's camera: once the final pose stands, the kernel converts it
 * (get_camera_in_world_from_pixpose -> sfm_to_nerf_pose -> nerf_matrix_to_ngp, float64, pixtrack/utils/pose_utils.py:24,
 * pixtrack/utils/ingp_utils.py:47-63) and stores the 12 camera floats into up to two renderer camera slots
 * (pxt_ngp_camera_slot) and, optionally, a host-visible record cam_out13 (12 floats, then 1.0 stored last with
 * system-scope release; -1.0 instead, with the slots left untouched, when the refinement failed or timed out: the
 * camera record and the slots are valid only if out[12] == out[13] == 0).  A render enqueued behind this launch with pxt_ngp_render_frame(camera_from_slot = 1) then needs
 * no conversion launch of its own (pxt_ngp_render_both_from_pose's one-thread kernel).  cam_host == NULL: pxt_lm_refine. */
typedef struct {
  double conv27[27];   /* nerf2sfm centroid (3), 3 / avglen, R (4x4 row-major), totp (3), snapshot scale, offset (3) */
  float* cam_slot[2];  /* device float[12] each, or NULL */
  float* cam_out13;    /* pinned host or device float[13], or NULL */
} pxt_lm_camera;
int pxt_lm_refine_cam(const float* p3d, const uint8_t* point_mask, int32_t n_points, const pxt_lm_level* levels_host,
                      int32_t n_levels, const float* T_init_host, const pxt_lm_conf* conf_host, float* out, float* log,
                      void* workspace, const pxt_lm_camera* cam_host, void* stream);

/* K independent refinements in ONE persistent launch - K objects tracked in lock-step on one GPU (BASELINE configs[3]:
 * the eight objects of /root/reference/config/<object>.sh, one tracker per object as
 * pixtrack/pose_trackers/pixloc_tracker_r9.py:287-318 builds it; on fewer than 8 GPUs a rank carries 8 / N of them).
 * Each problem is exactly one pxt_lm_refine_cam call - own points, levels, initial pose, output record, log, workspace
 * (pxt_lm_workspace_bytes() each, all different) and optional camera record - and is solved by its own share of the
 * grid: workgroup i works on problem i mod K, so the problems never wait for each other and an iteration's
 * inter-workgroup exchange and single-lane solve are paid once per iteration for all K.  conf_host is shared;
 * conf_host->n_workgroups is the grid PER PROBLEM; 0 = the library's default: 256 / K rounded down to a multiple of 8 (at
 * most 128), raised to what one round of 8-lane point groups needs for the largest problem (ceil(n_points / 64), a multiple
 * of 8, at most 128).  Either way the grid is then capped so that all K x n_workgroups workgroups are resident at once
 * (one 8-wave workgroup per CU).  A problem's result is bit-identical to pxt_lm_refine_cam with the same EFFECTIVE grid
 * (the fixed-order folds depend on the number of workgroups only): pass an explicit n_workgroups <= resident / K to pin it.  batch_workspace: device, pxt_lm_batch_workspace_bytes(K); it holds
 * the K parameter records, copied there from pinned staging memory ahead of the launch in `stream`. */
#define PXT_LM_MAX_BATCH 16
typedef struct {
  const float* p3d;
  const uint8_t* point_mask; /* may be NULL */
  int32_t n_points;
  const pxt_lm_level* levels_host;
  int32_t n_levels;
  const float* T_init_host;    /* 12 floats */
  float* out;                  /* device or pinned host, 16 + PXT_MAX_LEVELS */
  float* log;                  /* device or pinned host, or NULL */
  void* workspace;             /* device, pxt_lm_workspace_bytes(); one per problem */
  const pxt_lm_camera* cam_host; /* or NULL */
} pxt_lm_problem;
int64_t pxt_lm_batch_workspace_bytes(int32_t n_problems);
int pxt_lm_refine_batch(const pxt_lm_problem* problems_host, int32_t n_problems, const pxt_lm_conf* conf_host,
                        void* batch_workspace, void* stream);

/* Bytes of scratch pxt_lm_refine needs (control words + tagged partial sums), independent of N.  One workspace serves
 * one launch at a time; its content carries over between launches (the tags of a launch continue above those of the
 * previous one, so nothing is zeroed per launch).  Zero it once after allocation, and again after a launch that
 * reported PXT_E_TIMEOUT in out[13] (the kernel also skips a margin of tags on that path). */
int64_t pxt_lm_workspace_bytes(void);

/* -------------------------------------------------------------------------
 * Sparse reference observations.
 *
 * Replaces PoseTrackerRefiner.interp_sparse_observations
 * (pixtrack/localization/pixloc_pose_refiners.py:327-368) plus the per-point
 * stacking/normalisation that refine_pose_using_features does afterwards (A.4):
 * transform the reference image's 3-D points with `T`, project into the reference
 * camera of each level, bilinearly sample the RAW (un-normalised) HWC map, L2-
 * normalise the C descriptor channels of the sample, keep the sampled confidence in
 * channel C.  valid[n] = 1 iff the point is visible and inside the padded image on
 * EVERY level (line :356).
 * ---------------------------------------------------------------------- */
typedef struct {
  const float* fmap;  /* raw reference map, HWC [h][w][cstride] */
  float* out;         /* [n_points][cstride] */
  int32_t h, w, C, cstride;
  float cam[10];      /* reference camera scaled by reference_scale and level scale */
  int32_t ndist;
  /* The map may be a WINDOW of the level's full map (round 5: the reference pass runs on a crop of the reference render
   * that holds every input pixel the sampled points depend on - the pyramid's dependency radius around the points'
   * bounding box - instead of on the whole render; pixloc_pose_refiners.py:282-290 computes the dense maps only to sample
   * them at the points, :236,316 `del`).  full_w > 0: the map holds columns [x0, x0 + w) and rows [y0, y0 + h) of a
   * full_w x full_h level; projection, visibility and the padded in-image test use the full level, texels are read at
   * (u - x0, v - y0) - bit for bit the samples of the full map wherever both hold the same texels.  full_w = 0: no window. */
  int32_t x0, y0, full_w, full_h;
} pxt_sample_level;

int pxt_sample_sparse(const float* p3d, int32_t n_points, const float* T_host /* 12 floats */,
                      const pxt_sample_level* levels_host, int32_t n_levels, int32_t pad,
                      int32_t normalize, uint8_t* valid /* [n_points] */, void* stream);

/* -------------------------------------------------------------------------
 * UNet feature pyramid (pixloc `UNet`, experiment pixloc_megadepth).
 *
 * Replaces `pred = self.model({"image": image_tensor})`
 * (pixtrack/localization/feature_extractor.py:48) including prepare_input's
 * HWC->CHW, /255 (:31-32).  Weights live in an opaque context created from a flat
 * host blob (layout documented in pixtrack_amd/unet.py: pack_unet_weights).
 * ---------------------------------------------------------------------- */
typedef struct pxt_unet pxt_unet;

int pxt_unet_create(const void* weights_host, int64_t n_bytes, pxt_unet** out_ctx);
int pxt_unet_destroy(pxt_unet* ctx);
/* Scratch the forward pass needs for an H x W input. */
int64_t pxt_unet_workspace_bytes(const pxt_unet* ctx, int32_t H, int32_t W);

/* Diagnostic for a NEW checkpoint (not on the frame's path): pixloc runs this network in fp32, this library stores its
 * activations as fp16.  After a single-image pxt_unet_forward on `workspace`, writes for each of the 17 convolutions
 * stats[2 * l] = largest |activation| (float) and stats[2 * l + 1] = number of non-finite values (uint32 bits) of the
 * layer's output as it sits in the workspace; -1 for the two layers that are never materialised (the first one when it is
 * computed inside the second one's staging, the last one when the fine head is fused).  A non-zero count, or a maximum
 * near 65504, means the checkpoint needs rescaling (or bf16 storage) before its poses can be trusted. */
int pxt_unet_activation_stats(pxt_unet* ctx, int32_t H, int32_t W, const void* workspace, float* stats_device /* [34] */,
                              void* stream);

/* image: HWC, 3 channels, values 0..255 (float32 if image_is_u8 == 0, else uint8).
 * mask: optional H x W uint8 (0/1) multiplied into the image first
 *       (pixloc_tracker_r9.py:224-225), NULL for none.
 * out_maps[l]: HWC float32 [h_l][w_l][cstride_l], l = 0..2 (strides 1,4,16);
 *       channel C_l is sigmoid(-uncertainty) (A.5); if normalize != 0 the C_l
 *       descriptor channels are L2-normalised per pixel (A.4, query side). */
int pxt_unet_forward(pxt_unet* ctx, const void* image, int32_t image_is_u8, const uint8_t* mask,
                     int32_t H, int32_t W, float* const out_maps[3], const int32_t out_cstride[3],
                     int32_t normalize, void* workspace, void* stream);

/* The same pass over n_images (<= PXT_UNET_MAX_BATCH) equally sized images in one set of
 * launches: a frame's reference render and its masked query
 * (pixloc_pose_refiners.py:282-290 and :255, two model() calls per frame in the
 * reference) share every weight fetch, and the small deep layers get twice the tiles.
 * images[i] / image_is_u8[i] / masks[i] (masks may be NULL, entries may be NULL) /
 * normalize[i] as in pxt_unet_forward; out_maps[3*i + l] = image i, level l.  Each image's
 * result is independent of its batch neighbours; a layer's split-K factor depends on the
 * batch size, so a batched map equals the single-image one up to fp32 summation order. */
#define PXT_UNET_MAX_BATCH 16
int64_t pxt_unet_workspace_bytes_batch(const pxt_unet* ctx, int32_t n_images, int32_t H, int32_t W);
int pxt_unet_forward_batch(pxt_unet* ctx, int32_t n_images, const void* const* images,
                           const int32_t* image_is_u8, const uint8_t* const* masks, int32_t H, int32_t W,
                           float* const* out_maps, const int32_t out_cstride[3], const int32_t* normalize,
                           void* workspace, void* stream);

/* Constant-tile skipping in encoder blocks 1-3 (on by default; $PXT_UNET_SKIP=0 turns it off for the process).  The
 * tracker's images are mostly constant - the masked query is exactly 0 outside the dilated silhouette
 * (pixloc_tracker_r9.py:224-225), the NeRF reference render exactly 0 outside the object - and a convolution tile whose
 * whole dependency cone (through every layer up to it) lies in such a region and inside the image equals ONE vector per
 * layer.  Such tiles (found conservatively from `mask` / a uint8 image on an 8 x 8 block grid) are filled with that
 * vector - obtained once per layer plan by running the layer's own kernel configuration on a constant map, hence the
 * very bits the tile would have computed: the maps are unchanged, bit for bit.  An image with neither a mask nor a
 * uint8 type is computed in full. */
int pxt_unet_set_tile_skip(pxt_unet* ctx, int32_t on);

/* How a batch of MORE than two images plans its layers.  0 (default): tile configuration and split-K factor chosen for
 * the batch as launched (large batches need no split-K: the fastest).  1: every layer takes the plan a SINGLE image of
 * the tracker's two-stream pair pass takes, whatever the batch size, so each image's maps are bit for bit the maps the
 * one-object tracker computes for it (a layer's fp32 summation order is its tile's K walk and its split-K partition).
 * Used by the lock-step multi-object tracker's parity tests; workspace sizes cover both. */
int pxt_unet_set_batch_plan(pxt_unet* ctx, int32_t per_image_plan);

/* Two images of possibly DIFFERENT sizes (H[i] x W[i]) as two single-image passes side by side on two streams - a frame's
 * reference render (reference camera x reference_scale, pixloc_pose_refiners.py:145-152) and its masked query
 * (feature_extractor.py:48 twice per frame).  Arguments as pxt_unet_forward_batch with n_images = 2; out_maps[3 i + k];
 * workspace >= pxt_unet_workspace_bytes_pair. */
int64_t pxt_unet_workspace_bytes_pair(const pxt_unet* ctx, const int32_t H[2], const int32_t W[2]);
int pxt_unet_forward_pair(pxt_unet* ctx, const void* const* images, const int32_t* image_is_u8,
                          const uint8_t* const* masks, const int32_t H[2], const int32_t W[2],
                          float* const* out_maps, const int32_t out_cstride[3], const int32_t* normalize,
                          void* workspace, void* stream);
/* Deferred join of the pair entry (round 4): with pxt_unet_set_defer_join(ctx, 1), pxt_unet_forward_pair (and the
 * two-image pxt_unet_forward_batch that runs through it) returns with image 0's maps complete in the caller's stream
 * order and image 1's pass still running on the library's side stream; the caller may enqueue work that needs image 0
 * only - the tracker's sparse sampling of the REFERENCE maps (pixloc_pose_refiners.py:327-368) - and must call
 * pxt_unet_pair_join(ctx, stream) before anything reads image 1's maps (it makes `stream` wait for that pass; a no-op
 * when nothing is pending; the next forward call on the context joins by itself). */
int pxt_unet_set_defer_join(pxt_unet* ctx, int32_t on);
int pxt_unet_pair_join(pxt_unet* ctx, void* stream);


/* One 3x3 convolution (pad 1) of the pyramid as a stand-alone call, for layer-by-layer
 * parity tests against torch.nn.functional.conv2d (SURVEY KAT-6) and for profiling:
 * in  [H][W][Cin]  fp16 NHWC (Cin % 32 == 0), weights [Cout][3][3][Cin] fp16
 * (Cout % 32 == 0), bias [Cout] float32, out [H][W][Cout] fp16; fused bias (+ReLU). */
int pxt_conv3x3_nhwc_f16(const void* in, int32_t H, int32_t W, int32_t Cin, const void* weights,
                         const float* bias, int32_t Cout, int32_t relu, void* out, void* stream);

/* The same layer with the taps already in the kernel's MFMA-fragment order (what pxt_unet_create
 * prepares once for every layer of the pyramid), for profiling the convolution alone and for
 * choosing tile configurations:
 *   pxt_conv3x3_packed_bytes   bytes of the packed taps (= Cout*9*Cin*2), < 0 on bad dims
 *   pxt_conv3x3_pack_weights   [Cout][3][3][Cin] fp16 (device) -> packed (device)
 *   pxt_conv3x3_packed         in / out as above; pool_out (optional) [H/2][W/2][Cout] receives the
 *                              2x2 max-pool of the output; cfg 0 = automatic, 1..6 = tile
 *                              configuration (csrc/pxt_unet.hip kV2Cfgs); splits > 1 = split-K over
 *                              the input-channel chunks with splitk_ws >= splits*H*W*Cout*4 bytes. */
int64_t pxt_conv3x3_packed_bytes(int32_t Cin, int32_t Cout);
int pxt_conv3x3_pack_weights(const void* weights, int32_t Cin, int32_t Cout, void* packed, void* stream);
int pxt_conv3x3_packed(const void* in, int32_t H, int32_t W, int32_t Cin, const void* packed, const float* bias,
                       int32_t Cout, int32_t relu, void* out, void* pool_out, int32_t cfg, int32_t splits,
                       void* splitk_ws, int64_t splitk_ws_bytes, void* stream);

/* -------------------------------------------------------------------------
 * instant-ngp style NeRF inference renderer.
 *
 * Replaces `testbed.render(width, height, spp, True)` and the setters pixtrack
 * drives around it (pixtrack/visualization/run_vis_on_poses.py:38-56,
 * pixtrack/utils/ingp_utils.py:22-44).
 * ---------------------------------------------------------------------- */
typedef struct pxt_ngp pxt_ngp;

typedef struct {
  int32_t n_levels, n_features, log2_hashmap, base_res;
  float per_level_scale;
  int32_t grid_cascades;   /* occupancy cascades, 128^3 each */
  float aabb_scale;        /* scene box = [0.5 - s/2, 0.5 + s/2]^3 */
  float cone_angle;        /* dt growth (1/256 when aabb_scale > 1) */
  float depth_scale;       /* colour written in Depth mode = depth * depth_scale */
  int32_t linear_colors;   /* 0: Shade applies srgb_to_linear to every finished ray's colour before the spp mean
                              (instant-ngp's shade_kernel_nerf for snapshots trained on LDR images - pixtrack's);
                              1: the snapshot was trained in linear colours (HDR set): no conversion */
} pxt_ngp_model;

int pxt_ngp_create(const pxt_ngp_model* model_host, const void* grid_params_f16_host,
                   int64_t n_grid_params, const void* mlp_params_f16_host, int64_t n_mlp_params,
                   const uint8_t* occupancy_bits_host, int64_t n_occ_bytes, pxt_ngp** out_ctx);
int pxt_ngp_destroy(pxt_ngp* ctx);

typedef struct {
  float cam[12];        /* camera-to-world 3x4 row-major, ALREADY in ngp coordinates */
  float focal;          /* pixels; fov_axis = 0 (run_vis_on_poses.py:38) */
  float k1;             /* render_with_camera_distortion (ingp_utils.py:34) */
  float aabb_min[3], aabb_max[3]; /* render_aabb (ingp_utils.py:41-42) */
  float background[4];  /* ingp_utils.py:23-24 */
  float min_transmittance; /* ingp_utils.py:37 */
  int32_t width, height, spp;
  int32_t mode;         /* 0 Shade, 1 Depth (run_vis_on_poses.py:49-56) */
} pxt_ngp_view;

/* out_rgba: float32 [height][width][4], linear colour (render(..., linear=True)). */
int pxt_ngp_render(pxt_ngp* ctx, const pxt_ngp_view* view_host, float* out_rgba,
                   uint64_t* stats /* device, 4 counters (added to) or NULL: [0] samples composited, [1] rays that hit the
                                      render box, [2] wave steps that shaded samples (64 sample slots each), [3] hash-grid
                                      levels of those steps fetched as one 4 x 4 x 4 box per wave (of 16 per step) */,
                   void* stream);

/* Shade AND Depth of the SAME view in one march: bit-for-bit what two pxt_ngp_render calls
 * (mode 0, then mode 1) of this view produce, with the per-sample gathers and the density
 * network evaluated once.  pixtrack renders the mask (Depth, query camera) and the reference
 * image (Shade, SfM camera x reference_scale) at the same pose every frame
 * (pixloc_tracker_r9.py:224,227); when those two cameras coincide the tracker uses this. */
int pxt_ngp_render_both(pxt_ngp* ctx, const pxt_ngp_view* view_host, float* out_rgba, float* out_depth_rgba,
                        uint64_t* stats, void* stream);

/* pxt_ngp_render_both of a view whose CAMERA is not known on the host yet: `pose12` (device-readable, e.g. the pinned
 * record pxt_lm_refine writes: world->camera R row-major, then t, float32) is turned into the camera by a one-thread
 * kernel ahead of the render in `stream`, in the float64 arithmetic of the host chain it replaces
 * (pixtrack/utils/pose_utils.py:24-27 -> ingp_utils.py:47-63 -> Testbed.set_nerf_camera_matrix); view_host->cam is
 * ignored.  conv27 (host): nerf2sfm centroid[3], 3 / avglen, R[16] row-major, totp[3], then the snapshot's scale and
 * offset[3].  cam_out13 (optional, pinned host): the 12 camera floats + a completion word set last (system scope),
 * so that the caller can verify, once the pose reaches it, that the render used the bits it would have passed itself.
 * out_depth_rgba == NULL: ONE render in view_host->mode (0 Shade, 1 Depth) into out_rgba.
 * This lets a tracker enqueue frame t+1's render behind frame t's LM launch instead of after its result. */
int pxt_ngp_render_both_from_pose(pxt_ngp* ctx, const pxt_ngp_view* view_host, const float* pose12,
                                  const double* conv27_host, float* cam_out13, float* out_rgba,
                                  float* out_depth_rgba, uint64_t* stats, void* stream);

/* One render with the outputs the tracking loop consumes written by the renderer's last kernel itself, and - for a
 * render queued behind the LM launch - the camera taken from this context's camera slot instead of a conversion launch
 * (between-stage fusion).
 *   mode 0 Shade, 1 Depth, 2 Shade + Depth of the same view in one march (pxt_ngp_render_both).
 *   out->rgba / depth_rgba : the float32 images of pxt_ngp_render(_both); each may be NULL when its 8-bit stand-in is given
 *   out->rgb_u8  [H][W][3] : get_nerf_image's `(rgba[:, :, :3] * 255).astype(uint8)` of the Shade image
 *                            (pixtrack/visualization/run_vis_on_poses.py:52-54, alpha_thresh 0) - what pxt_rgba_to_u8 makes of rgba
 *   out->depth_nz [H][W]   : 1 where `uint8(depth * 255) != 0` - get_mask's plane before the morphology
 *                            (pixtrack/pose_trackers/pixloc_tracker_r9.py:210-212); feed it to pxt_depth_mask_plane
 *   camera_from_slot != 0  : view_host->cam is ignored; the 12 camera floats are read from pxt_ngp_camera_slot(ctx),
 *                            which an earlier kernel of the same stream wrote (pxt_lm_refine_cam's epilogue). */
typedef struct {
  float* rgba;
  float* depth_rgba;
  uint8_t* rgb_u8;
  uint8_t* depth_nz;
} pxt_ngp_outputs;
int pxt_ngp_render_frame(pxt_ngp* ctx, const pxt_ngp_view* view_host, int32_t mode, int32_t camera_from_slot,
                         const pxt_ngp_outputs* out, uint64_t* stats, void* stream);
float* pxt_ngp_camera_slot(pxt_ngp* ctx); /* device float[12], owned by the context */

/* pxt_ngp_render_frame for n_renders (<= PXT_NGP_MAX_BATCH) DIFFERENT contexts in ONE chain of launches on ONE stream:
 *   - a frame's two renders of different cameras - the mask's Depth render at the query camera and the reference image's
 *     Shade render at SfM camera 1 x reference_scale, at the same pose
 *     (pixtrack/pose_trackers/pixloc_tracker_r9.py:145-152, 207-214): modes {1, 0}, the second context made with
 *     pxt_ngp_create_shared;
 *   - K objects tracked in lock-step, one tracker (and one pyngp.Testbed) per object as
 *     pixtrack/pose_trackers/pixloc_tracker_r9.py:287-318 builds them: modes {2, 2, ...}.
 * Every launch of the chain carries a stage of every render (the shade stage of one half of the rays beside the march stage
 * of the other half: csrc/pxt_ngp.hip, ngp_stage_kernel), so nothing depends on how streams are dealt to hardware queues.
 * ctxs[k] / views_host[k] / modes[k] / outs_host[k] / stats[k] (stats or stats[k] may be NULL) are what K calls of
 * pxt_ngp_render_frame(ctxs[k], &views_host[k], modes[k], camera_from_slot, &outs_host[k], stats[k], stream) would take; the
 * images are bit for bit those calls' images.  Modes and sizes may differ per render.  batch_workspace: device memory of
 * pxt_ngp_batch_workspace_bytes(n_renders) bytes for the parameter records (may be NULL for n_renders <= 2, whose records
 * travel as kernel arguments); it must stay untouched until the chain has finished (a per-stream buffer is enough). */
#define PXT_NGP_MAX_BATCH 16
int64_t pxt_ngp_batch_workspace_bytes(int32_t n_renders);
int pxt_ngp_render_frame_batch(pxt_ngp* const* ctxs, const pxt_ngp_view* views_host, int32_t n_renders,
                               const int32_t* modes, int32_t camera_from_slot, const pxt_ngp_outputs* outs_host,
                               uint64_t* const* stats, void* batch_workspace, void* stream);

/* A second context over the SAME snapshot as `src`: own ray lists, counters and camera slot, shared (reference-counted)
 * hash table / MLP / occupancy tables - what a frame's second render needs (the testbed pixtrack drives is one object with
 * one snapshot, pixtrack/utils/ingp_utils.py:22-44, rendered twice per frame with different cameras).  Destroy each context
 * with pxt_ngp_destroy, in any order. */
int pxt_ngp_create_shared(pxt_ngp* src, pxt_ngp** out_ctx);

/* A render of >= 2^19 rays is cut into n pipes - equal slices of the rays - whose stages share the launches of the
 * render's chain one stage apart (the latency-bound march of one slice beside the gather-bound shade stage of
 * another): the image is bit for bit the same.  n = 0 restores the default (2, or $PXT_NGP_PIPES); n = 1: one pipe, every
 * launch carries ONE stage (isolated per-stage timing); n <= 4. */
int pxt_ngp_set_pipelines(pxt_ngp* ctx, int32_t n);

/* Live HIP-event timing of the renderer's dominant launches (those of ngp_stage_kernel that carry a shade stage: the hash-grid
 * gathers + both MLPs + compositing of a round's samples), for the
 * roofline line of bench.py.  every_nth > 0: those launches of every every_nth-th render
 * are bracketed by an event pair recorded on the render's own stream (an event record is a
 * marker packet between kernels, so sampling keeps the measurement from slowing what it
 * measures); 0 disables.  pxt_ngp_timing_read synchronises those events, returns their summed
 * duration (ms) and the number of timed launches, and clears the list. */
int pxt_ngp_timing_enable(pxt_ngp* ctx, int32_t every_nth);
int pxt_ngp_timing_read(pxt_ngp* ctx, float* total_ms_host, int32_t* n_launches_host);

/* Network query at caller-given points, ngp coordinates + unit view directions (device
 * float32 [n][3] each): out[n][4] = (density logit, r, g, b).  Same device code as the
 * renderer's per-sample evaluation; used by the parity tests (SURVEY KAT-7). */
int pxt_ngp_query(pxt_ngp* ctx, const float* pos, const float* dir, int32_t n, float* out, void* stream);

/* -------------------------------------------------------------------------
 * Small image ops on the path (host cv2/numpy calls in the reference).
 * ---------------------------------------------------------------------- */
/* get_mask (pixloc_tracker_r9.py:207-214): depth RGBA -> uint8((d*255)) != 0 ->
 * erode 5x5 x n_erode -> dilate 5x5 x n_dilate.  tmp: 2*H*W bytes. */
int pxt_depth_mask(const float* depth_rgba, int32_t H, int32_t W, int32_t n_erode,
                   int32_t n_dilate, uint8_t* mask_out, uint8_t* tmp, void* stream);
/* The same mask from the `uint8(depth * 255) != 0` byte plane a render wrote itself (pxt_ngp_render_frame's depth_nz):
 * bit for bit pxt_depth_mask of that render's float depth image.  tmp (2*H*W bytes) is only needed when
 * 2 * (n_erode + n_dilate) > 16 and may be NULL otherwise. */
int pxt_depth_mask_plane(const uint8_t* depth_nz, int32_t H, int32_t W, int32_t n_erode, int32_t n_dilate,
                         uint8_t* mask_out, uint8_t* tmp, void* stream);
/* get_nerf_image tail (run_vis_on_poses.py:52-54): alpha threshold, *255, ->uint8. */
int pxt_rgba_to_u8(const float* rgba, int32_t H, int32_t W, float alpha_thresh, uint8_t* rgb_out,
                   void* stream);
/* pixloc resize(image, size, max, "linear") == cv2.INTER_LINEAR on float32 HWC
 * (feature_extractor.py:45). */
int pxt_resize_linear(const float* src, int32_t H, int32_t W, int32_t C, float* dst, int32_t Ho,
                      int32_t Wo, void* stream);

/* Which pixels of an image that pxt_resize_linear produces can be non-zero (uint8 [Ho][Wo], 1 = may be non-zero): 0 only
 * where every source pixel under the 2 x 2 bilinear taps, grown by one pixel, is inactive - `mask` 0 where the source was
 * multiplied by a mask (a masked query, feature_extractor.py:41-45 resizes after masking), else a uint8 source that is 0
 * in all three channels (a NeRF render's background).  Passed on as the `mask` of pxt_unet_forward (a multiplication by
 * exactly 0 or 1 there), it gives the constant-tile skipping its source for images above the extractor's size limit. */
int pxt_resize_activity(const uint8_t* mask, const uint8_t* image_u8, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                        uint8_t* active_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIXTRACK_HIP_H */
