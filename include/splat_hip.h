/*
 * splat_hip.h -- C ABI of libsplat_hip.so, the MI355X (gfx950) Gaussian-splat
 * rasterizer (forward + backward).
 *
 * This is the drop-in boundary for the one hot path of SplaTAM: the un-vendored
 * extension `diff_gaussian_rasterization._C` (pinned at
 * /root/reference/requirements.txt:15, imported at
 * /root/reference/scripts/splatam.py:37 and /root/reference/utils/recon_helpers.py:2).
 * Each entry point below names the reference-side call it replaces.  Every
 * pointer is a raw DEVICE pointer owned by the caller (PyTorch's caching
 * allocator in the Python binding); the library never allocates, never frees,
 * never synchronises the device and launches everything on the stream it is
 * handed.  All functions return 0 on success and a SPLAT_E_* code otherwise
 * (nothing is thrown across the ABI); splat_error_string() maps codes to text.
 *
 * Conventions (/root/reference/utils/recon_helpers.py:8-13):
 *   - viewmatrix / projmatrix: 16 floats, element (row r, col c) at m[c*4+r],
 *     i.e. the bytes of the settings tensor `w2c^T` / `(P w2c)^T` as stored.
 *   - quaternions (r,x,y,z), not renormalised; opacities already in (0,1);
 *     scales already exponentiated (/root/reference/utils/slam_helpers.py:131-138).
 *   - images are planar [C][H][W] float32.
 */
#ifndef SPLAT_HIP_H
#define SPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPLAT_ABI_VERSION 10       /* 10: group binning behind the reference API (SplatState.group_* in splat_preprocess_forward / splat_render_forward,
                                      SPLAT_LAYOUT_GROUPS), SplatState.tile_recs (staged records handed from the forward to the backward composite), SplatCamera.bg == NULL = black, SplatGrads.flags (SPLAT_GRADS_UPSTREAM_SCALE), SplatState.status_host,
                                      SplatState.tile_order entries hold tile + 1 (a zeroed buffer is the natural order) and are laid out on request (SPLAT_LAYOUT_TILE_ORDER),
                                      SplatState.tile_queue (persistent composites: measured, not adopted, removed) is gone;
                                      9: scratch layouts (splat_workspace_bytes, splat_state_layout / _bind, splat_iter_workspace_layout / _bind);
                                      8: SplatIterWorkspace.d_cam is SPLAT_ITER_DCAM floats (loss terms, status snapshot, gated-iteration count),
                                      the Adam steps skip while the capacity flag is up (SplatAdamMap.gate), per-group bc2_sqrt;
                                      7: SplatState.long_items (work-item table of the multi-workgroup sort);
                                      6: SplatState.tile_row_begin / _end, SplatLossConfig.defer_finish, splat_iter_finish (tile-row-sharded tracking);
                                      5: SplatState.group_count / group_recs / group_stride (group binning), splat_iter_mapping_step; 4: densification (splat_iter_means2d_accumulate, splat_map_densify_select / _duplicate);
                                      3: SplatState.keys_alt / long_base (multi-workgroup sort of lists beyond LDS); 2: map edits,
                                      splat_iter_render / _tracking_step, outlier scratch in SplatIterWorkspace */
#define SPLAT_TILE 16            /* tile edge in pixels (one 256-thread workgroup per tile, one wave64 per 8x8 quadrant) */
#define SPLAT_MAX_CHANNELS 8     /* colour channels per call: 3 for the reference API, up to 8 for fused passes */
#define SPLAT_GRAD_STRIDE 16     /* floats per Gaussian in the backward accumulator (one 64-byte line) */
#define SPLAT_GROUP_TILES 2      /* group binning: a group is 2 x 2 tiles (SplatState.group_count) */
#define SPLAT_COUNTER_STRIDE 32  /* uint32 words between two tile counters: one 128-byte line per counter, so that
                                    the ~200 atomics a tile receives do not serialise with its neighbours' */

enum {
    SPLAT_OK = 0,
    SPLAT_E_INVALID = 1,         /* bad argument (null pointer, channel count, negative size) */
    SPLAT_E_LAUNCH = 2,          /* hipGetLastError() != hipSuccess after a launch */
    SPLAT_E_UNSUPPORTED = 3      /* feature not built into this library */
};

/* Mirrors the reference settings tuple field by field
 * (/root/reference/utils/recon_helpers.py:14-26); tensors become device pointers. */
typedef struct SplatCamera {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    const float *bg;             /* [channels], or NULL = black (every SplaTAM camera: /root/reference/utils/recon_helpers.py:17); with NULL
                                    the backward composite drops the background term of dL/dalpha at compile time */
    float scale_modifier;
    const float *viewmatrix;     /* [16] */
    const float *projmatrix;     /* [16] */
    int32_t sh_degree;
    const float *campos;         /* [3] (only read when shs != NULL) */
    int32_t prefiltered;
} SplatCamera;

/* The per-Gaussian inputs of GaussianRasterizer.__call__
 * (/root/reference/scripts/splatam.py:249; kwargs built at
 * /root/reference/utils/slam_helpers.py:131-138). */
typedef struct SplatGaussians {
    int32_t P;                   /* number of Gaussians */
    int32_t channels;            /* colour channels C (3 through the reference API) */
    const float *means3D;        /* [P][3] */
    const float *opacities;      /* [P] */
    const float *colors_precomp; /* [P][C] or NULL when shs is given */
    const float *scales;         /* [P][3] or NULL when cov3D_precomp is given */
    const float *rotations;      /* [P][4] or NULL when cov3D_precomp is given */
    const float *cov3D_precomp;  /* [P][6] or NULL */
    const float *shs;            /* [P][sh_coeffs][3] or NULL */
    int32_t sh_coeffs;           /* M */
} SplatGaussians;

/* State that lives from forward to backward (the reference keeps the same
 * information in its geomBuffer / binningBuffer / imgBuffer byte tensors). */
typedef struct SplatState {
    /* per-Gaussian geometry, written by splat_preprocess_forward */
    float *depth;                /* [P]     view-space z */
    float *xy;                   /* [P][2]  pixel centre */
    float *conic_opacity;        /* [P][4]  inverse 2D covariance (xx, xy, yy) + opacity */
    uint32_t *rect;              /* [P][2]  tile rect packed (minx | miny<<16, maxx | maxy<<16); max exclusive */
    int32_t *radii;              /* [P]     3-sigma radius in pixels, 0 = culled (an OUTPUT of the reference API) */
    float *rgb;                  /* [P][3]  colours evaluated from SH (NULL unless shs is used) */
    uint8_t *clamped;            /* [P][3]  SH clamp flags (NULL unless shs is used) */
    /* per-tile */
    uint32_t *tile_count;        /* [T * max(sub_bins, 1) * SPLAT_COUNTER_STRIDE] instances per tile, counter t at t * SPLAT_COUNTER_STRIDE */
    uint32_t *tile_base;         /* [T+1]   exclusive prefix sum of tile_count */
    uint32_t *tile_cursor;       /* [T * max(sub_bins, 1) * SPLAT_COUNTER_STRIDE] scatter cursors, same spacing (spare words of the
                                    lines hold the partial records of the tile scan) */
    /* per-instance ((Gaussian, tile) pairs) */
    uint64_t *keys;              /* [capacity] (float bits of depth << 32) | Gaussian id, bucketed by tile */
    uint32_t *point_list;        /* [capacity] Gaussian ids, each tile's slice sorted by key */
    int64_t capacity;
    /* optional (NULL: none): [capacity][12] floats, the STAGED RECORD of every list entry as the forward composite held it -- conic
     * (pre-scaled) + opacity, colours, centre, id, quadrant mask: 48 bytes, at the entry's position in point_list.  The forward
     * composite writes it (calls with up to three colour channels + depth, and the fused iteration's six), the backward composite
     * then re-stages a tile's list with one coalesced read per entry instead of gathering conic / centre / colours through the
     * id and repeating the culling tests.  Valid only for the backward pass that follows THAT forward pass on this state */
    float *tile_recs;
    /* scratch of the multi-workgroup sort of per-tile lists longer than 4096 entries (NULL: such a list is sorted in place
     * by one workgroup -- correct, but O(n log^2 n) barrier stages) */
    uint64_t *keys_alt;          /* [capacity] ping-pong partner of `keys` for the merge passes */
    uint32_t *long_base;         /* [T+1] first work item of every tile with a long list */
    uint32_t *long_items;        /* [capacity / 1024 + T + 1] tile of every work item (1024 keys of a long list), written by the scan
                                  * with long_base; NULL: the kernels find an item's tile by a binary search of long_base (13 dependent
                                  * reads per item: ~4x slower run sort) */
    /* group binning (fused iteration, bucketed lists short enough for the composite's own sort): the per-Gaussian kernel
     * files ONE record per touched GROUP of SPLAT_GROUP_TILES x SPLAT_GROUP_TILES tiles (slots taken per (workgroup, group)
     * through an LDS histogram: ~1 global atomic per Gaussian in random row order, far fewer in creation order, instead of one
     * per (Gaussian, tile) instance); the forward
     * composite of a tile filters its group's records by their tile rectangle, sorts, and publishes point_list / tile_count
     * exactly as the per-tile buckets would have held them.  NULL / group_stride 0: per-tile buckets */
    uint32_t *group_count;       /* [G * SPLAT_COUNTER_STRIDE] records per group, G = ceil(tiles_x / 2) * ceil(tiles_y / 2); zero between iterations */
    uint32_t *group_recs;        /* [G * group_stride][4] (Gaussian id, float bits of depth, rect word 0, rect word 1) */
    int32_t max_list_hint;       /* longest tile list if the host knows it (status[2] of an earlier read), 0 = unknown */
    int32_t order_hint;          /* fused iteration, bucketed lists: non-zero = the map is (mostly) in creation order (neighbouring rows
                                    are neighbouring pixels: SplaTAM appends one Gaussian per pixel in scan order), so a workgroup's
                                    instances fall on a few tiles and the bucket slots are taken per (workgroup, tile) through LDS;
                                    0 = unknown / random order: one returning atomic per instance.  Results do not depend on it */
    int32_t sub_bins;            /* exact path: counters per tile (a power of two; 0 / 1 = one).  tile_count / tile_cursor then hold
                                    T * sub_bins counters ((tile * sub_bins + (Gaussian index & (sub_bins - 1))) * SPLAT_COUNTER_STRIDE):
                                    spreads the count / scatter atomics of very long lists over several addresses.  The
                                    lists themselves are unchanged.  Must be 0 / 1 with bucketed lists */
    int32_t tile_stride;         /* 0: compact lists, tile t = [tile_base[t], tile_base[t+1]) (the exact path);
                                    > 0: BUCKETED lists (fused iteration only): tile t = [t*stride, t*stride + min(count, stride)),
                                    filled by the per-Gaussian kernel itself -- no scan, no scatter pass; a tile that
                                    receives more than `stride` instances sets status[1] and is truncated */
    int32_t group_stride;        /* > 0 (with group_count, group_recs, tile_stride > 0 and 0 < max_list_hint <= 819): group binning,
                                    records per group bucket (>= SPLAT_GROUP_TILES^2 * tile_stride can never overflow first) */
    /* fused iteration only: composite just the tile ROWS [tile_row_begin, tile_row_end) of the frame (0, 0 = all).  The
     * per-Gaussian kernels still run over the whole map; lists, planes, per-pixel state and loss sums are produced for those
     * rows only, and the per-Gaussian partial sums (hence the pose sums) hold those rows' share.  This is how the tracking
     * iteration shards over GPUs: every quantity the pose gradient needs is a SUM over pixels, so the ranks' copies of
     * SplatIterWorkspace.sums add up to the whole frame's (one all-reduce of 16 KB per iteration; splat_iter_finish) */
    int32_t tile_row_begin, tile_row_end;
    /* launch order of the composites' workgroups (fused iteration, whole frames; NULL / NULL = the natural order).  A launch of
     * 3 225 tile workgroups of very different length on 1 024 resident slots leaves a long tail (time-weighted occupancy 67-76 %,
     * profiles/r04_k7_account.md): the forward composite leaves a work estimate per tile in tile_work ([T]: the sum over the
     * tile's four 8x8 quadrants of the deepest list entry any pixel blended -- what the backward composite will walk), eight extra
     * workgroups of the iteration's last kernel turn it into tile_order ([8 * ceil(T / 8)]: the tiles of every XCD band, heaviest first; an
     * entry holds tile + 1, 0 = not written yet: the natural tile of the slot -- a ZERO-INITIALISED buffer is the natural order --,
     * 0xFFFFFFFF = no tile),
     * and the NEXT iteration that is given this tile_order buffer starts its
     * composites' workgroups in that order (a caller that alternates between views keeps one buffer per view: the estimate belongs
     * to the view it was measured on -- splatam_amd/fused.py).  Only a schedule: any permutation of each band gives the same results. */
    uint32_t *tile_work;
    uint32_t *tile_order;
    /* per-pixel */
    float *final_T;              /* [H][W] */
    int32_t *n_contrib;          /* [H][W] 1-based list position of the last contributor */
    /* status words: [0] num_rendered  [1] overflow flag (num_rendered > capacity)
     *               [2] longest tile list  [3] a list longer than the wave-sort limit met a stale max_list_hint
     *               that had skipped the long-list sort kernel (the lists are then NOT sorted: re-run) */
    int32_t *status;             /* [4] */
    /* optional (NULL: none): ONE int32 in pinned HOST memory that the group-binning kernels set to 1 whenever they raise status[1] or
     * status[3] -- a caller that never waits for the device between the forward and the backward pass reads it (after an event that
     * follows the forward composite) without queueing a copy.  The caller zeroes it before the call. */
    int32_t *status_host;
    /* optional (NULL: none; group binning behind the reference API only): the backward accumulator [P][SPLAT_GRAD_STRIDE] that the backward
     * pass of THIS forward pass will use -- splat_preprocess_forward zeroes row i beside Gaussian i's geometry (stores the kernel has
     * room for), and splat_backward with SPLAT_GRADS_ACCUM_ZEROED then skips its 64-byte-per-Gaussian memset launch */
    float *accum_to_zero;
} SplatState;

const char *splat_error_string(int code);
int splat_abi_version(void);
/* sizeof() of the struct named `name` ("SplatState", "SplatIterWorkspace", ...) as this library was compiled, 0 for an unknown
 * name: a binding that mirrors the structs by hand (ctypes, cgo, JNA) checks its layout against it at load time. */
size_t splat_sizeof(const char *name);

size_t splat_num_tiles(int32_t width, int32_t height);


/* GROUP BINNING behind the reference API (ABI 10) -- the front end of the fused iteration for callers of splat_forward / splat_backward
 * who know that the scene's per-tile lists are short (an earlier call on the same scene left its longest list in status[2]):
 *   st->tile_stride  = S > 0      bucket of tile t in point_list = [t * S, t * S + count); S <= 1024 entries is all the composite sorts
 *   st->max_list_hint = L with L + L / 4 <= 1024   (the longest list the caller expects)
 *   st->group_count / group_recs / group_stride (>= 4 * S can never overflow first), st->capacity >= S * tiles; st->keys may be NULL
 * splat_preprocess_forward then zeroes group_count and status, and files ONE 16-byte record per (Gaussian, touched group of 2 x 2 tiles);
 * splat_bin_forward is a no-op; splat_render_forward (3 colour channels) filters, sorts and publishes every tile's list itself
 * (point_list, tile_count) and composites; splat_backward walks the published lists.  Two launches forward instead of six, and NOTHING
 * the host has to read before the composite may be launched (the exact path reads status[0] to size keys / point_list, as the CUDA
 * original reads num_rendered).  A list longer than S or 1024 entries, or a group bucket that overflowed, raises status[1] / status[3]:
 * the images and gradients of that call are then built from TRUNCATED lists (memory-safe, wrong) and the caller must repeat the call
 * on exact lists (tile_stride = 0).  status[0] and status[2] are not maintained in this mode.
 * The lists of this mode are the library's own business (built, sorted, published and replayed by its kernels; `rect` / `radii` keep the
 * reference's values): a Gaussian is filed only in those tiles of its ceil(3 sigma) rectangle that can hold a pixel with
 * alpha >= 1/255 (csrc/splat_math.h live_tile_rect) -- 12-13 % fewer entries at the SplaTAM workloads, identical images and gradients
 * (a dropped entry fails the alpha test at every pixel of its tile).  The exact path keeps the reference's lists entry for entry. */

/* K1 + tile scan.  Replaces the first half of `_C.rasterize_gaussians`
 * (preprocess, prefix sum).  Writes depth/xy/conic_opacity/rect/radii,
 * tile_count/tile_base/tile_cursor and status[0..2].  The host may read
 * status[0] (num_rendered) to size keys/point_list, as the reference does.
 * The scan compares the count with st->capacity and publishes EMPTY ranges (status[1] set) when it is larger: a caller that
 * allocates its lists AFTER this call sets st->capacity to an upper bound (e.g. INT64_MAX / 2) for this call and to the real
 * capacity before splat_bin_forward (tests/capi_smoke.c, splatam_amd/rasterizer.py do). */
int splat_preprocess_forward(const SplatCamera *cam, const SplatGaussians *g, SplatState *st, void *stream);

/* Scatter (Gaussian, tile) instances into per-tile buckets and sort every
 * bucket by (depth bits, id).  Replaces duplicateWithKeys + the global radix
 * sort + identifyTileRanges of the reference.  A no-op that leaves status[1]
 * set when num_rendered > st->capacity.
 * CONTRACT: the second half of ONE forward pass -- call it on the SplatState that splat_preprocess_forward has just filled, with the
 * same cam, the same g->P and unchanged st->sub_bins / st->tile_stride (the caller may only set keys / point_list / capacity in
 * between, after reading status[0]).  With sub_bins > 1 the per-tile counters are split over sub-bins by (workgroup of 4 096
 * Gaussians) and the scatter walks the SAME partition the count used; a caller that fills tile_count itself, or changes P or
 * sub_bins between the two calls, gets ranges that do not match the scatter (undefined lists).  splat_forward() does both. */
int splat_bin_forward(const SplatCamera *cam, const SplatGaussians *g, SplatState *st, void *stream);

/* Tile-wise alpha compositing.  out_color [C][H][W], out_depth [H][W].
 * `colors` is [P][C]: colors_precomp, or st->rgb when SH were evaluated. */
int splat_render_forward(const SplatCamera *cam, const SplatGaussians *g, SplatState *st,
                         float *out_color, float *out_depth, void *stream);

/* All three of the above back to back (capacity must already be sufficient). */
int splat_forward(const SplatCamera *cam, const SplatGaussians *g, SplatState *st,
                  float *out_color, float *out_depth, void *stream);

/* Gradient outputs of `_C.rasterize_gaussians_backward`
 * (autograd backward reached from /root/reference/scripts/splatam.py:702,854). */
typedef struct SplatGrads {
    const float *dL_dcolor;      /* [C][H][W] incoming gradient of out_color */
    float *accum;                /* [P][SPLAT_GRAD_STRIDE] scratch, zeroed by the library */
    float *dL_dmeans3D;          /* [P][3] */
    float *dL_dmeans2D;          /* [P][3] NDC-space gradient of the projected centre, z = 0 */
    float *dL_dcolors;           /* [P][C] (NULL when shs are used) */
    float *dL_dopacities;        /* [P] */
    float *dL_dscales;           /* [P][3] or NULL.  Gradient w.r.t. the UNMODIFIED scales the caller passed: it carries the factor
                                    cam->scale_modifier (Sigma = R diag((modifier s)^2) R^T).  DEVIATION from the CUDA original as recalled
                                    (SURVEY.md Appendix A): its computeCov3D adjoint returns dL/d(modifier s) without that factor.  Equal at
                                    modifier 1.0 -- every SplaTAM configuration, the viewers render without gradients --; a caller that
                                    trains with another modifier and wants the original's numbers divides by it.  The oracle follows this
                                    library (oracle/raster_ref.c), tests/test_gpu_configs.py covers modifier 1.6 */
    float *dL_drotations;        /* [P][4] or NULL */
    float *dL_dcov3D;            /* [P][6] or NULL */
    float *dL_dshs;              /* [P][M][3] or NULL */
    int32_t flags;               /* SPLAT_GRADS_* */
} SplatGrads;
#define SPLAT_GRADS_POISON_IF_FLAGGED 2 /* group binning behind the reference API: splat_preprocess_backward writes NaN into EVERY gradient output when
                                        the forward pass of this state raised status[1] or status[3] (its lists were truncated).  For a caller that
                                        launches the backward pass without having looked at the flags: gradients formed on truncated lists never
                                        reach an optimizer as plausible numbers */
#define SPLAT_GRADS_ACCUM_ZEROED 4 /* gr->accum is zero already (SplatState.accum_to_zero of the forward pass): splat_render_backward does not clear it */
#define SPLAT_GRADS_UPSTREAM_SCALE 1 /* dL_dscales WITHOUT the factor cam->scale_modifier: the numbers the CUDA original returns (its computeCov3D
                                        adjoint forms dL/d(modifier * s) and hands it out as dL/ds; see dL_dscales above).  Identical at modifier 1.0 */

/* Per-pixel back-to-front replay; accumulates per-Gaussian partial sums into gr->accum. */
int splat_render_backward(const SplatCamera *cam, const SplatGaussians *g, const SplatState *st,
                          SplatGrads *gr, void *stream);

/* Per-Gaussian chain rule from gr->accum to every dL_d* output. */
int splat_preprocess_backward(const SplatCamera *cam, const SplatGaussians *g, const SplatState *st,
                              SplatGrads *gr, void *stream);

/* Both of the above back to back. */
int splat_backward(const SplatCamera *cam, const SplatGaussians *g, const SplatState *st,
                   SplatGrads *gr, void *stream);

/* `_C.mark_visible` of the reference extension: present[i] = (view-space z > 0.2). */
int splat_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, uint8_t *present, void *stream);

/* Do two rasterizer calls see the same geometry?  *differ (device int32) = 0 when opacities [P], scales [P][3] and rotations [P][4]
 * of call A and call B hold the same bits, non-zero otherwise.  The two renders of the reference's get_loss
 * (/root/reference/scripts/splatam.py:249,253) receive equal-valued but distinct tensors (sigmoid / exp / normalize are applied
 * twice: /root/reference/utils/slam_helpers.py:124-139, 234-249); a binding that has proven the rest equal (same means3D storage,
 * same camera) may then hand the second call the first call's SplatState -- geometry and sorted lists -- and run only
 * splat_render_forward for it. */
int splat_same_geometry(int32_t P, const float *opacities_a, const float *opacities_b, const float *scales_a, const float *scales_b,
                        const float *rotations_a, const float *rotations_b, int32_t *differ, void *stream);

/* Kernel-only timing helper for bench.py: runs `fn` (0 = render forward,
 * 1 = render backward) `iters` times on `stream` between two hipEvents created
 * on that same stream and returns the mean milliseconds per launch in *ms. */
int splat_time_kernel(int fn, int iters, const SplatCamera *cam, const SplatGaussians *g, SplatState *st,
                      SplatGrads *gr, float *out_color, float *out_depth, void *stream, float *ms);


/* ------------------------------------------------------------------------------------------------------------
 * Fused SplaTAM iteration: everything the reference does in Python around the two
 * rasterizer calls of one optimisation iteration, as a handful of kernels with no host synchronisation:
 *   transform_to_frame + transformed_params2rendervar + transformed_params2depthplussilhouette
 *     (/root/reference/utils/slam_helpers.py:252-304, :124-139, :234-249)  -> one per-Gaussian kernel,
 *   the RGB and the depth/silhouette renders (/root/reference/scripts/splatam.py:249,253)
 *     -> ONE 6-channel composite (r, g, b, z, 1, z^2) over shared geometry and lists,
 *   the masked losses of get_loss (/root/reference/scripts/splatam.py:256-290, calc_ssim
 *     /root/reference/utils/slam_external.py:54-97) and their gradients -> per-pixel kernels,
 *   loss.backward() through the render variables and the camera pose -> one per-Gaussian kernel,
 *   optimizer.step() (/root/reference/scripts/splatam.py:160-166,704,860) -> fused Adam kernels,
 *   the best-candidate pose bookkeeping of the tracking loop (/root/reference/scripts/splatam.py:706-711).
 * Every argument of get_loss is supported (use_l1, use_sil_for_loss, ignore_outlier_depth_loss with torch.median's
 * exact lower median, tracking / mapping / do_ba); the one thing the fused path does not produce is the colour pass'
 * own means2D.grad (gradient-based densification, off in every SLAM config): the caller uses the two-call path for it.
 * ------------------------------------------------------------------------------------------------------------ */

/* The reference's `params` dict (/root/reference/scripts/splatam.py:120-157).  Adam updates it in place. */
typedef struct SplatMap {
    int32_t P;
    int32_t isotropic;           /* log_scales is [P][1] (1) or [P][3] (0) */
    float *means3D;              /* [P][3] world frame */
    float *rgb_colors;           /* [P][3] */
    float *unnorm_rotations;     /* [P][4] */
    float *logit_opacities;      /* [P] */
    float *log_scales;           /* [P][1] or [P][3] */
    float *cam_unnorm_rots;      /* [1][4][num_frames] */
    float *cam_trans;            /* [1][3][num_frames] */
    int32_t num_frames;
} SplatMap;

/* `curr_data` of get_loss. */
typedef struct SplatFrameData {
    const float *im;             /* [3][H][W] */
    const float *depth;          /* [1][H][W] */
    const float *w2c;            /* [16] row-major curr_data['w2c'] (first-frame world-to-camera) */
    int32_t time_idx;            /* iter_time_idx: which camera pose of the map transforms the Gaussians */
} SplatFrameData;

/* Arguments of get_loss (/root/reference/scripts/splatam.py:214-216). */
typedef struct SplatLossConfig {
    int32_t tracking;            /* 1: tracking=True (summed losses); 0: mapping=True (means + SSIM) */
    int32_t camera_grad;         /* transform_to_frame(camera_grad=...) */
    int32_t gaussians_grad;      /* transform_to_frame(gaussians_grad=...) */
    int32_t use_sil_for_loss;
    float sil_thres;
    int32_t use_l1;
    int32_t ignore_outlier_depth_loss;
    float w_im;                  /* loss_weights['im'] */
    float w_depth;               /* loss_weights['depth'] */
    int32_t defer_finish;        /* 1: stop before the last kernel (pose gradient, loss value, pose Adam step): the caller completes the
                                    partial sums (all-reduce over the ranks that composited other tile rows) and calls splat_iter_finish */
    int32_t fused_composite;     /* tracking with a pixel-local loss (no outlier rejection), lists the composite sorts itself (with map
                                    gradients wanted -- ws->d_rgb_colors / d_logit_opacities -- the kernel carries the backward composite's
                                    mapping form): 1 = forward composite, loss and backward composite as ONE kernel (the backward pass
                                    walks the batch the forward pass left in LDS; ws->out6, dL_dout6, st.final_T, st.n_contrib are NOT
                                    written), 2 = the same kernel, planes written as well, 0 = two kernels.  Ignored where it does not apply */
} SplatLossConfig;

#define SPLAT_ITER_SUMS 32       /* doubles per copy of the partial sums */
#define SPLAT_ITER_SUM_COPIES 64 /* copies: workgroups spread their atomics over them (one hot cache line otherwise) */
#define SPLAT_ITER_DCAM 32       /* floats of SplatIterWorkspace.d_cam */

/* Device scratch + outputs of one fused iteration; every array is caller-owned.  The caller ZERO-INITIALISES sums,
 * st.tile_count, accum and dL_dout6 once; each iteration leaves them zeroed again (the kernels that consume a buffer
 * reset it), so the steady state has no memset launches. */
typedef struct SplatIterWorkspace {
    SplatState st;               /* geometry, lists (fixed capacity; overflow -> st.status[1]) and per-pixel state */
    float *feat8;                /* [P][8]  r, g, b, z, 1, z^2, 0, 0 */
    float *out6;                 /* [6][H][W] rendered r, g, b, depth, silhouette, depth^2 */
    float *dL_dout6;             /* [6][H][W] */
    float *accum;                /* [P][SPLAT_GRAD_STRIDE] */
    float *ssim_maps;            /* [9][H][W] mapping only (NULL for tracking) */
    double *sums;                /* [SPLAT_ITER_SUM_COPIES][SPLAT_ITER_SUMS]: [0] masked depth L1 sum, [1] image L1 sum,
                                    [2] mask count, [3] SSIM map sum, [8..23] camera-pose partial sums, [31] != 0: a capacity flag was
                                    raised on some band of a tile-row-sharded iteration (cfg.defer_finish) */
    float *max_2D_radius;        /* [P] variables['max_2D_radius'], updated in place, or NULL */
    /* gradients of the map (written when non-NULL; means3D / unnorm_rotations only with gaussians_grad) */
    float *d_means3D;            /* [P][3] */
    float *d_rgb_colors;         /* [P][3] */
    float *d_unnorm_rotations;   /* [P][4] */
    float *d_logit_opacities;    /* [P] */
    float *d_log_scales;         /* [P][1|3] */
    float *d_cam;                /* [SPLAT_ITER_DCAM] the iteration's report, written by its last kernel:
                                    [0..3] dL/dcam_unnorm_rots[...,t], [4..6] dL/dcam_trans[...,t], [7] loss,
                                    [8..11] the raw sums [0..3] of this iteration,
                                    [12] STICKY "lists overflowed / unsorted" flag (set by any iteration whose status[1] or
                                         status[3] was set; cleared by the host).  While it is up the Adam steps of this ABI
                                         (inside splat_iter_tracking_step / _mapping_step / _finish, splat_iter_adam_pose, and
                                         splat_iter_adam_map with SplatAdamMap.gate) leave parameters, moments and the
                                         best-candidate record untouched: an iteration on truncated lists never moves the map,
                                    [13] the median depth error of this iteration when ignore_outlier_depth_loss is set,
                                    [14] loss_weights['depth'] * the depth term, [15] loss_weights['im'] * the image term of [7]
                                         (weighted_losses['depth' / 'im'] of /root/reference/scripts/splatam.py:339),
                                    [16..19] int32 bit patterns of st.status[0..3] as this iteration left them (instances,
                                         overflow, longest list, stale hint),
                                    [20] int32: 1 when THIS iteration was flagged, [21] int32: iterations whose Adam step was
                                         skipped since the host last cleared it, 10 spare */
    /* only read when cfg->ignore_outlier_depth_loss (/root/reference/scripts/splatam.py:266-268), NULL otherwise */
    float *outlier_err;          /* [H*W] scratch: depth error per pixel */
    uint32_t *outlier_scratch;   /* splat_map_scratch_words(H*W) words: histograms of the radix selection of torch.median */
} SplatIterWorkspace;

/* ------------------------------------------------------------------------------------------------------------
 * Scratch layouts: what a binding needs to size and wire the caller-owned scratch WITHOUT reading the Python binding
 * (`splat_workspace_bytes(P, W, H, R_cap)`).  The reference extension sizes its geomBuffer / binningBuffer /
 * imgBuffer through a resize callback into torch byte tensors; here the library describes ONE slab: every array of a SplatState
 * (+ SplatGrads.accum) or of a SplatIterWorkspace by field name, with its size, its offset in the slab (SPLAT_SLAB_ALIGN-aligned)
 * and whether the caller must zero it once before the first call.  `splat_*_bind` writes the pointers into the struct.
 * A caller may also allocate the arrays one by one from `bytes` and set the pointers itself (the Python binding of the fused
 * iteration does: its per-Gaussian arrays grow with the map).
 * ------------------------------------------------------------------------------------------------------------ */
#define SPLAT_SLAB_ALIGN 256
typedef struct SplatArrayInfo {
    const char *name;            /* field name in SplatState / SplatGrads / SplatIterWorkspace ("st.keys" for the embedded state) */
    size_t bytes;
    size_t offset;               /* in the slab */
    int32_t zero_init;           /* 1: zero it once before the first call (the kernels leave it zeroed afterwards) */
} SplatArrayInfo;
#define SPLAT_LAYOUT_SH 1        /* shs are used: SplatState.rgb / clamped */
#define SPLAT_LAYOUT_LONG_LISTS 2 /* lists may exceed 4 096 entries: keys_alt, long_items (always needed while the lengths are unknown) */
#define SPLAT_LAYOUT_BACKWARD 4  /* SplatGrads.accum */
#define SPLAT_LAYOUT_SSIM 8      /* fused iteration: ssim_maps (mapping) */
#define SPLAT_LAYOUT_OUTLIER 16  /* fused iteration: outlier_err / outlier_scratch (ignore_outlier_depth_loss) */
#define SPLAT_LAYOUT_GROUPS 32   /* splat_state_layout: group binning behind the reference API -- group_count (right in front of status: the library
                                    zeroes both with one memset), group_recs for `group_stride` = 4 * tile_stride records per group, where
                                    tile_stride = capacity / tiles; no key buckets */
#define SPLAT_LAYOUT_RECS 128     /* tile_recs: 48 bytes x capacity (a forward pass that a backward pass will follow) */
#define SPLAT_LAYOUT_TILE_ORDER 64 /* splat_iter_workspace_layout / _bind: st.tile_work, st.tile_order (launch order of the composites; the
                                    library's first iteration on a workspace treats an order buffer it has not written yet as the natural
                                    order -- see SplatState.tile_order) */
#define SPLAT_LAYOUT_MAX_ARRAYS 48
/* One rasterizer call (forward, + backward with SPLAT_LAYOUT_BACKWARD): arrays of SplatState for P Gaussians, a width x height
 * image, `sub_bins` counters per tile (0 / 1 = one) and lists of `capacity` instances.  Writes up to `max_entries` entries to `out`
 * (NULL: count only) and the slab size to *total_bytes; returns the number of arrays, or -SPLAT_E_INVALID. */
int splat_state_layout(int32_t P, int32_t width, int32_t height, int32_t sub_bins, int64_t capacity, int32_t flags,
                       SplatArrayInfo *out, int32_t max_entries, size_t *total_bytes);
/* Points the fields of *st (and gr->accum when gr != NULL and the layout has it) into `slab` and sets st->capacity / st->sub_bins;
 * fields the layout does not name are left as they are. */
int splat_state_bind(SplatState *st, SplatGrads *gr, void *slab, const SplatArrayInfo *arrays, int32_t n, int32_t sub_bins,
                     int64_t capacity);
/* The slab size of splat_state_layout(P, width, height, 1, capacity, SPLAT_LAYOUT_LONG_LISTS | SPLAT_LAYOUT_BACKWARD): enough for
 * any call without SHs whose lists hold at most `capacity` instances. */
size_t splat_workspace_bytes(int32_t P, int32_t width, int32_t height, int64_t capacity);
/* The fused iteration's SplatIterWorkspace (its embedded state as "st.<field>"): `group_stride` records per group bucket (0: no group
 * binning).  Not in the layout: the gradient outputs d_* and max_2D_radius (the caller's own tensors). */
int splat_iter_workspace_layout(int32_t P, int32_t width, int32_t height, int64_t capacity, int32_t group_stride, int32_t flags,
                                SplatArrayInfo *out, int32_t max_entries, size_t *total_bytes);
int splat_iter_workspace_bind(SplatIterWorkspace *ws, void *slab, const SplatArrayInfo *arrays, int32_t n, int64_t capacity,
                              int32_t group_stride);
size_t splat_iter_workspace_bytes(int32_t P, int32_t width, int32_t height, int64_t capacity, int32_t group_stride, int32_t flags);

/* get_loss + loss.backward() of one iteration.  On return (stream order) ws->d_* hold the gradients and
 * ws->d_cam[7] the loss value.  The fused iteration renders on a ZERO background, as setup_camera builds it
 * (/root/reference/utils/recon_helpers.py:17): cam->bg must point at six zeros (the backward composite drops the term). */
int splat_iter_loss_backward(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame,
                             const SplatLossConfig *cfg, SplatIterWorkspace *ws, void *stream);

/* torch.optim.Adam over the five Gaussian groups (mapping: /root/reference/scripts/splatam.py:160-166 with
 * eps = 1e-15, lr per group from the config).  step_size[k] = lr_k / (1 - beta1^t_k) and bc2_sqrt[k] = sqrt(1 - beta2^t_k)
 * are formed by the caller in double, as torch does -- per group, because torch keeps a step count per parameter and a
 * parameter the caller re-created (pruning, opacity reset) restarts or skips its count.  Group order: means3D, rgb_colors,
 * unnorm_rotations, logit_opacities, log_scales.  exp_avg / exp_avg_sq have the shapes of the parameters.  gate: NULL, or
 * the d_cam of the iteration that formed the gradients -- the step is skipped while d_cam[12] (the capacity flag) is up. */
typedef struct SplatAdamMap {
    float beta1, beta2, eps;
    float bc2_sqrt[5];
    float step_size[5];
    const float *grad[5];
    float *exp_avg[5];
    float *exp_avg_sq[5];
    const float *gate;
} SplatAdamMap;
int splat_iter_adam_map(const SplatMap *map, const SplatAdamMap *opt, void *stream);

/* Adam step of the camera pose of frame `time_idx` (tracking: default eps 1e-8) followed by the reference's
 * best-candidate bookkeeping: if loss < min_loss, remember the UPDATED pose.  d_cam: the report of the iteration
 * (gradient [0..6], loss [7]); skipped while d_cam[12] is up.
 * state [24] floats: exp_avg q(4) t(3), exp_avg_sq q(4) t(3), min_loss, candidate q(4) t(3), 2 spare. */
#define SPLAT_POSE_STATE 24
int splat_iter_adam_pose(const SplatMap *map, int32_t time_idx, const float *d_cam, float *state,
                         float beta1, float beta2, float eps, float bc2_sqrt, float step_size_rot, float step_size_trans,
                         void *stream);

/* One whole tracking iteration (/root/reference/scripts/splatam.py:690-711) in one call: splat_iter_loss_backward with
 * cfg->tracking set, with the pose's Adam step and the best-candidate bookkeeping of splat_iter_adam_pose folded into its last
 * kernel (one launch less per iteration).  `state` and the step sizes as for splat_iter_adam_pose. */
typedef struct SplatPoseAdam {
    float *state;                /* [SPLAT_POSE_STATE] */
    float beta1, beta2, eps, bc2_sqrt, step_size_rot, step_size_trans;
} SplatPoseAdam;
int splat_iter_tracking_step(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame,
                             const SplatLossConfig *cfg, SplatIterWorkspace *ws, const SplatPoseAdam *adam, void *stream);

/* The last kernel of an iteration whose splat_iter_loss_backward / splat_iter_tracking_step ran with cfg->defer_finish: totals
 * ws->sums into ws->d_cam (pose gradient, loss), resets them, and takes the pose's Adam step when `adam` is given (as
 * splat_iter_tracking_step would have).  Tile-row-sharded tracking: all-reduce ws->sums (SPLAT_ITER_SUM_COPIES x SPLAT_ITER_SUMS
 * doubles, sum) over the ranks between the two calls; every rank then holds the same sums and takes the same step. */
int splat_iter_finish(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame, const SplatLossConfig *cfg,
                      SplatIterWorkspace *ws, const SplatPoseAdam *adam, void *stream);

/* Between an iteration that ran with cfg->defer_finish and its all-reduce: folds the SPLAT_ITER_SUM_COPIES copies of ws->sums into
 * copy 0 and zeroes the others, so that the exchange carries SPLAT_ITER_SUMS doubles (256 bytes) instead of all the copies (16 KB);
 * splat_iter_finish totals the copies either way. */
int splat_iter_fold_sums(double *sums, void *stream);

/* One whole single-view mapping iteration (/root/reference/scripts/splatam.py:846-863: get_loss, backward, optimizer.step) in
 * one call: splat_iter_loss_backward with cfg->tracking clear, with splat_iter_adam_map folded into its last kernel (every
 * Gaussian's parameters, gradients and moments are touched once).  adam->grad[k] != NULL: group k is stepped; it must then be the
 * ws->d_* buffer of group k (the gradients are still written there) -- or ws->d_* of the group is NULL and the gradient is formed,
 * stepped on and NOT stored (a loop that discards its gradients after the step).  Not for the view-sharded batch: there the gradients are
 * exchanged between splat_iter_loss_backward and splat_iter_adam_map. */
int splat_iter_mapping_step(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame,
                            const SplatLossConfig *cfg, SplatIterWorkspace *ws, const SplatAdamMap *adam, void *stream);

/* Kernel-only timing helper for bench.py on the fused path: fn 0 = 6-channel composite forward reading published lists, 1 = 6-channel
 * composite backward (the kernel alone: the launches accumulate on top of each other and the accumulator is zeroed again AFTER the
 * timed bracket, so the workspace stays usable), 2 = the forward composite in the form the iteration launches on short lists
 * (it filters its group's records / reads its bucket, sorts and publishes the tile's list itself; needs bucketed lists and a list
 * length hint <= 819, SPLAT_E_INVALID otherwise), 3 = forward (sorting form when the state allows it) and backward composite
 * alternating, `iters` pairs (time(3) - time(2 or 0) = the backward composite between other kernels), 4 = the backward composite in its
 * tracking form (depth colour sum only, no opacity sum), launched `iters` times on `stream` between two hipEvents; the workspace must hold
 * the state of a completed splat_iter_loss_backward. */
int splat_iter_time_kernel(int fn, int iters, const SplatCamera *cam, int32_t P, SplatIterWorkspace *ws, void *stream, float *ms);

/* ------------------------------------------------------------------------------------------------------------
 * Map growth and maintenance: the per-frame callers that change the NUMBER of Gaussians.
 * The reference re-allocates every parameter tensor with torch.cat / boolean indexing
 * (/root/reference/scripts/splatam.py:378-420, /root/reference/utils/slam_external.py:139-188); here the map is a
 * capacity-managed struct of arrays that is edited in place: rows are appended at the tail / compacted stably, so
 * that after an edit the first `count` rows hold exactly what the reference's tensors would hold, in the same order.
 * Every edit leaves its result count in store->counts (device); the host reads it once per edit (the reference
 * synchronises at the same places: `if torch.sum(non_presence_mask) > 0`, boolean indexing).
 * ------------------------------------------------------------------------------------------------------------ */

/* The map arrays with room for `capacity` rows + what has to move with them. */
typedef struct SplatMapStore {
    SplatMap map;                /* map.P = rows in use, as the HOST knows them (it is the launch bound) */
    int32_t capacity;            /* rows every per-Gaussian array can hold */
    float *exp_avg[5];           /* Adam moments of the live optimizer, group order of SplatAdamMap; NULL = none */
    float *exp_avg_sq[5];
    float *max_2D_radius;        /* variables['max_2D_radius'] [capacity] or NULL */
    float *means2D_gradient_accum; /* variables['means2D_gradient_accum'] [capacity] or NULL */
    float *denom;                /* variables['denom'] [capacity] or NULL */
    float *timestep;             /* variables['timestep'] [capacity] or NULL */
    int32_t *counts;             /* device [8]: [0] rows after the edit, [1] rows appended / removed by it,
                                    [2] the append did not fit `capacity` (then nothing was written),
                                    [3] add_new_gaussians: sum(non_presence_mask) BEFORE the valid-depth mask,
                                    [4] float bits of the median used by add_new_gaussians, 3 spare */
} SplatMapStore;

/* Forward-only pass of the fused composite (kernels F1, K2..K4, K6; no loss, no gradients, nothing accumulated):
 * ws->out6 = r, g, b, depth, silhouette, depth^2 of `map` seen from pose `frame->time_idx`.  This is the render of
 * add_new_gaussians (/root/reference/scripts/splatam.py:381-385) and of the evaluation / keyframe code.
 * frame->im / frame->depth may be NULL. */
int splat_iter_render(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame, SplatIterWorkspace *ws,
                      void *stream);

/* Which pixels of an RGB-D frame become new Gaussians. */
enum {
    SPLAT_ADD_VALID_DEPTH = 0,   /* initialize_first_timestep: mask = depth > 0 (/root/reference/scripts/splatam.py:197-202) */
    SPLAT_ADD_NON_PRESENCE = 1   /* add_new_gaussians: silhouette < sil_thres, or rendered depth behind the measured one
                                    by more than 50 x the median depth error; and depth > 0 (:386-408) */
};

typedef struct SplatAddArgs {
    int32_t mode;                /* SPLAT_ADD_* */
    int32_t width, height;
    const float *im;             /* [3][H][W] curr_data['im'] */
    const float *depth;          /* [1][H][W] curr_data['depth'] */
    const float *out6;           /* [6][H][W] render at the tracked pose (planes 3 = depth, 4 = silhouette are read);
                                    NULL for SPLAT_ADD_VALID_DEPTH */
    float fx, fy, cx, cy;        /* curr_data['intrinsics'] (the densification intrinsics) */
    float sil_thres;
    int32_t time_idx;            /* SPLAT_ADD_NON_PRESENCE: pose of the map that back-projects; value of variables['timestep'] */
    const float *w2c;            /* SPLAT_ADD_VALID_DEPTH: [16] row-major rigid world-to-camera of the frame (device) */
    float *err;                  /* scratch [H*W] floats (depth error) */
    uint32_t *scratch;           /* scratch, splat_map_scratch_words(width*height) uint32 words */
} SplatAddArgs;

/* uint32 words of SplatAddArgs.scratch / SplatPruneArgs.scratch for `n` pixels / rows */
size_t splat_map_scratch_words(int64_t n);

/* get_pointcloud + initialize_new_params + the torch.cat of add_new_gaussians / initialize_params
 * (/root/reference/scripts/splatam.py:67-116, 350-376, 378-420): appends one Gaussian per selected pixel, in pixel
 * order: means3D = c2w [x z, y z, z, 1] with (x, y) = ((u - cx) / fx, (v - cy) / fy), rgb from `im`, rotation (1,0,0,0),
 * logit opacity 0, log scale log(sqrt((z / ((fx + fy) / 2))^2)) (1 or 3 columns), timestep = time_idx; when anything was
 * selected before the valid-depth mask, max_2D_radius / means2D_gradient_accum / denom are zeroed for ALL rows, as the
 * reference does.  Moments (exp_avg*) of appended rows are zeroed.  counts[0..4] are written. */
int splat_map_add_new_gaussians(SplatMapStore *store, const SplatAddArgs *args, void *stream);

typedef struct SplatPruneArgs {
    float removal_opacity_threshold; /* remove rows with sigmoid(logit_opacity) < this */
    int32_t remove_big;          /* also remove rows whose largest exp(log_scale) > big_scale */
    float big_scale;             /* 0.1 * variables['scene_radius'] */
    const uint8_t *to_remove;    /* [P] caller-supplied flags (remove_points); NULL = form them from the two rules above */
    uint8_t *flags;              /* scratch [capacity] */
    float *stage;                /* scratch: ((capacity + 3) & ~3) * splat_map_row_floats(store) floats */
    uint32_t *scratch;           /* splat_map_scratch_words(capacity) words */
} SplatPruneArgs;

/* floats per row over every non-NULL array of the store (the staging buffer of a prune holds capacity, rounded up to a
 * multiple of 4, times this) */
int32_t splat_map_row_floats(const SplatMapStore *store);

/* prune_gaussians' removal rule + remove_points (/root/reference/utils/slam_external.py:139-188): stable compaction of
 * the five parameter arrays, the Adam moments and the per-Gaussian variables.  counts[0], counts[1] are written. */
int splat_map_prune(SplatMapStore *store, const SplatPruneArgs *args, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Gradient-based densification (/root/reference/utils/slam_external.py:100-104, 191-240; on in
 * /root/reference/configs/replica/gaussian_splatting.py:82, off in the SLAM configs).
 * ------------------------------------------------------------------------------------------------------------ */

/* accumulate_mean2d_gradient for the iteration whose splat_iter_loss_backward has just run on `ws`: the COLOUR pass' own
 * dL/dmeans2D (what variables['means2D'].grad holds in the reference: the r, g, b planes of the loss gradient only, not the
 * depth render's) is formed by one more backward composite over the three colour planes, and for every Gaussian seen by the
 * render (radius > 0):  means2D_gradient_accum += |dL/dmeans2D.xy|,  denom += 1.  means2D_grad ([P][2], may be NULL)
 * receives the gradient itself; means2D_gradient_accum and denom may BOTH be NULL when only the gradient is wanted (a caller that
 * keeps the reference's own accumulate_mean2d_gradient statement: splatam_amd.plugin).  Leaves ws->accum zeroed. */
int splat_iter_means2d_accumulate(const SplatCamera *cam, const SplatMap *map, SplatIterWorkspace *ws,
                                  float *means2D_gradient_accum, float *denom, float *means2D_grad, void *stream);

enum { SPLAT_DENSIFY_CLONE = 0, SPLAT_DENSIFY_SPLIT = 1 };

typedef struct SplatDensifyArgs {
    int32_t mode;                /* CLONE: grads >= grad_thresh and max exp(log_scales) <= small_scale; SPLIT: ... > small_scale */
    float grad_thresh;           /* densify_dict['grad_thresh'] */
    float small_scale;           /* 0.01 * variables['scene_radius'] */
    int32_t rows_with_grad;      /* rows [0, rows_with_grad) carry an accumulated gradient (store->means2D_gradient_accum / denom,
                                    NaN -> 0); later rows -- the clones appended just before the split -- count as 0 (padded_grad) */
    int32_t num_to_split_into;   /* n (SPLIT) */
    const float *samples;        /* splat_map_duplicate, SPLIT: [S * n][3] draws of torch.normal(0, exp(log_scales)[to_split].repeat(n, 3)) */
    uint8_t *flags;              /* [capacity] selection flags: written by _select, read by _duplicate */
    uint32_t *scratch;           /* splat_map_scratch_words(capacity) words */
} SplatDensifyArgs;

/* Selection: flags + store->counts[1] = selected rows S, counts[0] = rows after appending S (CLONE) / S * n (SPLIT) rows,
 * counts[2] = the append would not fit `capacity`. */
int splat_map_densify_select(SplatMapStore *store, const SplatDensifyArgs *args, void *stream);

/* Appends the selected rows (after a _select with the same args): CLONE: copies, in row order; SPLIT: n blocks of the selected
 * rows (torch's v[mask].repeat(n, 1)) with means3D += build_rotation(unnorm_rotations) @ sample and
 * log_scales = log(exp(log_scales) / (0.8 n)).  Adam moments and max_2D_radius / means2D_gradient_accum / denom of the new rows
 * are zero.  The caller then sets map.P = counts[0]; the split originals are removed with splat_map_prune(to_remove = flags). */
int splat_map_duplicate(SplatMapStore *store, const SplatDensifyArgs *args, void *stream);

/* Developer switches used by scripts/ (never by the product path): key 0 = skip the per-tile count atomics of K1 (timing
 * experiment; results are then invalid); key 4 = measurement builds of the fused backward composite (bits: 1 = per-workgroup
 * wall-clock stamps into the buffer of splat_debug_stamps, 2 = stage the batches but visit nothing, 4 = no accumulator atomics,
 * 8 = phase 1 only; results are then invalid; 0 = the product kernel).  Returns the previous value, -1 for an unknown key. */
int splat_debug_option(int key, int value);
/* device buffer of 2 x (workgroups of the launch) int64 for splat_debug_option(4, 1): (start, end) of every workgroup in
 * wall_clock64() ticks (100 MHz); NULL switches it off. */
int splat_debug_stamps(void *buffer);

#ifdef __cplusplus
}
#endif
#endif /* SPLAT_HIP_H */
