// capi.hip -- the extern "C" surface declared in include/splat_hip.h.
#include <cstring>

#include "splat_device.h"

using namespace splat;

namespace {
int check(hipError_t e) { return e == hipSuccess ? SPLAT_OK : SPLAT_E_LAUNCH; }

bool valid_inputs(const SplatCamera *cam, const SplatGaussians *g) {
    if (!cam || !g) return false;
    if (g->P < 0 || cam->image_width <= 0 || cam->image_height <= 0) return false;
    if (cam->image_width > 65535 * SPLAT_TILE || cam->image_height > 65535 * SPLAT_TILE) return false;
    if (g->channels < 1 || g->channels > SPLAT_MAX_CHANNELS) return false;
    if (!cam->viewmatrix || !cam->projmatrix) return false;         // (bg == NULL: black)
    if (g->P > 0) {
        if (!g->means3D || !g->opacities) return false;
        if ((g->colors_precomp == nullptr) == (g->shs == nullptr)) return false;
        const bool sr = g->scales && g->rotations;
        if (sr == (g->cov3D_precomp != nullptr)) return false;
        if (g->shs) {
            if (g->channels != 3 || !cam->campos || cam->sh_degree < 0 || cam->sh_degree > 3) return false;
            if (g->sh_coeffs < (cam->sh_degree + 1) * (cam->sh_degree + 1)) return false;
        }
    }
    return true;
}
bool valid_state(const SplatGaussians *g, const SplatState *st, bool need_lists) {
    if (!st || !st->tile_count || !st->tile_base || !st->tile_cursor || !st->status) return false;
    if (g->P > 0 && (!st->depth || !st->xy || !st->conic_opacity || !st->rect || !st->radii)) return false;
    if (g->shs && g->P > 0 && (!st->rgb || !st->clamped)) return false;
    // (group binning: the composite builds the lists from the group records -- no key buckets)
    const bool groups = st->group_stride > 0 && st->group_count && st->group_recs && st->tile_stride > 0;
    if (need_lists && (st->capacity < 0 || (st->capacity > 0 && ((!st->keys && !groups) || !st->point_list)))) return false;
    if (st->tile_stride < 0 || st->group_stride < 0) return false;
    return true;
}
}  // namespace

extern "C" {

const char *splat_error_string(int code) {
    switch (code) {
        case SPLAT_OK: return "ok";
        case SPLAT_E_INVALID: return "invalid argument";
        case SPLAT_E_LAUNCH: return "HIP launch failed";
        case SPLAT_E_UNSUPPORTED: return "unsupported";
        default: return "unknown error";
    }
}

int splat_abi_version(void) { return SPLAT_ABI_VERSION; }

size_t splat_sizeof(const char *name) {
    if (!name) return 0;
#define SPLAT_SIZEOF_CASE(T) if (strcmp(name, #T) == 0) return sizeof(T)
    SPLAT_SIZEOF_CASE(SplatCamera); SPLAT_SIZEOF_CASE(SplatGaussians); SPLAT_SIZEOF_CASE(SplatState); SPLAT_SIZEOF_CASE(SplatGrads);
    SPLAT_SIZEOF_CASE(SplatMap); SPLAT_SIZEOF_CASE(SplatFrameData); SPLAT_SIZEOF_CASE(SplatLossConfig); SPLAT_SIZEOF_CASE(SplatIterWorkspace);
    SPLAT_SIZEOF_CASE(SplatAdamMap); SPLAT_SIZEOF_CASE(SplatPoseAdam); SPLAT_SIZEOF_CASE(SplatMapStore); SPLAT_SIZEOF_CASE(SplatAddArgs);
    SPLAT_SIZEOF_CASE(SplatPruneArgs); SPLAT_SIZEOF_CASE(SplatDensifyArgs); SPLAT_SIZEOF_CASE(SplatArrayInfo);
#undef SPLAT_SIZEOF_CASE
    return 0;
}

size_t splat_num_tiles(int32_t width, int32_t height) {
    if (width <= 0 || height <= 0) return 0;
    return (size_t)((width + SPLAT_TILE - 1) / SPLAT_TILE) * (size_t)((height + SPLAT_TILE - 1) / SPLAT_TILE);
}

// bucketed lists behind the reference API exist in ONE form: group binning with lists the composite sorts itself, three colour channels,
// buckets inside the capacity (include/splat_hip.h "GROUP BINNING behind the reference API"); anything else with tile_stride > 0 is refused
static bool valid_list_mode(const SplatCamera *cam, const SplatGaussians *g, const SplatState *st) {
    if (st->tile_stride == 0) return true;
    if (!group_binning(*st, cam->image_width, cam->image_height) || g->channels != 3 || g->shs) return false;
    return (long long)st->tile_stride * (long long)splat_num_tiles(cam->image_width, cam->image_height) <= st->capacity && st->point_list;
}

int splat_preprocess_forward(const SplatCamera *cam, const SplatGaussians *g, SplatState *st, void *stream) {
    if (!valid_inputs(cam, g) || !valid_state(g, st, false) || !valid_list_mode(cam, g, st)) return SPLAT_E_INVALID;
    return check(launch_preprocess_forward(*cam, *g, *st, (hipStream_t)stream));
}

int splat_bin_forward(const SplatCamera *cam, const SplatGaussians *g, SplatState *st, void *stream) {
    if (!valid_inputs(cam, g) || !valid_state(g, st, true)) return SPLAT_E_INVALID;
    // (splat_preprocess_forward of the same state counted per workgroup under the same predicate)
    return check(launch_bin_forward(*cam, *g, *st, (hipStream_t)stream, true, true));
}

int splat_render_forward(const SplatCamera *cam, const SplatGaussians *g, SplatState *st, float *out_color,
                         float *out_depth, void *stream) {
    if (!valid_inputs(cam, g) || !valid_state(g, st, true)) return SPLAT_E_INVALID;
    if (!out_color || !out_depth || !st->final_T || !st->n_contrib || !valid_list_mode(cam, g, st)) return SPLAT_E_INVALID;
    return check(launch_render_forward(*cam, *g, *st, out_color, out_depth, (hipStream_t)stream));
}

int splat_forward(const SplatCamera *cam, const SplatGaussians *g, SplatState *st, float *out_color, float *out_depth,
                  void *stream) {
    int rc = splat_preprocess_forward(cam, g, st, stream);
    if (rc) return rc;
    rc = splat_bin_forward(cam, g, st, stream);
    if (rc) return rc;
    return splat_render_forward(cam, g, st, out_color, out_depth, stream);
}

static bool valid_grads(const SplatGaussians *g, const SplatGrads *gr) {
    if (!gr || !gr->dL_dcolor) return false;
    if (g->P > 0) {
        if (!gr->accum || !gr->dL_dmeans3D || !gr->dL_dmeans2D || !gr->dL_dopacities) return false;
        if ((gr->dL_dscales == nullptr) != (gr->dL_drotations == nullptr)) return false;
        if (g->shs ? !gr->dL_dshs : !gr->dL_dcolors) return false;
    }
    return true;
}

int splat_render_backward(const SplatCamera *cam, const SplatGaussians *g, const SplatState *st, SplatGrads *gr,
                          void *stream) {
    if (!valid_inputs(cam, g) || !valid_state(g, st, true) || !valid_grads(g, gr)) return SPLAT_E_INVALID;
    if (!st->final_T || !st->n_contrib) return SPLAT_E_INVALID;
    return check(launch_render_backward(*cam, *g, *st, *gr, (hipStream_t)stream));
}

int splat_preprocess_backward(const SplatCamera *cam, const SplatGaussians *g, const SplatState *st, SplatGrads *gr,
                              void *stream) {
    if (!valid_inputs(cam, g) || !valid_state(g, st, false) || !valid_grads(g, gr)) return SPLAT_E_INVALID;
    return check(launch_preprocess_backward(*cam, *g, *st, *gr, (hipStream_t)stream));
}

int splat_backward(const SplatCamera *cam, const SplatGaussians *g, const SplatState *st, SplatGrads *gr, void *stream) {
    int rc = splat_render_backward(cam, g, st, gr, stream);
    if (rc) return rc;
    return splat_preprocess_backward(cam, g, st, gr, stream);
}

int splat_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, uint8_t *present, void *stream) {
    if (P < 0 || (P > 0 && (!means3D || !present)) || !viewmatrix) return SPLAT_E_INVALID;
    return check(launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream));
}

int splat_same_geometry(int32_t P, const float *opacities_a, const float *opacities_b, const float *scales_a, const float *scales_b,
                        const float *rotations_a, const float *rotations_b, int32_t *differ, void *stream) {
    if (P < 0 || !differ) return SPLAT_E_INVALID;
    if (P > 0 && (!opacities_a || !opacities_b || !scales_a || !scales_b || !rotations_a || !rotations_b)) return SPLAT_E_INVALID;
    return check(launch_same_geometry(P, opacities_a, opacities_b, scales_a, scales_b, rotations_a, rotations_b, differ, (hipStream_t)stream));
}

int splat_time_kernel(int fn, int iters, const SplatCamera *cam, const SplatGaussians *g, SplatState *st, SplatGrads *gr,
                      float *out_color, float *out_depth, void *stream, float *ms) {
    if (!ms || iters <= 0 || !valid_inputs(cam, g) || !valid_state(g, st, true)) return SPLAT_E_INVALID;
    if (fn == 1 && !valid_grads(g, gr)) return SPLAT_E_INVALID;
    if (fn != 0 && fn != 1) return SPLAT_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return SPLAT_E_LAUNCH;
    hipError_t err = hipSuccess;
    // the accumulator memset of the backward is part of launch_render_backward; time the kernel alone
    // by issuing the memsets first and recording around the launches only for fn == 0; for fn == 1 the
    // (tiny) memset is inside the bracket and documented as such.
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters && err == hipSuccess; ++i)
        err = fn == 0 ? launch_render_forward(*cam, *g, *st, out_color, out_depth, s) : launch_render_backward(*cam, *g, *st, *gr, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / iters;
    return check(err);
}

static int iter_loss_backward_impl(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame,
                                   const SplatLossConfig *cfg, SplatIterWorkspace *ws, const SplatPoseAdam *adam, void *stream,
                                   const SplatAdamMap *map_adam = nullptr) {
    if (!cam || !map || !frame || !cfg || !ws) return SPLAT_E_INVALID;
    if (map->P < 0 || cam->image_width <= 0 || cam->image_height <= 0 || !cam->viewmatrix || !cam->projmatrix) return SPLAT_E_INVALID;
    if (map->num_frames <= 0 || frame->time_idx < 0 || frame->time_idx >= map->num_frames) return SPLAT_E_INVALID;
    if (!map->cam_unnorm_rots || !map->cam_trans || !frame->im || !frame->depth || !frame->w2c) return SPLAT_E_INVALID;
    if (map->P > 0 && (!map->means3D || !map->rgb_colors || !map->unnorm_rotations || !map->logit_opacities || !map->log_scales))
        return SPLAT_E_INVALID;
    if (cfg->ignore_outlier_depth_loss && (!ws->outlier_err || !ws->outlier_scratch)) return SPLAT_E_INVALID;
    const SplatState &st = ws->st;
    if (!st.tile_count || !st.tile_base || !st.tile_cursor || !st.status || !st.final_T || !st.n_contrib) return SPLAT_E_INVALID;
    if (map->P > 0 && (!st.depth || !st.xy || !st.conic_opacity || !st.rect || !st.radii || !ws->feat8 || !ws->accum)) return SPLAT_E_INVALID;
    if (st.capacity <= 0 || !st.keys || !st.point_list || st.tile_stride < 0) return SPLAT_E_INVALID;
    if (st.tile_stride > 0 && (long long)st.tile_stride * (long long)splat_num_tiles(cam->image_width, cam->image_height) > st.capacity)
        return SPLAT_E_INVALID;
    if (!ws->out6 || !ws->dL_dout6 || !ws->sums || !ws->d_cam) return SPLAT_E_INVALID;
    if (!cfg->tracking && !ws->ssim_maps) return SPLAT_E_INVALID;
    if (st.tile_row_begin < 0 || st.tile_row_end < 0 || (st.tile_row_end > 0 && st.tile_row_end <= st.tile_row_begin) ||
        st.tile_row_end > (cam->image_height + SPLAT_TILE - 1) / SPLAT_TILE)
        return SPLAT_E_INVALID;
    if (st.tile_row_end > st.tile_row_begin && (!cfg->tracking || cfg->ignore_outlier_depth_loss || map_adam)) return SPLAT_E_INVALID;
    return check(launch_iter_loss_backward(*cam, *map, *frame, *cfg, *ws, (hipStream_t)stream, adam, map_adam));
}

int splat_iter_loss_backward(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame,
                             const SplatLossConfig *cfg, SplatIterWorkspace *ws, void *stream) {
    return iter_loss_backward_impl(cam, map, frame, cfg, ws, nullptr, stream);
}

int splat_iter_tracking_step(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame,
                             const SplatLossConfig *cfg, SplatIterWorkspace *ws, const SplatPoseAdam *adam, void *stream) {
    if (!cfg || !cfg->tracking || !cfg->camera_grad || !adam || !adam->state) return SPLAT_E_INVALID;
    return iter_loss_backward_impl(cam, map, frame, cfg, ws, adam, stream);
}

int splat_iter_finish(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame, const SplatLossConfig *cfg,
                      SplatIterWorkspace *ws, const SplatPoseAdam *adam, void *stream) {
    if (!cam || !map || !frame || !cfg || !ws || !ws->sums || !ws->d_cam || !ws->st.status) return SPLAT_E_INVALID;
    if (!map->cam_unnorm_rots || !map->cam_trans || map->num_frames <= 0 || frame->time_idx < 0 || frame->time_idx >= map->num_frames)
        return SPLAT_E_INVALID;
    if (adam && !adam->state) return SPLAT_E_INVALID;
    return check(launch_iter_finish(*cam, *map, *frame, *cfg, *ws, (hipStream_t)stream, adam));
}

int splat_iter_fold_sums(double *sums, void *stream) {
    if (!sums) return SPLAT_E_INVALID;
    return check(launch_iter_fold_sums(sums, (hipStream_t)stream));
}

int splat_iter_mapping_step(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame,
                            const SplatLossConfig *cfg, SplatIterWorkspace *ws, const SplatAdamMap *adam, void *stream) {
    if (!cfg || cfg->tracking || !cfg->gaussians_grad || !adam || !ws) return SPLAT_E_INVALID;
    // a stepped group (adam->grad[k] != NULL) takes the gradient this iteration forms in registers; adam->grad[k] names the buffer it is
    // ALSO written to -- or, when the workspace carries no buffer for the group (ws->d_* NULL), it is not stored at all (a loop that
    // discards its gradients after the step, as the reference's zero_grad(set_to_none=True) does: 48 bytes per Gaussian not written)
    const float *const grads[5] = {ws->d_means3D, ws->d_rgb_colors, ws->d_unnorm_rotations, ws->d_logit_opacities, ws->d_log_scales};
    for (int k = 0; k < 5; ++k)
        if (adam->grad[k] && ((grads[k] && adam->grad[k] != grads[k]) || !adam->exp_avg[k] || !adam->exp_avg_sq[k])) return SPLAT_E_INVALID;
    return iter_loss_backward_impl(cam, map, frame, cfg, ws, nullptr, stream, adam);
}

int splat_iter_adam_map(const SplatMap *map, const SplatAdamMap *opt, void *stream) {
    if (!map || !opt || map->P < 0) return SPLAT_E_INVALID;
    for (int k = 0; k < 5; ++k)
        if (opt->grad[k] && (!opt->exp_avg[k] || !opt->exp_avg_sq[k])) return SPLAT_E_INVALID;
    return check(launch_iter_adam_map(*map, *opt, (hipStream_t)stream));
}

int splat_iter_adam_pose(const SplatMap *map, int32_t time_idx, const float *d_cam, float *state, float beta1, float beta2,
                         float eps, float bc2_sqrt, float step_size_rot, float step_size_trans, void *stream) {
    if (!map || !d_cam || !state || !map->cam_unnorm_rots || !map->cam_trans) return SPLAT_E_INVALID;
    if (time_idx < 0 || time_idx >= map->num_frames) return SPLAT_E_INVALID;
    return check(launch_iter_adam_pose(*map, time_idx, d_cam, state, beta1, beta2, eps, bc2_sqrt, step_size_rot, step_size_trans,
                                       (hipStream_t)stream));
}

int splat_iter_time_kernel(int fn, int iters, const SplatCamera *cam, int32_t P, SplatIterWorkspace *ws, void *stream, float *ms) {
    if (!ms || iters <= 0 || !cam || !ws || fn < 0 || fn > 4 || P < 0) return SPLAT_E_INVALID;
    if (!ws->feat8 || !ws->out6 || !ws->dL_dout6 || !ws->accum || !ws->st.tile_base || !ws->st.point_list) return SPLAT_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return SPLAT_E_LAUNCH;
    hipError_t err = hipSuccess;
    // bucketed lists: the iteration's last kernel consumed and reset the tile counters and left the counts in the cursor words
    SplatState st = ws->st;
    if (st.tile_stride > 0) st.tile_count = st.tile_cursor;
    // the backward composite ALONE (fn 1, 4): the records in tile_recs belong to whatever forward composite ran last on this state
    // (the one-kernel tracking form writes none for a list's last batch): it gathers.  fn 3 alternates with the forward composite,
    // which writes them: the product's pair
    if (fn == 1 || fn == 4) st.tile_recs = nullptr;
    // fn 2: the forward composite in the form the iteration launches when the lists are short -- it filters its group's records (or
    // reads its bucket), sorts and publishes the tile's list itself.  The iteration's last kernel left the groups' record counts in
    // word 1 of their counter lines and the tiles' counts in the cursor words; the re-published lists and counts are the same values
    bool sort_form = false;
    // fn 3: forward and backward composite ALTERNATING, as the iteration issues them (the forward one in its sorting form when the
    // state allows it): time(fn 3) - time(fn 2 or 0) is the backward composite between other kernels -- 30 launches of it in a row
    // read ~10 % longer than rocprofv3's per-kernel average of the loop
    const bool can_sort = st.tile_stride > 0 && st.max_list_hint > 0 && st.max_list_hint + st.max_list_hint / 4 <= 1024;
    if (fn == 2 || (fn == 3 && can_sort)) {
        if (!can_sort) return SPLAT_E_INVALID;
        sort_form = true;
        if (st.group_stride > 0 && st.group_count && st.group_recs) st.group_count = st.group_count + 1;
        else st.group_stride = 0;
    } else {
        st.group_stride = 0;
    }
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters && err == hipSuccess; ++i) {
        if (fn != 1 && fn != 4) err = launch_render_forward_feat8(*cam, ws->feat8, st, ws->out6, sort_form, s);
        if ((fn == 1 || fn == 3) && err == hipSuccess) err = launch_render_backward_feat8(*cam, ws->feat8, st, ws->dL_dout6, ws->accum, P, false, true, s);
        if (fn == 4) err = launch_render_backward_feat8(*cam, ws->feat8, st, ws->dL_dout6, ws->accum, P, false, false, s, false);     // tracking form
    }
    (void)hipEventRecord(e1, s);
    // the timed backward launches accumulated into ws->accum: restore the workspace invariant (every iteration leaves the
    // accumulator zeroed; fused_backward_kernel relies on it) outside the timed bracket
    if ((fn == 1 || fn == 3 || fn == 4) && err == hipSuccess && P > 0) err = hipMemsetAsync(ws->accum, 0, sizeof(float) * SPLAT_GRAD_STRIDE * (size_t)P, s);
    (void)hipEventSynchronize(e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / iters;
    return check(err);
}

static bool valid_iter_common(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame, const SplatIterWorkspace *ws) {
    if (!cam || !map || !frame || !ws) return false;
    if (map->P < 0 || cam->image_width <= 0 || cam->image_height <= 0 || !cam->viewmatrix || !cam->projmatrix) return false;
    if (map->num_frames <= 0 || frame->time_idx < 0 || frame->time_idx >= map->num_frames) return false;
    if (!map->cam_unnorm_rots || !map->cam_trans || !frame->w2c) return false;
    if (map->P > 0 && (!map->means3D || !map->rgb_colors || !map->unnorm_rotations || !map->logit_opacities || !map->log_scales))
        return false;
    const SplatState &st = ws->st;
    if (!st.tile_count || !st.tile_base || !st.tile_cursor || !st.status || !st.final_T || !st.n_contrib) return false;
    if (map->P > 0 && (!st.depth || !st.xy || !st.conic_opacity || !st.rect || !st.radii || !ws->feat8)) return false;
    if (st.capacity <= 0 || !st.keys || !st.point_list || st.tile_stride < 0) return false;
    if (st.tile_stride > 0 && (long long)st.tile_stride * (long long)splat_num_tiles(cam->image_width, cam->image_height) > st.capacity)
        return false;
    return ws->out6 != nullptr;
}

int splat_iter_render(const SplatCamera *cam, const SplatMap *map, const SplatFrameData *frame, SplatIterWorkspace *ws, void *stream) {
    if (!valid_iter_common(cam, map, frame, ws)) return SPLAT_E_INVALID;
    return check(launch_iter_render(*cam, *map, *frame, *ws, (hipStream_t)stream));
}

size_t splat_map_scratch_words(int64_t n) { return map_scratch_words(n < 0 ? 0 : n); }

static bool valid_store(const SplatMapStore *st) {
    if (!st || !st->counts || st->capacity < 0 || st->map.P < 0 || st->map.P > st->capacity) return false;
    const SplatMap &m = st->map;
    if (st->capacity > 0 && (!m.means3D || !m.rgb_colors || !m.unnorm_rotations || !m.logit_opacities || !m.log_scales)) return false;
    for (int k = 0; k < 5; ++k)
        if ((st->exp_avg[k] == nullptr) != (st->exp_avg_sq[k] == nullptr)) return false;
    return true;
}

int32_t splat_map_row_floats(const SplatMapStore *store) {
    if (!store) return 0;
    return map_row_floats(*store);
}

int splat_map_add_new_gaussians(SplatMapStore *store, const SplatAddArgs *a, void *stream) {
    if (!valid_store(store) || !a) return SPLAT_E_INVALID;
    if (a->width < 0 || a->height < 0 || (long long)a->width * a->height > 0x7fffffffLL) return SPLAT_E_INVALID;
    if (a->mode != SPLAT_ADD_VALID_DEPTH && a->mode != SPLAT_ADD_NON_PRESENCE) return SPLAT_E_INVALID;
    if (!a->scratch || !a->im || !a->depth || a->fx == 0.f || a->fy == 0.f) return SPLAT_E_INVALID;
    if (a->mode == SPLAT_ADD_NON_PRESENCE) {
        if (!a->out6 || !a->err || !store->map.cam_unnorm_rots || !store->map.cam_trans) return SPLAT_E_INVALID;
        if (a->time_idx < 0 || a->time_idx >= store->map.num_frames) return SPLAT_E_INVALID;
    } else if (!a->w2c) {
        return SPLAT_E_INVALID;
    }
    return check(launch_map_add(*store, *a, (hipStream_t)stream));
}

int splat_map_prune(SplatMapStore *store, const SplatPruneArgs *a, void *stream) {
    if (!valid_store(store) || !a || !a->scratch) return SPLAT_E_INVALID;
    if (store->map.P > 0 && (!a->flags || !a->stage)) return SPLAT_E_INVALID;
    return check(launch_map_prune(*store, *a, (hipStream_t)stream));
}

int splat_iter_means2d_accumulate(const SplatCamera *cam, const SplatMap *map, SplatIterWorkspace *ws, float *gaccum, float *denom,
                                  float *means2D_grad, void *stream) {
    if (!cam || !map || !ws || map->P < 0 || cam->image_width <= 0 || cam->image_height <= 0) return SPLAT_E_INVALID;
    if (map->P > 0 && ((!gaccum != !denom) || (!gaccum && !means2D_grad) || !ws->accum || !ws->feat8 || !ws->dL_dout6 || !ws->st.radii || !ws->st.conic_opacity)) return SPLAT_E_INVALID;
    if (!ws->st.point_list || !ws->st.final_T || !ws->st.n_contrib || !ws->st.tile_base) return SPLAT_E_INVALID;
    return check(launch_iter_means2d_accumulate(*cam, *map, *ws, gaccum, denom, means2D_grad, (hipStream_t)stream));
}

static bool valid_densify(const SplatMapStore *store, const SplatDensifyArgs *a) {
    if (!valid_store(store) || !a || !a->flags || !a->scratch) return false;
    if (a->mode != SPLAT_DENSIFY_CLONE && a->mode != SPLAT_DENSIFY_SPLIT) return false;
    if (a->rows_with_grad < 0 || a->rows_with_grad > store->map.P) return false;
    if (a->rows_with_grad > 0 && (!store->means2D_gradient_accum || !store->denom)) return false;
    if (a->mode == SPLAT_DENSIFY_SPLIT && a->num_to_split_into < 1) return false;
    return true;
}

int splat_map_densify_select(SplatMapStore *store, const SplatDensifyArgs *a, void *stream) {
    if (!valid_densify(store, a)) return SPLAT_E_INVALID;
    return check(launch_map_densify_select(*store, *a, (hipStream_t)stream));
}

int splat_map_duplicate(SplatMapStore *store, const SplatDensifyArgs *a, void *stream) {
    if (!valid_densify(store, a)) return SPLAT_E_INVALID;
    if (a->mode == SPLAT_DENSIFY_SPLIT && !a->samples) return SPLAT_E_INVALID;
    return check(launch_map_duplicate(*store, *a, (hipStream_t)stream));
}

int splat_debug_option(int key, int value) {
    if (key == 0) { const int old = g_debug_skip_count; g_debug_skip_count = value; return old; }
    if (key == 4) { const int old = g_debug_k7_bits; g_debug_k7_bits = value; return old; }
    return -1;
}

int splat_debug_stamps(void *buffer) {
    g_debug_stamps = reinterpret_cast<long long *>(buffer);
    return SPLAT_OK;
}

// test hook (tests/test_gpu_primitives.py): see launch_selftest in binning.hip
int splat_selftest(int which, const void *in, void *out, int n, void *stream) {
    return check(launch_selftest(which, in, out, n, (hipStream_t)stream));
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// scratch layouts (include/splat_hip.h: "Scratch layouts")
// ---------------------------------------------------------------------------------------------------------
namespace {
struct LayoutWriter {
    SplatArrayInfo *out;
    int32_t max_entries, n = 0;
    size_t offset = 0;
    void add(const char *name, size_t bytes, int zero_init) {
        offset = (offset + SPLAT_SLAB_ALIGN - 1) / SPLAT_SLAB_ALIGN * SPLAT_SLAB_ALIGN;
        if (out && n < max_entries) out[n] = SplatArrayInfo{name, bytes, offset, zero_init};
        ++n;
        offset += bytes;
    }
};

// the arrays of a SplatState; `prefix`: the names as fields of SplatState ("") or of SplatIterWorkspace ("st.").  `iter`: the fused
// iteration's state (bucketed lists + group binning + launch order; its counters must start zeroed: no memset launches in the loop)
void state_arrays(LayoutWriter &w, bool iter, int32_t P, int32_t width, int32_t height, int32_t sub_bins, int64_t capacity,
                  int32_t group_stride, int32_t flags) {
    const size_t T = splat_num_tiles(width, height), S = sub_bins > 1 ? (size_t)sub_bins : 1, n = (size_t)P, cap = (size_t)capacity;
    const size_t G = (size_t)(((width + SPLAT_TILE - 1) / SPLAT_TILE + SPLAT_GROUP_TILES - 1) / SPLAT_GROUP_TILES) *
                     (size_t)(((height + SPLAT_TILE - 1) / SPLAT_TILE + SPLAT_GROUP_TILES - 1) / SPLAT_GROUP_TILES);
    const size_t HW = (size_t)width * (size_t)height;
    const int z = iter ? 1 : 0;
#define NAME(f) (iter ? "st." f : f)
    w.add(NAME("depth"), 4 * n, 0);
    w.add(NAME("xy"), 8 * n, 0);
    w.add(NAME("conic_opacity"), 16 * n, 0);
    w.add(NAME("rect"), 8 * n, 0);
    w.add(NAME("radii"), 4 * n, z);
    if (flags & SPLAT_LAYOUT_SH) {
        w.add(NAME("rgb"), 12 * n, 0);
        w.add(NAME("clamped"), 3 * n, 0);
    }
    w.add(NAME("tile_count"), 4 * T * S * SPLAT_COUNTER_STRIDE, z);
    w.add(NAME("tile_base"), 4 * (T + 1), 0);
    w.add(NAME("tile_cursor"), 4 * T * S * SPLAT_COUNTER_STRIDE, 0);
    if (iter || !(flags & SPLAT_LAYOUT_GROUPS)) w.add(NAME("keys"), 8 * cap, 0);
    w.add(NAME("point_list"), 4 * cap, 0);
    if (flags & SPLAT_LAYOUT_RECS) w.add(NAME("tile_recs"), 48 * cap, 0);
    if (flags & SPLAT_LAYOUT_LONG_LISTS) {
        w.add(NAME("keys_alt"), 8 * cap, 0);
        w.add(NAME("long_items"), 4 * (cap / 1024 + T + 1), 0);
    }
    w.add(NAME("long_base"), 4 * (T + 1), z);
    if (iter) {
        w.add("st.group_count", 4 * G * SPLAT_COUNTER_STRIDE, 1);
        if (group_stride > 0) w.add("st.group_recs", 16 * G * (size_t)group_stride, 0);
        if (flags & SPLAT_LAYOUT_TILE_ORDER) {
            w.add("st.tile_work", 4 * T, 1);
            w.add("st.tile_order", 4 * 8 * ((T + 7) / 8), 1);        // (zero = the natural order: SplatState.tile_order)
        }
    }
    w.add(NAME("final_T"), 4 * HW, 0);
    w.add(NAME("n_contrib"), 4 * HW, 0);
    if (!iter && (flags & SPLAT_LAYOUT_GROUPS)) {
        // group binning behind the reference API: the counters sit right in front of the status words (the library zeroes both with
        // one memset per call: launch_preprocess_forward)
        if (group_stride > 0) w.add("group_recs", 16 * G * (size_t)group_stride, 0);
        // (padded to the slab alignment: the status words then sit EXACTLY behind it, and the padding is the array's own)
        w.add("group_count", (4 * G * SPLAT_COUNTER_STRIDE + SPLAT_SLAB_ALIGN - 1) / SPLAT_SLAB_ALIGN * SPLAT_SLAB_ALIGN, 0);
    }
    w.add(NAME("status"), 4 * 4, z);
#undef NAME
}

bool layout_args_ok(int32_t P, int32_t width, int32_t height, int64_t capacity) {
    return P >= 0 && width > 0 && height > 0 && capacity >= 0 && width <= 65535 * SPLAT_TILE && height <= 65535 * SPLAT_TILE;
}
}  // namespace

extern "C" {

int splat_state_layout(int32_t P, int32_t width, int32_t height, int32_t sub_bins, int64_t capacity, int32_t flags,
                       SplatArrayInfo *out, int32_t max_entries, size_t *total_bytes) {
    if (!layout_args_ok(P, width, height, capacity) || sub_bins < 0 || (sub_bins & (sub_bins - 1)) != 0) return -SPLAT_E_INVALID;
    LayoutWriter w{out, max_entries};
    const size_t tiles = splat_num_tiles(width, height);
    const int64_t tile_stride = tiles ? capacity / (int64_t)tiles : 0;
    if ((flags & SPLAT_LAYOUT_GROUPS) && (tile_stride < 1 || tile_stride > (1 << 20) || sub_bins > 1)) return -SPLAT_E_INVALID;
    state_arrays(w, false, P, width, height, sub_bins, capacity, (flags & SPLAT_LAYOUT_GROUPS) ? (int32_t)(SPLAT_GROUP_TILES * SPLAT_GROUP_TILES * tile_stride) : 0, flags);
    if (flags & SPLAT_LAYOUT_BACKWARD) w.add("accum", 4 * (size_t)SPLAT_GRAD_STRIDE * (size_t)P, 0);
    if (total_bytes) *total_bytes = (w.offset + SPLAT_SLAB_ALIGN - 1) / SPLAT_SLAB_ALIGN * SPLAT_SLAB_ALIGN;
    return w.n;
}

size_t splat_workspace_bytes(int32_t P, int32_t width, int32_t height, int64_t capacity) {
    size_t total = 0;
    if (splat_state_layout(P, width, height, 1, capacity, SPLAT_LAYOUT_LONG_LISTS | SPLAT_LAYOUT_BACKWARD, nullptr, 0, &total) < 0) return 0;
    return total;
}

int splat_state_bind(SplatState *st, SplatGrads *gr, void *slab, const SplatArrayInfo *arrays, int32_t n, int32_t sub_bins,
                     int64_t capacity) {
    if (!st || !slab || !arrays || n < 0 || ((uintptr_t)slab % SPLAT_SLAB_ALIGN) != 0) return SPLAT_E_INVALID;
    char *base = static_cast<char *>(slab);
    for (int32_t i = 0; i < n; ++i) {
        const char *name = arrays[i].name;
        void *p = base + arrays[i].offset;
        if (!name) return SPLAT_E_INVALID;
        if (strncmp(name, "st.", 3) == 0) name += 3;
#define BIND(field, T) if (strcmp(name, #field) == 0) { st->field = static_cast<T>(p); continue; }
        BIND(depth, float *) BIND(xy, float *) BIND(conic_opacity, float *) BIND(rect, uint32_t *) BIND(radii, int32_t *)
        BIND(rgb, float *) BIND(clamped, uint8_t *) BIND(tile_count, uint32_t *) BIND(tile_base, uint32_t *) BIND(tile_cursor, uint32_t *)
        BIND(keys, uint64_t *) BIND(point_list, uint32_t *) BIND(tile_recs, float *) BIND(keys_alt, uint64_t *) BIND(long_base, uint32_t *) BIND(long_items, uint32_t *)
        BIND(group_count, uint32_t *) BIND(group_recs, uint32_t *) BIND(tile_work, uint32_t *) BIND(tile_order, uint32_t *)
        BIND(final_T, float *) BIND(n_contrib, int32_t *) BIND(status, int32_t *)
#undef BIND
        if (strcmp(name, "accum") == 0) { if (gr) gr->accum = static_cast<float *>(p); continue; }
    }
    st->capacity = capacity;
    st->sub_bins = sub_bins;
    return SPLAT_OK;
}

int splat_iter_workspace_layout(int32_t P, int32_t width, int32_t height, int64_t capacity, int32_t group_stride, int32_t flags,
                                SplatArrayInfo *out, int32_t max_entries, size_t *total_bytes) {
    if (!layout_args_ok(P, width, height, capacity) || group_stride < 0) return -SPLAT_E_INVALID;
    LayoutWriter w{out, max_entries};
    const size_t n = (size_t)P, HW = (size_t)width * (size_t)height;
    state_arrays(w, true, P, width, height, 1, capacity, group_stride, flags | SPLAT_LAYOUT_LONG_LISTS);
    w.add("feat8", 4 * 8 * n, 0);
    w.add("out6", 4 * 6 * HW, 0);
    w.add("dL_dout6", 4 * 6 * HW, 1);
    w.add("accum", 4 * (size_t)SPLAT_GRAD_STRIDE * n, 1);
    if (flags & SPLAT_LAYOUT_SSIM) w.add("ssim_maps", 4 * 9 * HW, 0);
    w.add("sums", 8 * (size_t)SPLAT_ITER_SUM_COPIES * SPLAT_ITER_SUMS, 1);
    w.add("d_cam", 4 * SPLAT_ITER_DCAM, 1);
    if (flags & SPLAT_LAYOUT_OUTLIER) {
        w.add("outlier_err", 4 * HW, 0);
        w.add("outlier_scratch", 4 * splat_map_scratch_words((int64_t)HW), 1);
    }
    if (total_bytes) *total_bytes = (w.offset + SPLAT_SLAB_ALIGN - 1) / SPLAT_SLAB_ALIGN * SPLAT_SLAB_ALIGN;
    return w.n;
}

size_t splat_iter_workspace_bytes(int32_t P, int32_t width, int32_t height, int64_t capacity, int32_t group_stride, int32_t flags) {
    size_t total = 0;
    if (splat_iter_workspace_layout(P, width, height, capacity, group_stride, flags, nullptr, 0, &total) < 0) return 0;
    return total;
}

int splat_iter_workspace_bind(SplatIterWorkspace *ws, void *slab, const SplatArrayInfo *arrays, int32_t n, int64_t capacity,
                              int32_t group_stride) {
    if (!ws) return SPLAT_E_INVALID;
    int rc = splat_state_bind(&ws->st, nullptr, slab, arrays, n, 1, capacity);
    if (rc) return rc;
    char *base = static_cast<char *>(slab);
    for (int32_t i = 0; i < n; ++i) {
        const char *name = arrays[i].name;
        void *p = base + arrays[i].offset;
#define BIND(field, T) if (strcmp(name, #field) == 0) { ws->field = static_cast<T>(p); continue; }
        BIND(feat8, float *) BIND(out6, float *) BIND(dL_dout6, float *) BIND(accum, float *) BIND(ssim_maps, float *)
        BIND(sums, double *) BIND(d_cam, float *) BIND(outlier_err, float *) BIND(outlier_scratch, uint32_t *)
#undef BIND
    }
    ws->st.group_stride = ws->st.group_recs ? group_stride : 0;
    return SPLAT_OK;
}

}  // extern "C"
