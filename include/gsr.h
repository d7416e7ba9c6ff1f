/*
 * gsr.h - C ABI of the MI355X-native differentiable Gaussian rasterizer (libgsr_hip.so).
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json:north_star.  It replaces the
 * native half of the third-party module the reference imports at
 *     /root/reference/gaussian_renderer/__init__.py:14
 *         from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
 * i.e. what `_C.rasterize_gaussians`, `_C.rasterize_gaussians_backward` and `_C.mark_visible` of
 * graphdeco-inria/diff-gaussian-rasterization@9c5c2028 (pin: reference results.md:2, .gitmodules:4-6) do for
 * the call at gaussian_renderer/__init__.py:90-109.  The source of that module is NOT in the reference tree
 * (SURVEY.md 0.1), so entry points cite the reference call site / contract they serve.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked "host";
 *   - all tensors are contiguous; fp32 unless stated; matrices are the reference's row-major 4x4 tensors
 *     (world_view_transform = W2C^T, full_proj_transform = (P.W2C)^T; reference scene/cameras.py:69-71);
 *   - `stream` is a hipStream_t passed as void*; every call enqueues on it and is re-entrant per stream;
 *   - the library never allocates device memory on the hot path: the caller owns three opaque state
 *     buffers (geometry / binning / image) that carry forward -> backward, sized by the gsr_*_bytes()
 *     queries (the reference's rasterizer keeps the same three buffers as torch uint8 tensors in its
 *     autograd ctx - SURVEY.md 8b "Ownership");
 *   - return value < 0 is an error code; gsr_last_error() gives the message (thread-local).
 */
#ifndef GSR_H_
#define GSR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 7: gsr_debug_mx_reduce (the compositing backward's sums on the matrix pipe: opt-in form GSR_BWD_REDUCE=mfma, measured slower);
 *    the image state carries the frame's walk classes (gsr_image_state_bytes grew; gsr_debug_walk_views); the backward's scratch
 *    carries validity flags of the gradient records (gsr_backward_scratch_bytes grew; gsr_debug_set_flags_min_r);
 * 6: host_status word 0 bit 0 = radix-sort look-back time-out (was reserved; debug = 1 fails the call), gsr_debug_wave_reduce_pk,
 *    gsr_forward_async_culled (host_status word 0 bit 1 / word 6 = a truncated tile list was too short);
 * 5: gsr_fused_adam.dynamic + gsr_adam_set_dynamic (optimizer factors in device memory, for HIP-graph replay), gsr_l1_mean_*;
 * 4: gsr_forward_async(num_rendered_out) / gsr_forward_rerender (verified speculation), gsr_sh_rank1_*; 3: gsr_backward_adam */
#define GSR_ABI_VERSION 7

enum {
  GSR_OK = 0,
  GSR_ERR_INVALID_ARGUMENT = -1, /* bad combination of inputs (both/neither of shs|colors_precomp ...) */
  GSR_ERR_HIP = -2,              /* a HIP runtime call failed */
  GSR_ERR_PREFILTERED_CULLED = -3, /* prefiltered=1 but a point failed the near-plane test */
  GSR_ERR_TOO_MANY_INSTANCES = -4, /* num_rendered (or the capacity) exceeds 2^30 - 1 */
  GSR_ERR_STATE_TOO_SMALL = -5   /* a caller-provided state buffer is smaller than required */
};

/* GaussianRasterizationSettings, reference gaussian_renderer/__init__.py:36-50 (13 fields). */
typedef struct gsr_settings {
  int32_t image_height;
  int32_t image_width;
  float tanfovx;
  float tanfovy;
  const float* bg;          /* [3]  */
  float scale_modifier;
  const float* viewmatrix;  /* [16] */
  const float* projmatrix;  /* [16] */
  int32_t sh_degree;        /* active degree, 0..3 */
  const float* campos;      /* [3]  */
  int32_t prefiltered;
  int32_t debug;            /* 1: synchronise and check after every kernel */
  int32_t antialiasing;
} gsr_settings;

/* Arguments of GaussianRasterizer.forward, reference gaussian_renderer/__init__.py:90-109.
 * Exactly one of {shs (+ optional dc), colors_precomp} and exactly one of {scales+rotations, cov3D_precomp}
 * must be non-NULL (same rule, same error, as the reference's rasterizer). */
typedef struct gsr_gaussians {
  int32_t P;                   /* number of Gaussians */
  int32_t sh_coeffs;           /* coefficients per channel stored in `shs` (row stride), e.g. 16 (or 15 with dc) */
  const float* means3D;        /* [P,3] */
  const float* dc;             /* [P,1,3] or NULL: SH band 0 kept separately (the `separate_sh` call form, :90-99) */
  const float* shs;            /* [P,sh_coeffs,3] or NULL */
  const float* colors_precomp; /* [P,3] or NULL */
  const float* opacities;      /* [P] */
  const float* scales;         /* [P,3] or NULL */
  const float* rotations;      /* [P,4] (w,x,y,z), used as given, or NULL */
  const float* cov3D_precomp;  /* [P,6] (xx,xy,xz,yy,yz,zz) or NULL */
  int32_t raw_activations;     /* 0: opacities / scales / rotations are activated values (the reference's call form).
                                * 1: they are the model's RAW parameters (scene/gaussian_model.py:55-57 _scaling, _rotation,
                                *    _opacity): exp / normalize / sigmoid (:38-46) are applied on load, and gsr_backward returns
                                *    the gradients w.r.t. the raw parameters - the model then needs no activation kernels in
                                *    the training step.  Ignored for cov3D_precomp. */
} gsr_gaussians;

/* Gradients returned by the backward, in the order the reference's autograd Function returns them
 * (SURVEY.md 8(a) a3).  Every non-NULL pointer is fully written (zeros for culled Gaussians). */
typedef struct gsr_grads {
  float* dL_dmeans3D;   /* [P,3] */
  float* dL_dmeans2D;   /* [P,3] screen-space gradient in NDC units, z = 0 (consumed by
                           scene/gaussian_model.py:431-433 add_densification_stats) */
  float* dL_ddc;        /* [P,1,3] or NULL */
  float* dL_dshs;       /* [P,sh_coeffs,3] or NULL (NULL although `shs` was given: allowed with `dc` - only dL_ddc is formed,
                           the view-sharded "sh_rank1" exchange rebuilds the rest from it) */
  float* dL_dcolors;    /* [P,3] or NULL (colors_precomp) */
  float* dL_dopacities; /* [P] */
  float* dL_dscales;    /* [P,3] or NULL */
  float* dL_drotations; /* [P,4] or NULL */
  float* dL_dcov3D;     /* [P,6] or NULL (cov3D_precomp) */
  /* Optional (all three or none): the backward also performs this view's add_densification_stats (reference
   * scene/gaussian_model.py:431-433) and max_radii2D update (train.py:159) - where radii > 0: xyz_gradient_accum +=
   * |dL_dmeans2D.xy|, denom += 1, max_radii2D = max(max_radii2D, radii) - the same arithmetic as gsr_densification_stats,
   * without the extra pass over P. */
  float* xyz_gradient_accum; /* [P,1] */
  float* denom;              /* [P,1] */
  float* max_radii2D;        /* [P]   */
} gsr_grads;

int gsr_abi_version(void);
const char* gsr_last_error(void);

/* State-buffer sizes (bytes).  Binning state depends on num_rendered, known after gsr_forward_prepare. */
size_t gsr_geometry_state_bytes(int32_t P);
size_t gsr_image_state_bytes(int32_t image_width, int32_t image_height);
size_t gsr_binning_state_bytes(int32_t P, int32_t image_width, int32_t image_height, int64_t num_rendered);
size_t gsr_backward_scratch_bytes(int32_t P, int64_t num_rendered);

/* Forward, phase 1: preprocess (projection, EWA covariance, SH->RGB), depth ordering and the tile-count
 * prefix sum.  Writes radii[P] (int32; 0 = culled; reference :118-121 `radii`, `visibility_filter`).
 * Waits once for num_rendered to arrive on the host (the reference rasterizer has the same single read-back); the
 * count is taken right after the projection kernel, so on return the depth sort / prefix sum may still be running on
 * `stream` - everything later is stream-ordered behind them.  Returns num_rendered >= 0 or an error code. */
int64_t gsr_forward_prepare(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state,
                            size_t geometry_bytes, int32_t* radii, void* stream);

/* Split form of phase 1 for the view-sharded data-parallel trainer: gsr_forward_prepare_geometry does everything
 * gsr_forward_prepare does EXCEPT the SH -> RGB evaluation (it does not read `dc` / `shs`); gsr_forward_shade fills the colours
 * and must run before gsr_forward_render.  Between the two calls the caller may wait for the SH coefficients of this step
 * (81 % of the gradient bytes) to finish their all-reduce + Adam update on another stream.  Results are bitwise identical
 * to the fused gsr_forward_prepare. */
int64_t gsr_forward_prepare_geometry(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state,
                                     size_t geometry_bytes, int32_t* radii, void* stream);
int gsr_forward_shade(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, void* stream);

/* Forward, phase 2: instance emission, tile sort, tile ranges and 16x16-tile alpha compositing.
 * Writes out_color[3,H,W] and out_invdepth[1,H,W] (reference :90,:101 `rendered_image`, `depth_image`).
 * `for_backward` != 0 additionally records what gsr_backward needs in the state buffers (the emission slot of every list
 * position rides through the tile sort); 0 = forward-only render, a gsr_backward on these buffers is then undefined. */
int gsr_forward_render(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state,
                       void* binning_state, size_t binning_bytes, int64_t num_rendered, void* image_state,
                       size_t image_bytes, float* out_color, float* out_invdepth, int32_t for_backward,
                       void* stream);

/* gsr_forward_prepare_geometry's companion that evaluates the colours as LATE as possible: instance emission, tile sort and
 * tile ranges run first (none of them reads `dc` / `shs`), then `stream` waits for `sh_ready_event` (a hipEvent_t recorded by
 * the caller after the SH coefficients' update; NULL = no wait), then the SH -> RGB pass (what gsr_forward_shade does), then
 * the compositing: the whole binning stage overlaps an SH exchange / update running on another stream.  Same results. */
int gsr_forward_render_shade(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state,
                             void* binning_state, size_t binning_bytes, int64_t num_rendered, void* image_state,
                             size_t image_bytes, float* out_color, float* out_invdepth, int32_t for_backward,
                             void* sh_ready_event, void* stream);

/* Speculative forward (phase 1 + phase 2 in ONE call, sized by the caller's estimate): for callers that keep grow-only state
 * buffers (a SLAM / training loop).  The reference's rasterizer reads num_rendered back in the middle of every forward to size
 * its binning buffer (SURVEY.md 2.3 "D2H num_rendered"; call site gaussian_renderer/__init__.py:90-109), which leaves the device
 * idle while the host allocates and enqueues the rest.  Here the binning state is sized by the caller for `capacity` instances
 * (gsr_binning_state_bytes(.., capacity)), the WHOLE frame is enqueued, num_rendered stays on the device and every later stage
 * reads min(num_rendered, capacity) from there.  Two ways to use it:
 *
 *   verified (num_rendered_out != NULL; what diff_gaussian_rasterization does by default): after everything is enqueued the
 *     call waits until the count has reached the host - the kernel that knows it stores it straight into pinned memory, with
 *     the binning and compositing stages still queued behind it, so the device does not idle - and returns it in
 *     *num_rendered_out.  If it exceeds `capacity` the frame just enqueued was composited from a TRUNCATED instance list: the
 *     caller grows its binning state and calls gsr_forward_rerender, which repeats phase 2 exactly; outputs and state are then
 *     those of the blocking pair, bit for bit.  Every frame is exact.
 *   unverified (num_rendered_out == NULL): no wait at all.  A frame beyond `capacity` loses the surplus - emitted last: its
 *     farthest splats with tile_local_sort = 0, the Gaussians with the highest indices with tile_local_sort = 1 - never an
 *     out-of-bounds access; the caller learns it later from `host_status`.  gsr_backward / gsr_backward_adam on such a frame
 *     are NO-OPS by construction (every backward kernel reads the count): zero gradients, no optimizer update, no statistics.
 *
 *   host_status: NULL or 8 words of host memory, filled asynchronously on `stream` at the END of the call's work:
 *                [0] bit 0 = a radix-sort look-back wait timed out on the device (broken inter-workgroup hand-off: the frame is
 *                mis-sorted; with debug = 1 the call itself fails with GSR_ERR_HIP), [1] bit 0 = a prefiltered point failed the near-plane test, [2],[3] = num_rendered (lo, hi),
 *                [4] = longest tile list of the frame if it exceeds 2048 entries, else 0 (tile_local_sort only).
 *                Pinned memory (hipHostMalloc / torch pin_memory) is written by the compositing kernel itself through its
 *                device mapping; any other host memory gets a hipMemcpyAsync.  Read it after an event recorded behind this
 *                call has completed.
 *   tile_local_sort: 0 = the binning of the blocking path (global depth sort of the Gaussians, emission in depth order, stable
 *                tile sort).  1 = no global depth order: emission in index order, the same stable tile sort, then every
 *                tile orders ITS list by (depth bits, id) in LDS (binning.hip, k_tile_depth_sort) - identical lists, about
 *                0.07 ms less per frame at 1 M Gaussians / 1080p (the tile counts are then also scanned inside the projection
 *                and emission kernels: four launches fewer); lists longer than 4096 entries take a slow in-memory path, so a
 *                caller should fall back to 0 when host_status[4] approaches that (diff_gaussian_rasterization/_workspace.py).
 *   defer_color / sh_ready_event: as gsr_forward_prepare_geometry + gsr_forward_render_shade (0 / NULL: fused colour pass).
 * The matching gsr_backward takes `capacity` as its num_rendered.  Same kernels, same results as the blocking pair whenever
 * num_rendered <= capacity. */
int gsr_forward_async(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, size_t geometry_bytes,
                      int32_t* radii, void* binning_state, size_t binning_bytes, int64_t capacity, void* image_state,
                      size_t image_bytes, float* out_color, float* out_invdepth, int32_t for_backward,
                      int32_t defer_color, void* sh_ready_event, uint32_t* host_status, int32_t tile_local_sort,
                      void* stream, int64_t* num_rendered_out /* host, or NULL */);

/* gsr_forward_async with TILE LISTS TRUNCATED BY DEPTH (round 4).  tile_depth_cutoff: `tiles` uint32 words owned by the caller, one
 * array per VIEW it renders repeatedly (a training set's cameras), initialised to 0xFFFFFFFF.  Every call UPDATES it: the
 * compositing kernel leaves, per tile, the depth bits of the list entry at 1.75 x (+ 48) the position of the deepest one any pixel
 * of the tile needed before it saturated (T < 1e-4) - 0xFFFFFFFF if a pixel never saturated.  With apply != 0 (honoured only for an
 * unverified frame, num_rendered_out == NULL, in the tile-local binning form, debug off) the array is also USED: a (tile, Gaussian)
 * instance whose depth lies behind its tile's cut-off is neither counted nor emitted - the tile's depth-ordered list loses its
 * tail.  If every pixel of a truncated tile still saturates inside its list, image, depth, final_T, n_contrib and all gradients
 * are those of the untruncated frame bit for bit (the loop never reached the missing tail).  If one does not, the frame flags
 * itself: status word 0 bit 1 on the device - gsr_backward / gsr_backward_adam are then NO-OPS, as for a frame beyond the capacity -
 * and word 6 of host_status (clear it before the call; final once an event recorded behind the call has completed); the caller
 * renders that view again with apply = 0.  Where a scene saturates early (dense captures) this removes most of the R-proportional
 * work of the step: emission, both tile-sort passes, per-tile ordering, the gradient-record gather. */
int gsr_forward_async_culled(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, size_t geometry_bytes,
                             int32_t* radii, void* binning_state, size_t binning_bytes, int64_t capacity, void* image_state,
                             size_t image_bytes, float* out_color, float* out_invdepth, int32_t for_backward,
                             int32_t defer_color, void* sh_ready_event, uint32_t* host_status, int32_t tile_local_sort,
                             void* stream, int64_t* num_rendered_out, uint32_t* tile_depth_cutoff, int32_t apply);
/* Phase 2 once more on the state a gsr_forward_async call with the same `s`, `g`, geometry / image state and tile_local_sort
 * left behind, for a (larger) binning state of `capacity` >= the count that call reported: instance emission, tile sort,
 * ranges, per-tile ordering, compositing.  Stream-ordered behind the first attempt; overwrites its outputs. */
int gsr_forward_rerender(const gsr_settings* s, const gsr_gaussians* g, void* geometry_state, void* binning_state,
                         size_t binning_bytes, int64_t capacity, void* image_state, size_t image_bytes, float* out_color,
                         float* out_invdepth, int32_t for_backward, int32_t tile_local_sort, uint32_t* host_status,
                         void* stream);

/* Backward of the calls above.  dL_dinvdepth may be NULL (treated as zero).  `num_rendered`: the value gsr_forward_prepare
 * returned (and gsr_forward_render was given), or the `capacity` given to gsr_forward_async. */
int gsr_backward(const gsr_settings* s, const gsr_gaussians* g, const int32_t* radii,
                 const void* geometry_state, const void* binning_state, const void* image_state,
                 int64_t num_rendered, const float* dL_dcolor, const float* dL_dinvdepth,
                 void* scratch, size_t scratch_bytes, const gsr_grads* grads, void* stream);

/* gsr_backward with the optimizer step folded in (single-GPU training step: reference train.py:139 loss.backward() followed
 * by :170-179 optimizer.step(), when nothing sits between the two - no gradient exchange, no accumulation over views, no
 * densification at this iteration).  The six parameter groups of reference scene/gaussian_model.py:160-168 are updated IN PLACE
 * by the backward's last kernel from the gradients it holds in registers / LDS; those gradients (59 floats per Gaussian at SH
 * degree 3) are never written to memory, and no separate optimizer kernel re-reads them.
 *   Requirements: raw_activations = 1 (opacities / scales / rotations are the model's raw parameters), `dc` and `shs` passed
 *   separately (the separate_sh call form), no colors_precomp / cov3D_precomp.  The arrays of `g` are the parameters themselves
 *   and are written to (the `const` of gsr_gaussians does not hold for this call).
 *   Group order of the arrays below: 0 xyz (g->means3D), 1 f_dc (g->dc), 2 f_rest (g->shs), 3 opacity, 4 scaling, 5 rotation.
 *   sparse = 0: torch.optim.Adam semantics (bias correction; `step` = 1-based step number AFTER this update), every row.
 *   sparse = 1: SparseGaussianAdam.step(radii > 0, P) semantics (reference train.py:173-176): rows with radii == 0 untouched,
 *               no bias correction.
 *   sparse = 2: as 0, but only the rows that have tile instances in this forward are updated here; the others (their gradient
 *               is exactly zero) must get their update from gsr_adam_step_culled_rows with the same `opt` values - a call
 *               that needs nothing from the backward and can therefore run on another stream while the compositing kernels
 *               (bound by VALU issue, not by HBM) are busy.  The pair equals sparse = 0 bit for bit.
 * grads->dL_dmeans2D is still written (densification statistics, scene/gaussian_model.py:431-433); the other members of `grads`
 * are ignored.  Same arithmetic as gsr_backward followed by gsr_adam_step / gsr_sparse_adam_step, bit for bit. */
typedef struct gsr_fused_adam {
  float* exp_avg[6];
  float* exp_avg_sq[6];
  float lr[6];
  int64_t step[6];
  double beta1, beta2, eps;
  int32_t sparse;
  /* Optional DEVICE pointer to GSR_ADAM_DYNAMIC_FLOATS floats (lr[6], lr / bias_correction1 [6], 1 / sqrt(bias_correction2) [6]),
   * or NULL.  When set, the kernels read the per-step factors from there instead of deriving them from `lr` / `step`: the call
   * then carries no per-step constant in its launch arguments and can be captured once into a HIP graph and replayed, with
   * gsr_adam_set_dynamic enqueued in front of every replay. */
  const float* dynamic;
} gsr_fused_adam;
#define GSR_ADAM_DYNAMIC_FLOATS 18
/* Computes the factors of `opt` (its lr / step / betas / sparse, exactly as gsr_backward_adam would) and enqueues a one-workgroup
 * kernel that stores them to `dynamic_dev`: the values travel as launch arguments, so the host may call it again at once. */
int gsr_adam_set_dynamic(const gsr_fused_adam* opt, float* dynamic_dev, void* stream);
int gsr_backward_adam(const gsr_settings* s, const gsr_gaussians* g, const int32_t* radii, const void* geometry_state,
                      const void* binning_state, const void* image_state, int64_t num_rendered, const float* dL_dcolor,
                      const float* dL_dinvdepth, void* scratch, size_t scratch_bytes, const gsr_grads* grads,
                      const gsr_fused_adam* opt, void* stream);

/* Dense Adam update (zero gradient: moments decay, the parameter follows its momentum) of the rows that reached no tile in the
 * forward whose geometry state is given; companion of gsr_backward_adam(opt->sparse = 2).  May be enqueued on any stream once
 * the forward call that filled `geometry_state` has been enqueued and that stream waits for it; the caller orders it before the
 * next forward.  `g`, `num_rendered` and `opt` as for gsr_backward_adam (same `step` values). */
int gsr_adam_step_culled_rows(const gsr_gaussians* g, const void* geometry_state, int64_t num_rendered,
                              const gsr_fused_adam* opt, void* stream);

/* GaussianRasterizer.markVisible (near-plane test).  present[P] uint8. */
int gsr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, uint8_t* present, void* stream);

/* Introspection for tests / bench: DEVICE pointers into the opaque state buffers (valid while the buffer lives). */
/* rec48: the packed 48-B splat records, 12 floats per Gaussian = (mean2D.xy, conic A' B') (conic C', opacity, cut-off, r)
 * (g, b, 1/depth, depth); clamped: one byte per Gaussian, bit c set = colour channel c was clamped at 0 (any out pointer may
 * be NULL) */
int gsr_debug_geometry_views(const void* geometry_state, int32_t P, const float** rec48, const uint32_t** depth_keys_sorted,
                             const uint32_t** order, const uint32_t** tiles_touched, const uint16_t** rect,
                             const uint32_t** offsets, const uint8_t** clamped);
/* test hook: 64-lane sums through the render backward's cross-lane reductions; in[10][64] -> out[20]:
 * out[0..9] = the ten-value tree, out[10..18] = the nine-value tree on rows 0..8, out[19] unused */
int gsr_debug_wave_reduce(const float* in640, float* out20, void* stream);
/* the same sums through the packed-pair trees (v_pk_add_f32 behind the swap stages) of k_render_bwd_tile; same layout */
int gsr_debug_wave_reduce_pk(const float* in640, float* out20, void* stream);
/* test hook: the matrix-pipe form of the same reduction (k_render_bwd_tile_mx: v_mfma_f32_16x16x4_f32 against the tile's pixel basis).
 * in[514] = h[4][64] (sub-block s = 0..3, lane: the pixel's dL/dopacity_eff), c[4][64] (the lane's channel sums), mu[2] (the 2-D mean
 * relative to the tile centre); out[10] = the gradient record's ten values (sum h dx, sum h dy, sum h dx^2, sum h dx dy, sum h dy^2,
 * sum h, c0..c3) with d = mu - pixel, pixel (x, y) of (s, lane) = ((lane & 7) + 8 (s & 1) - 7.5, (lane >> 3) + 8 (s >> 1) - 7.5) */
int gsr_debug_mx_reduce(const float* in514, float* out10, void* stream);
int gsr_debug_binning_views(const void* binning_state, int32_t image_width, int32_t image_height,
                            int64_t num_rendered, const uint32_t** point_list, const uint32_t** ranges);
/* Pair evaluations of the compositing forward (SURVEY.md 8(d) "FLOP model"): pairs[2*H*W] (uint32) = per pixel, the number of
 * list entries evaluated while the pixel was still compositing [0, H*W) and the number of entries it blended [H*W, 2*H*W),
 * counted by an instrumented build of the forward kernel on the state buffers of a finished forward.  (The backward's count is
 * the sum of n_contrib: it replays entries 1..n_contrib.) */
int gsr_debug_count_pairs(const gsr_settings* s, int32_t P, const void* geometry_state, const void* binning_state,
                          int64_t num_rendered, uint32_t* pairs, void* stream);
/* test / measurement hook: the library's stable LSD radix sort (sort_scan.hip) on caller-provided ping-pong buffers: keys k0
 * (input) / k1, values v0 / v1 (vals_iota != 0: value = index, v0 is not read), optional second payload w0 / w1 (both or
 * neither), key bits [0, bits); n_dev: optional device pointer to a 64-bit count (the kernels then sort min(*n_dev, n) keys).
 * tmp: gsr_debug_radix_tmp_bytes(n) bytes.  Returns 0 / 1 = the buffer set holding the result, or an error code. */
size_t gsr_debug_radix_tmp_bytes(int64_t n);
int gsr_debug_radix_sort(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, uint32_t* w0, uint32_t* w1, int64_t n,
                         int32_t bits, int32_t vals_iota, const uint32_t* n_dev, void* tmp, void* stream);
int gsr_debug_image_views(const void* image_state, int32_t image_width, int32_t image_height,
                          const float** final_T, const uint32_t** n_contrib);
/* The walk classes a forward with a backward to follow leaves in the image state (ABI 7): walk_cnt[classes] = tiles per class,
 * walk_list[classes][tiles] = the tiles of each class in the order their compositing workgroups finished, walk_of_tile[tiles] = the
 * deepest contributor of any pixel of the tile = the number of list entries its backward walks; class = exponent and two leading
 * mantissa bits of that number (walks below 4: the number itself; clamped at 65535).  k_render_bwd_tile takes the classes
 * longest first (GSR_BWD_LPT=0: index order).  Returns the number of classes (64). */
/* From how many tile instances on a frame's per-instance gradient records carry validity flags (one byte per emission slot behind
 * the records of the backward's scratch buffer: an instance behind its tile's walk gets no all-zero record, and the projection
 * backward reads none).  Default 2 500 000 (smaller frames keep the zero records: there the flags are one more link in a
 * latency-bound kernel).  Tests force either form: 0 = always, 2^32 - 1 = never; negative: only report.  Returns the previous value. */
int64_t gsr_debug_set_flags_min_r(int64_t min_instances);
int gsr_debug_walk_views(const void* image_state, int32_t image_width, int32_t image_height, const uint32_t** walk_cnt,
                         const uint32_t** walk_list, const uint32_t** walk_of_tile);

/* ---- callers of the hot path that the reference also takes from native modules ---- */

/* Fused SSIM map, 11x11 Gaussian window sigma 1.5, zero "same" padding (what reference utils/loss_utils.py:100-159
 * computes; serves `_C.fusedssim` of utils/loss_utils.py:16-38 and `fused_ssim.fused_ssim` of train.py:31-35,116-117).
 * img*, maps: [planes,H,W].  The three dm_* maps (all NULL or all non-NULL) are what the backward needs. */
int gsr_fused_ssim_forward(int32_t planes, int32_t H, int32_t W, float C1, float C2, const float* img1,
                           const float* img2, float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq,
                           float* dm_dsigma12, void* stream);
int gsr_fused_ssim_backward(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2,
                            const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq,
                            const float* dm_dsigma12, float* dL_dimg1, void* stream);

/* Fused training loss of reference train.py:114-121, (1-l)*mean|a-b| + l*(1-mean(ssim)) (SURVEY.md 8(f) f3 "Fused L1 + SSIM
 * loss"): forward writes the dm_* maps and per-block partial sums partials[2*gsr_fused_loss_blocks()] = (sum ssim, sum |a-b|);
 * a one-workgroup finalize adds them in a fixed order into the device scalar `loss`; backward reads dL/dloss from the device. */
int64_t gsr_fused_loss_blocks(int32_t planes, int32_t H, int32_t W);
int gsr_fused_l1_ssim_forward(int32_t planes, int32_t H, int32_t W, float C1, float C2, float lambda_dssim,
                              const float* img1, const float* img2, float* dm_dmu1, float* dm_dsigma1_sq,
                              float* dm_dsigma12, float* partials, float* loss /*device scalar*/, void* stream);
int gsr_fused_l1_ssim_backward(int32_t planes, int32_t H, int32_t W, float lambda_dssim, const float* img1,
                               const float* img2, const float* upstream /*device scalar dL/dloss or NULL*/,
                               const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                               float* dL_dimg1, void* stream);

/* weight * mean|(a - b) mask| over n floats and its gradient w.r.t. a: the inverse-depth regularisation term of a training
 * step (reference train.py:124-132, `Ll1depth`; `mask` = depth_mask, may be NULL), three launches instead of torch's dozen.
 * a, b, mask 16-byte aligned; partials: gsr_l1_mean_blocks() floats of scratch; out / upstream: DEVICE scalars (upstream NULL
 * = 1).  Deterministic. */
int32_t gsr_l1_mean_blocks(void);
int gsr_l1_mean_forward(int64_t n, float weight, const float* a, const float* b, const float* mask, float* partials, float* out,
                        void* stream);
int gsr_l1_mean_backward(int64_t n, float weight, const float* a, const float* b, const float* mask, const float* upstream,
                         float* grad, void* stream);

/* One-launch Adam over up to 8 tensors.  Dense = torch.optim.Adam semantics (reference scene/gaussian_model.py:169-170
 * default optimizer); sparse = `SparseGaussianAdam.step(visibility, N)` (reference train.py:37-41,173-176): rows of
 * invisible Gaussians untouched, no bias correction.  Array arguments are HOST arrays of `count` entries; betas / eps are
 * doubles so that 1-beta is formed in double like torch does (1.f-0.999f is off by 1.3e-5 relative). */
int gsr_adam_step(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const int64_t* numel, const float* lr, const int64_t* step, double beta1,
                  double beta2, double eps, void* stream);
int gsr_sparse_adam_step(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                         float* const* exp_avg_sq, const int64_t* numel, const float* lr, int64_t N,
                         const uint8_t* visible, double beta1, double beta2, double eps, void* stream);

/* GaussianModel.densify_and_prune (reference scene/gaussian_model.py:367-429 + optimizer surgery :274-344) as two passes.
 * plan: decides keep / clone / split per Gaussian with the reference's predicates (NaN grads -> 0; clone: small & grad >= thr;
 * split: large & grad >= thr; prune: sigmoid(opacity) < min_opacity, or world size > 0.1*extent when use_world_size_prune),
 * runs the prefix sums, and returns counts_host[3] = {kept originals, clones, split sources} (one stream sync).
 * apply: writes the new arrays in the reference's order [kept originals][clones][children copy 0][children copy 1];
 * in_ptrs / out_ptrs are HOST arrays of 18 device pointers: (xyz, f_dc, f_rest, opacity, scaling, rotation) x
 * (value, exp_avg, exp_avg_sq); moments may be NULL.  New size = keep + clone + 2*child. */
/* add_densification_stats (reference scene/gaussian_model.py:431-433) + max_radii2D update (train.py:159), one pass:
 * where radii > 0: accum += ||grad_means2D.xy||, denom += 1, max_radii2D = max(max_radii2D, radii). */
int gsr_densification_stats(int64_t P, const float* grad_means2D, const int32_t* radii, float* xyz_gradient_accum,
                            float* denom, float* max_radii2D, void* stream);
size_t gsr_densify_workspace_bytes(int64_t P);
int gsr_densify_plan(int64_t P, const float* xyz_gradient_accum, const float* denom, const float* scaling_raw,
                     const float* opacity_raw, float max_grad, float min_opacity, float extent, float percent_dense,
                     int32_t use_world_size_prune, void* workspace, size_t workspace_bytes, int64_t* counts_host,
                     void* stream);
int gsr_densify_apply(int64_t P, const void* workspace, const float* const* in_ptrs, float* const* out_ptrs,
                      const int32_t* row_floats, int64_t n_keep, int64_t n_clone, int64_t n_child, uint32_t seed,
                      int32_t* source_of_row, void* stream);

/* The model's parameter activations, fused (reference scene/gaussian_model.py:38-46 setup_functions, :101-121 getters):
 * scaling = exp(_scaling) [P,3], rotation = normalize(_rotation) = x / max(|x|, 1e-12) [P,4], opacity = sigmoid(_opacity)
 * [P,1]; what render() reads through pc.get_scaling / get_rotation / get_opacity (gaussian_renderer/__init__.py:55-62).
 * One launch each way instead of ~25 elementwise / reduce launches per training step.
 * backward: dL_d* of the three outputs (any may be NULL = zero) -> gradients of the raw parameters (always written). */
int gsr_gaussian_activations_forward(int32_t P, const float* raw_scaling, const float* raw_rotation,
                                     const float* raw_opacity, float* scaling, float* rotation, float* opacity,
                                     void* stream);
int gsr_gaussian_activations_backward(int32_t P, const float* raw_rotation, const float* scaling, const float* opacity,
                                      const float* dL_dscaling, const float* dL_drotation, const float* dL_dopacity,
                                      float* dL_draw_scaling, float* dL_draw_rotation, float* dL_draw_opacity,
                                      void* stream);

/* View-sharded data parallelism (the reference itself is single-GPU): with ONE view per rank per step the SH
 * gradient of a rank is rank one per Gaussian, dL/dsh[k][c] = basis_k(dir) * dL/drgb_c, and dL/df_dc = C0 * dL/drgb carries it
 * whole.  The ranks all-gather `gathered`[n_ranks][P + 1][3]: rows 0..P-1 = that rank's dL/df_dc, row P = its camera centre
 * (12 B per Gaussian and rank instead of 192 B all-reduced), and this call rebuilds the MEAN gradients of f_dc [P, 1, 3] and
 * f_rest [P, sh_coeffs_rest, 3] (scale = 1 / n_ranks), summing the ranks in order: identical bits on every rank.  `means3D` are
 * the positions the forwards saw.  Coefficients beyond the active degree get zeros. */
int gsr_sh_rank1_expand(int32_t P, int32_t n_ranks, int32_t sh_degree, int32_t sh_coeffs_rest, const float* means3D,
                        const float* gathered, float scale, float* dL_ddc_mean, float* dL_dsh_rest_mean, void* stream);
/* gsr_sh_rank1_expand followed by the dense Adam update (torch.optim.Adam semantics) of f_dc and f_rest, in ONE kernel: the
 * rebuilt gradients are never written to memory.  `opt`: groups 1 (f_dc) and 2 (f_rest) of a gsr_fused_adam are used (moments,
 * lr, 1-based step AFTER this update; sparse must be 0).  Bit-identical to gsr_sh_rank1_expand + gsr_adam_step on those tensors. */
int gsr_sh_rank1_adam(int32_t P, int32_t n_ranks, int32_t sh_degree, int32_t sh_coeffs_rest, const float* means3D,
                      const float* gathered, float scale, float* f_dc, float* f_rest, const gsr_fused_adam* opt, void* stream);

/* Per-kernel timing with HIP events on the launch stream (used by bench.py's roofline block).  A measurement aid, process-
 * global and meant for ONE host thread driving the library at a time: enabling it while several host threads launch
 * concurrently attributes times correctly per thread (the open event pair is thread-local) but adds a lock to every launch. */
void gsr_profile_enable(int32_t on);
void gsr_profile_reset(void);
/* Fills up to `max` entries; returns the number of distinct kernels.  `names` receives pointers to
 * static strings. */
int32_t gsr_profile_read(const char** names, double* total_ms, int64_t* calls, int32_t max);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H_ */
