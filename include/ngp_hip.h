/*
 * ngp_hip.h — C ABI of libngp_hip.so, the MI355X (gfx950) instant-NGP hot path.
 *
 * This is the drop-in boundary for the reference's native extension surface:
 *   - `vren` (pybind11 module, /root/reference/models/csrc/binding.cpp:323-342,
 *     prototypes in models/csrc/include/utils.h:10-170), and
 *   - the tiny-cuda-nn objects the reference's field uses
 *     (models/networks.py:40-163: Grid/Hash encoding, SphericalHarmonics,
 *     CutlassMLP), plus the optimizer step the trainer runs (train.py:244).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major data unless a
 *     parameter is explicitly documented as host memory;
 *   - the caller owns every buffer (allocated on the stream it passes in);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - return 0 on success, a negative NGP_E* code on failure; never throws,
 *     never allocates, never synchronises the device, no hidden global state
 *     (safe to capture into a hipGraph);
 *   - all floating point is fp32, indices are int32/int64 as the reference's;
 *   - empty batches (n == 0) are valid and return NGP_OK before any pointer is looked at;
 *   - the product build reads NO environment variable.  Switches for A/B timing (NGP_ADAM_BLOCKS, NGP_WGRAD_BLOCKS,
 *     NGP_MLP_NO_STREAM, NGP_MLP_NO_STREAM_WGRAD, NGP_MLP_STREAM_HEADS, NGP_ADAM_DENSE_ZERO, NGP_GRID_GATHER_OLD) and the
 *     superseded kernel variants (NGP_GRID_BWD_SIMPLE / _NOPAIR / _NOSLIDE, NGP_MARCH_LANE_PER_RAY, ...) exist only in the
 *     A/B build (-DNGP_AB_VARIANTS, `NGP_AB_VARIANTS=1 python -m instant-ngp-pp_amd.build`).
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef NGP_HIP_H
#define NGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGP_OK 0
#define NGP_EINVAL (-22)   /* bad argument (null pointer, unsupported size) */
#define NGP_ELAUNCH (-5)   /* hipLaunch reported an error */

#define NGP_MAX_LEVELS 32

/* Activation codes shared by the MLP entry points. */
enum ngp_activation {
    NGP_ACT_NONE = 0,
    NGP_ACT_RELU = 1,
    NGP_ACT_SIGMOID = 2,
    NGP_ACT_SOFTPLUS = 3, /* torch.nn.Softplus(beta=1, threshold=20), models/networks.py:56,59 */
    NGP_ACT_EXP = 4
};

/* library / build identification: returns a static string "ngp_hip <ver> gfx950". */
const char* ngp_version(void);

/* hex digest of the sources (csrc + this header + compiler flags) the library was built from; the Python
 * binding refuses a library whose digest differs from the header it derives its prototypes from. */
const char* ngp_build_id(void);

/* ------------------------------------------------------------------------
 * R1  ray / AABB and ray / sphere intersection
 * replaces vren.ray_aabb_intersect  (binding.cpp:4-16,  intersection.cu:25-100)
 *          vren.ray_sphere_intersect (binding.cpp:19-31, intersection.cu:124-197)
 * outputs: hit_cnt (n_rays) i32 = number of voxels hit (not capped),
 *          hits_t (n_rays,max_hits,2) f32, hits_idx (n_rays,max_hits) i64,
 *          both sorted ascending by t1 with unused slots (-1) first, exactly as
 *          the reference's torch::sort + gather leaves them.
 * ---------------------------------------------------------------------- */
int ngp_ray_aabb_intersect(const float* rays_o, const float* rays_d,
                           const float* centers, const float* half_sizes,
                           int n_rays, int n_voxels, int max_hits,
                           int32_t* hit_cnt, float* hits_t, int64_t* hits_idx,
                           void* stream);

int ngp_ray_sphere_intersect(const float* rays_o, const float* rays_d,
                             const float* centers, const float* radii,
                             int n_rays, int n_spheres, int max_hits,
                             int32_t* hit_cnt, float* hits_t, int64_t* hits_idx,
                             void* stream);

/* render() pre-amble (models/rendering.py:30): t1 in [0,near) -> near, in place on
 * hits_t (n_rays, max_hits, 2) slot 0. */
int ngp_clamp_near(float* hits_t, int n_rays, int max_hits, float near_distance, void* stream);

/* ------------------------------------------------------------------------
 * O1  occupancy-grid helpers
 * replaces vren.morton3D / morton3D_invert / packbits
 *          (binding.cpp:74-101, raymarching.cu:35-161)
 * ---------------------------------------------------------------------- */
int ngp_morton3D(const int32_t* coords, int n, int32_t* indices, void* stream);
int ngp_morton3D_invert(const int32_t* indices, int n, int32_t* coords, void* stream);
/* threshold_dev (device scalar) overrides `threshold` when non-NULL: lets the caller keep
 * min(mean_density, density_threshold) (networks.py:405-407) on the device, with no .item() sync. */
int ngp_packbits(const float* density_grid, int n_bytes, float threshold, const float* threshold_dev,
                 uint8_t* density_bitfield, void* stream);

/* NGP.update_density_grid's point generation + EMA, fused (models/networks.py:388-403):
 * xyzs_w[i] = (coords[i]/(G-1)*2-1)*(s-s/G) + (noise[i]*2-1)*s/G  (noise in [0,1))   */
int ngp_grid_cell_points(const int32_t* coords, const float* noise, int n, int grid_size,
                         float s, float* xyzs_w, void* stream);
/* The sampled branch of NGP.update_density_grid, fused (networks.py:308-333 + 388-398): for ONE cascade, m uniformly
 * random cells + m cells drawn uniformly from the occupied ones (density_grid_c > density_threshold; none
 * occupied: the m uniform ones only), in Morton bucket order, with their jittered world points
 * x_w = (coord/(G-1)*2-1)*(s-s/G) + U(-1,1)*s/G.  Randomness is a counter-based hash of (seed, sample, draw): the
 * same on every rank, independent of launch order.  workspace: ngp_grid_sample_workspace(G, m) int32 elements
 * (host-only query).  outputs: indices (2m) i32 Morton cell indices, xyzs_w (2m,3) f32. */
int64_t ngp_grid_sample_workspace(int grid_size, int m);
int ngp_grid_sample_cells(const float* density_grid_c, int grid_size, float density_threshold, int m,
                          int64_t seed, float s, int32_t* workspace, int32_t* indices, float* xyzs_w,
                          void* stream);
/* density_grid_tmp[c, indices] = sigmas (networks.py:398); several samples in one cell: the largest wins */
int ngp_density_grid_scatter_max(float* density_grid_tmp_c, const int32_t* indices, const float* sigmas, int n,
                                 void* stream);
/* EMA as ngp_density_grid_ema, plus threshold_out[0] = min(mean of the positive cells, density_threshold) and
 * threshold_out[1] = that mean (networks.py:405-407), kept on the device for ngp_packbits; partials: 1024 floats */
int ngp_density_grid_ema_threshold(float* density_grid, const float* density_grid_tmp, int n, float decay,
                                   float density_threshold, float* partials, float* threshold_out, void* stream);
/* grid = grid<0 ? grid : max(grid*decay, tmp)  (networks.py:400-403), n = K*G^3 */
int ngp_density_grid_ema(float* density_grid, const float* density_grid_tmp, int n,
                         float decay, void* stream);

/* ------------------------------------------------------------------------
 * R3  training ray marcher
 * replaces vren.raymarching_train (binding.cpp:104-131, raymarching.cu:166-332)
 *
 * Three launches: DDA count pass (one lane per ray, sample t's parked in
 * `t_scratch`), single-workgroup exclusive scan (writes rays_a and counter),
 * wave-per-ray expansion (coalesced xyzs/dirs/deltas/ts writes).
 * rays_a rows are in ray order with start indices monotone (a legal instance of
 * the reference's atomic order, raymarching.cu:237-241).
 *
 * t_scratch: (n_rays*max_samples) f32 workspace; ray_counts: (n_rays) i32 workspace.
 * xyzs/dirs: capacity `sample_capacity` rows (the reference allocates
 * n_rays*max_samples); rows >= counter[0] are left untouched unless zero_tail!=0,
 * in which case they are zero-filled like the reference's torch::zeros.
 * counter: (2) i32 -> {total samples, n_rays}.
 * ---------------------------------------------------------------------- */
int ngp_raymarching_train(const float* rays_o, const float* rays_d, const float* hits_t /* (n_rays,2) */,
                          const uint8_t* density_bitfield, int cascades, float scale,
                          float exp_step_factor, const float* noise, int grid_size,
                          int max_samples, int n_rays,
                          float* t_scratch, int32_t* ray_counts,
                          int64_t* rays_a, float* xyzs, float* dirs, float* deltas, float* ts,
                          int32_t* counter, int64_t sample_capacity, int zero_tail,
                          void* stream);

/* ------------------------------------------------------------------------
 * T1  test-time marcher / compositor
 * replaces vren.raymarching_test (binding.cpp:134-163, raymarching.cu:335-454)
 *          vren.composite_test_fw (binding.cpp:262-320, volumerendering.cu:314-423)
 * hits_t (n_rays_total,2) is updated in place; outputs are (n_alive,N_samples,.)
 * and must be zero-initialised by the caller (reference: torch::zeros).
 * ---------------------------------------------------------------------- */
int ngp_raymarching_test(const float* rays_o, const float* rays_d, float* hits_t,
                         const int64_t* alive_indices, const uint8_t* density_bitfield,
                         int cascades, float scale, float exp_step_factor, int grid_size,
                         int max_samples, int n_samples, int n_alive,
                         float* xyzs, float* dirs, float* deltas, float* ts,
                         int32_t* n_eff_samples, void* stream);

int ngp_composite_test_fw(const float* sigmas, const float* rgbs, const float* normals,
                          const float* normals_raw, const float* sems, const float* deltas,
                          const float* ts, const float* hits_t, int64_t* alive_indices,
                          float T_threshold, int classes, const int32_t* n_eff_samples,
                          int n_alive, int n_samples,
                          float* opacity, float* depth, float* rgb, float* normal,
                          float* normal_raw, float* sem, void* stream);

/* Device-driven rounds of the same loop (rendering.py:46-133): the number of alive rays, the samples per ray of the round
 * (N_samples = max(min(N_rays // N_alive, 64), min_samples)), the running `samples` sum of the loop head and the
 * total sample count live in a device record, so the host enqueues round after round without reading anything back.
 *   state int32[8]: [0] n_alive (caller sets it to n_rays, the rest to 0)  [1] N_samples of the round
 *                   [2] sum of N_samples so far  [3] done (no alive ray left, or the sum reached max_samples_total)
 *                   [4] rounds begun  [5] n_alive behind the last compaction (adopted by the next round's head)
 *                   [6:8] int64 total samples marched
 * ngp_test_round_begin      evaluates the loop head into state;
 * ngp_raymarching_test_rounds / ngp_composite_test_fw_rounds  = the two kernels above with (n_alive, n_samples) read from
 *                           state; launched for n_alive_bound >= n_alive rays (a stale host copy of the count is a
 *                           valid bound: it only decreases); outputs laid out (n_alive, N_samples, .) as above;
 * ngp_alive_compact         alive_out = alive_in[alive_in >= 0] in order (rendering.py:115), state[5] = its length;
 *                           block_counts: scratch of ceil(n_alive_bound / 1024) int32.
 * All of them do nothing once state[3] is set.  The caller zero-fills xyzs / dirs per round (padding rows are evaluated by
 * the field) up to its own bound on n_alive * N_samples. */
int ngp_test_round_begin(int32_t* state, int n_rays, int min_samples, int max_samples_total, void* stream);
int ngp_raymarching_test_rounds(const float* rays_o, const float* rays_d, float* hits_t, const int64_t* alive_indices,
                                const uint8_t* density_bitfield, int cascades, float scale, float exp_step_factor,
                                int grid_size, int max_samples, int32_t* state, int n_alive_bound, float* xyzs,
                                float* dirs, float* deltas, float* ts, int32_t* n_eff_samples, void* stream);
int ngp_composite_test_fw_rounds(const float* sigmas, const float* rgbs, const float* normals, const float* normals_raw,
                                 const float* sems, const float* deltas, const float* ts, int64_t* alive_indices,
                                 float T_threshold, int classes, const int32_t* n_eff_samples, const int32_t* state,
                                 int n_alive_bound, float* opacity, float* depth, float* rgb, float* normal,
                                 float* normal_raw, float* sem, void* stream);
int ngp_alive_compact(const int64_t* alive_in, int32_t* state, int n_alive_bound, int32_t* block_counts,
                      int64_t* alive_out, void* stream);

/* ------------------------------------------------------------------------
 * V1/V2  training compositor
 * replaces vren.composite_alpha_fw (binding.cpp:166-180, volumerendering.cu:5-63)
 *          vren.composite_train_fw (binding.cpp:183-208, volumerendering.cu:65-164)
 *          vren.composite_train_bw (binding.cpp:211-259, volumerendering.cu:167-311)
 * One 32-lane half-wave per rays_a row, segmented transmittance scan, early stop
 * at T <= T_threshold.  Every per-sample output row of the ray is written (zeros
 * past the stop), so callers need not pre-zero per-sample outputs; per-ray
 * outputs are written for every rays_a row.  In ngp_composite_train_bw the outputs
 * dL_dnormals_pred / dL_dsems may be NULL (together with their upstream gradients)
 * when the loss does not use the composited normal / semantic maps, and any of
 * the upstream gradients dL_dopacity / dL_ddepth / dL_drgb / dL_dws may be NULL,
 * which reads as all zeros.
 * ---------------------------------------------------------------------- */
int ngp_composite_alpha_fw(const float* sigmas, const float* deltas, const int64_t* rays_a,
                           float T_threshold, int n_rays, float* alphas, float* ws, void* stream);

int ngp_composite_train_fw(const float* sigmas, const float* rgbs, const float* normals_pred,
                           const float* sems, const float* deltas, const float* ts,
                           const int64_t* rays_a, float T_threshold, int classes, int n_rays,
                           int64_t* total_samples, float* opacity, float* depth, float* rgb,
                           float* normal_pred, float* sem, float* ws, void* stream);

int ngp_composite_train_bw(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb,
                           const float* dL_dnormal_pred, const float* dL_dsem, const float* dL_dws,
                           const float* sigmas, const float* rgbs, const float* normals_pred,
                           const float* ws, const float* deltas, const float* ts,
                           const int64_t* rays_a, const float* opacity, const float* depth,
                           const float* rgb, const float* normal_pred,
                           float T_threshold, int classes, int n_rays,
                           float* dL_dsigmas, float* dL_drgbs, float* dL_dnormals_pred,
                           float* dL_dsems, void* stream);

/* ------------------------------------------------------------------------
 * V3  Ref-NeRF normal regularisers
 * replaces vren.composite_refloss_fw/bw (binding.cpp, ref_loss.cu:4-175)
 * ---------------------------------------------------------------------- */
int ngp_composite_refloss_fw(const float* sigmas, const float* normals_diff, const float* normals_ori,
                             const float* deltas, const float* ts, const int64_t* rays_a,
                             float T_threshold, int n_rays, float* loss_o, float* loss_p,
                             void* stream);

int ngp_composite_refloss_bw(const float* dL_dloss_o, const float* dL_dloss_p,
                             const float* sigmas, const float* normals_diff, const float* normals_ori,
                             const float* deltas, const float* ts, const int64_t* rays_a,
                             const float* loss_o, const float* loss_p, float T_threshold, int n_rays,
                             float* dL_dsigmas, float* dL_dnormals_diff, float* dL_dnormals_ori,
                             void* stream);

/* ------------------------------------------------------------------------
 * D1  distortion loss
 * replaces vren.distortion_loss_fw/bw (binding.cpp, losses.cu:8-173)
 * ---------------------------------------------------------------------- */
int ngp_distortion_loss_fw(const float* ws, const float* deltas, const float* ts,
                           const int64_t* rays_a, int n_rays,
                           float* loss, float* ws_inclusive_scan, float* wts_inclusive_scan,
                           void* stream);

int ngp_distortion_loss_bw(const float* dL_dloss, const float* ws_inclusive_scan,
                           const float* wts_inclusive_scan, const float* ws, const float* deltas,
                           const float* ts, const int64_t* rays_a, int n_rays,
                           float* dL_dws, void* stream);

/* ------------------------------------------------------------------------
 * L1  loss glue, fused (each replaces a chain of torch elementwise ops in the reference)
 * ngp_nerf_loss      : NeRFLoss default terms reduced as train.py:307 does (sum of term means),
 *                      losses.py:96-105: terms[1] += mean (rgb-gt)^2, terms[2] += lambda_o mean(-o log o)
 *                      (o = opacity+1e-10), terms[3] += lambda_d mean(distortion) (per-ray distortion
 *                      loss of ngp_distortion_loss_fw, may be NULL), terms[0] += their sum; and the
 *                      gradients d_rgb = 2(rgb-gt)/(3n), d_opacity = lambda_o(-log o - 1)/n.
 * ngp_refloss_inputs : normals_diff = (n_raw-n_pred)^2, normals_ori = max(<n_raw, normalize(dir)>,0)^2
 *                      (rendering.py:243-245).
 * ngp_neg_normalize  : y = -F.normalize(x*scale3, eps=1e-6) on rows of 3 (networks.py:210,215), and
 *                      its backward.
 * ---------------------------------------------------------------------- */
int ngp_nerf_loss(const float* rgb, const float* target_rgb, const float* opacity,
                  const float* distortion, int n_rays, float lambda_opacity, float lambda_distortion,
                  float* terms /* (4), caller zeroes */, float* d_rgb, float* d_opacity, void* stream);
/* Default training recipe, everything between the field's raw outputs and the field's backward in ONE launch
 * (replaces, with the same arithmetic per ray: -F.normalize of d sigma/dx (scaled by scale3, device (3) or NULL) and of
 * the normal head, networks.py:198,215; softmax of the class logits, networks.py:219; composite_train_fw,
 * volumerendering.cu:65-164; the RefLoss inputs and forward, rendering.py:243-249 + ref_loss.cu; distortion_loss_fw/bw,
 * losses.cu; NeRFLoss's rgb / opacity / distortion terms reduced as sum(term.mean()), losses.py:96-105, train.py:307;
 * composite_train_bw, volumerendering.cu:167-311).  Outputs: the per-ray results of render() (total_samples int64,
 * opacity, depth, rgb, normal_pred, semantic, Ro = loss_o, Rp = loss_p (n_rays,3)), ws (N), vr_samples (1) int64 =
 * sum of total_samples, terms (4) = [loss, rgb, opacity, distortion], and the loss's gradients w.r.t. the field's
 * outputs dL_dsigmas (N), dL_drgbs (N,3).  terms and vr_samples are cleared here.  rays_a must cover every sample row;
 * classes <= 8; normal_head / sem_logits rows may be strided (ld_*). */
int ngp_render_loss_fused(const float* sigmas, const float* rgbs, const float* dsigma_dx, const float* scale3,
                          const float* normal_head, int64_t ld_normal, const float* sem_logits, int64_t ld_sem,
                          const float* dirs, const float* deltas, const float* ts, const int64_t* rays_a,
                          const float* target_rgb, const float* rgb_bg /* device (3) or NULL: rgb += bg (1 - opacity),
                          rendering.py:236-241 */, float T_threshold, int classes, int n_rays, float lambda_opacity,
                          float lambda_distortion, int64_t* total_samples, int64_t* vr_samples, float* opacity,
                          float* depth, float* rgb, float* normal_pred, float* sem, float* ws, float* loss_o,
                          float* loss_p, float* terms, float* dL_dsigmas, float* dL_drgbs, void* stream);
int ngp_refloss_inputs(const float* normals_raw, const float* normals_pred, const float* dirs, int64_t n,
                       float* normals_diff, float* normals_ori, void* stream);
int ngp_neg_normalize(const float* x, int64_t ldx, const float* scale3 /* device (3) or NULL */, int64_t n,
                      float* y, void* stream);
int ngp_neg_normalize_bwd(const float* x, int64_t ldx, const float* dL_dy, int64_t n, float* dL_dx,
                          void* stream);

/* ------------------------------------------------------------------------
 * Live samples of a marched batch (no counterpart in the reference: it evaluates the colour branch on every
 * sample and lets composite_train_fw, volumerendering.cu:84-114, ignore those behind the early-termination
 * point).  ngp_live_rows decides, with the compositing kernels' own transmittance bookkeeping, which samples of
 * every ray take part (all up to and including the one at which T <= T_threshold) and writes
 *   offsets  (n_rays)  scratch: first position of the ray's live samples in the list
 *   live_idx (N)       the first *n_live entries: ascending sample rows that take part
 *   inv_idx  (N)       position of a sample in that list, -1 for a sample behind its ray's stop
 *   n_live   (1)       device word (copy it to the host to size the compacted launches)
 * rays_a must cover every sample row (the marcher's output does).  Up to two dense (N,3) row blocks (positions,
 * directions) are moved into the compacted order on the way.  ngp_gather_rows / ngp_spread_rows(3) move (n, cols)
 * row blocks into / out of the compacted order (spread writes zeros to rows that are not in the list; the
 * three-block form takes dense blocks, leading dimension = cols).
 * ---------------------------------------------------------------------- */
int ngp_live_rows(const float* sigmas, const float* deltas, const int64_t* rays_a, float T_threshold,
                  int64_t n_rays, int32_t* offsets, int32_t* live_idx, int32_t* inv_idx, int32_t* n_live,
                  const float* a3 /* (N,3) or NULL */, float* a3_compact, const float* b3 /* (N,3) or NULL */,
                  float* b3_compact, void* stream);
int ngp_spread_rows3(const float* src0, int cols0, float* dst0, const float* src1, int cols1, float* dst1,
                     const float* src2, int cols2, float* dst2, const int32_t* inv_idx, int64_t n, void* stream);
int ngp_gather_rows(const float* src, int64_t ld_src, int cols, const int32_t* idx, int64_t n_out, float* dst,
                    int64_t ld_dst, void* stream);
int ngp_spread_rows(const float* src, int64_t ld_src, int cols, const int32_t* inv_idx, int64_t n, float* dst,
                    int64_t ld_dst, void* stream);

/* ------------------------------------------------------------------------
 * R4  torch_scatter.segment_csr(src, indptr) with sum reduction
 * (custom_functions.py:110-112).  src (n_rows, width), indptr (n_seg+1) i64.
 * ---------------------------------------------------------------------- */
int ngp_segment_csr_sum(const float* src, const int64_t* indptr, int n_seg, int width,
                        float* out, void* stream);

/* ------------------------------------------------------------------------
 * H1-H4  multiresolution hash grid (tcnn.Encoding otype Grid/HashGrid,
 * models/networks.py:40-52,67-76; semantics SURVEY.md Appendix B)
 * 3-D inputs in [0,1], linear interpolation, F in {1,2,4,8}.
 * ---------------------------------------------------------------------- */
typedef struct ngp_grid_desc {
    uint32_t n_levels;
    uint32_t n_features;            /* F */
    uint32_t offsets[NGP_MAX_LEVELS + 1]; /* in table rows (F floats each) */
    uint32_t resolution[NGP_MAX_LEVELS];
    float    scale[NGP_MAX_LEVELS];
} ngp_grid_desc;

/* Host-side helper: fills `desc` from the tcnn config keys; returns the number of
 * parameters (floats) of the table, or a negative error. */
int64_t ngp_grid_layout(int n_levels, int n_features, int log2_hashmap_size,
                        int base_resolution, double per_level_scale, ngp_grid_desc* desc);

/* y (n, L*F) = encode(x (n,3)); ldy / lddy = row stride in floats of y / dL_dy (>= L*F), so the
 * encoder can write into (and its backward read from) a column block of a wider matrix such as
 * rgb_net's [SH | grid features | appearance code] input (networks.py:229-231) with no concat. */
int ngp_grid_fwd(const ngp_grid_desc* desc /* host */, const float* table, const float* x,
                 int64_t n, float* y, int64_t ldy, void* stream);

/* dtable[(off+idx)*F+f] += w * dL_dy  (atomic fp32 scatter-add; caller zeroes dtable) */
int ngp_grid_bwd_param(const ngp_grid_desc* desc, const float* x, const float* dL_dy, int64_t lddy,
                       int64_t n, float* dtable, void* stream);
/* Same, with every sample's gradient row multiplied by row_scale[sample] (NULL = 1) as it is loaded:
 * dParam += w * row_scale[s] * dL_dy[s].  The density head has ONE output, so the gradient it sends into its
 * encoder is a per-sample multiple of d(sigma)/d(features) — which the forward pass already computed for the
 * analytic normals (networks.py:186-196): the backward needs no second data-gradient product. */
int ngp_grid_bwd_param_scaled(const ngp_grid_desc* desc, const float* x, const float* dL_dy, int64_t lddy,
                              const float* row_scale, int64_t n, float* dtable, void* stream);

/* dL_dx (n,3) = sum_l d enc_l / dx . dL_dy_l */
int ngp_grid_bwd_input(const ngp_grid_desc* desc, const float* table, const float* x,
                       const float* dL_dy, int64_t lddy, int64_t n, float* dL_dx, void* stream);

/* double backward of ngp_grid_bwd_input: given dL_ddLdx (n,3) (gradient flowing into
 * dL_dx), accumulates dtable (atomic) and writes dL_ddLdy (n, L*F).  Either output
 * may be NULL. */
int ngp_grid_bwd_bwd_input(const ngp_grid_desc* desc, const float* table, const float* x,
                           const float* dL_dy, int64_t lddy, const float* dL_ddLdx, int64_t n,
                           float* dtable, float* dL_ddLdy, void* stream);

/* ------------------------------------------------------------------------
 * H5  spherical harmonics (tcnn.Encoding otype SphericalHarmonics, degree 1..4,
 * networks.py:78-85,128-135).  x (n,3) in [0,1] -> y (n, degree^2).
 * ---------------------------------------------------------------------- */
int ngp_sh_fwd(const float* x, int64_t n, int degree, float* y, int64_t ldy, void* stream);

/* same basis of a raw view direction d (n,3): y = SH((normalize(d, eps=1e-6) + 1) / 2), the three
 * steps of networks.py:198,222 (F.normalize, remap to [0,1], dir_encoder) in one launch */
int ngp_sh_fwd_dirs(const float* d, int64_t n, int degree, float* y, int64_t ldy, void* stream);
int ngp_sh_bwd_input(const float* x, const float* dL_dy, int64_t n, int degree,
                     float* dL_dx, void* stream);

/* ------------------------------------------------------------------------
 * M1-M4  small MLP layers on f32 MFMA (v_mfma_f32_32x32x2_f32)
 * replaces tcnn.Network(CutlassMLP) (networks.py:89-163) and the torch xyz_net
 * (networks.py:54-58).  Weights are (n_out, n_in) row-major (tcnn / nn.Linear).
 *
 * ngp_linear_fwd : y (n, n_out) = act(x (n, n_in) . W^T + b)      b may be NULL
 * ngp_linear_bwd_input : dx (n, n_in) = dz (n, n_out) . W
 * ngp_linear_bwd_weight: dW (n_out, n_in) += dz^T . x ; db (n_out) += sum dz
 *                        (atomic split-K over samples; caller zeroes dW/db)
 * ngp_act_bwd : dz = dy * act'(.) expressed through the layer OUTPUT y (post-activation):
 *               ReLU y>0, Sigmoid y(1-y), Exp y, Softplus 1-exp(-y) (= sigmoid of the input).
 * ngp_mlp_hidden_bwd : hidden-layer backward of a 2-layer MLP with a narrow output (n_out<=16,
 *               H in {32,64,128}): dz2 = dOut*act2'(out) (optional output), dz1 = (dz2.W2)*act1'(hidden);
 *               dOut == NULL means all ones (analytic d(sigma)/dx of the density head).
 * ldx/ldy/lddz/lddx/ldw are row strides in floats (>= the logical width); a W sub-block
 * (e.g. the columns of rgb_net's first layer that see the grid features) is addressed by
 * offsetting W and keeping ldw.
 * ---------------------------------------------------------------------- */
int ngp_linear_fwd(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* b,
                   int64_t n, int n_in, int n_out, int activation,
                   float* y, int64_t ldy, float* z_pre /* optional (n,n_out) pre-activation, may be NULL */,
                   void* stream);

/* Forward of a 2-layer MLP in one launch: hidden (n, H<=128) = act1(x W1^T + b1) is stored (the
 * backward needs it), out (n, n_out<=8) = act2(hidden W2^T + b2) is formed from the activated tile
 * in the MFMA kernel's epilogue — xyz_net (networks.py:54-59), rgb_net (89-100), norm_pred_header
 * (102-111), semantic_header (114-123, up to 8 classes).  b1 / b2 may be NULL (tcnn networks have no biases). */
int ngp_mlp2_fwd(const float* x, int64_t ldx, const float* W1, int64_t ldw1, const float* b1, int act1,
                 const float* W2, int64_t ldw2, const float* b2, int act2, int64_t n, int n_in, int H,
                 int n_out, float* hidden, int64_t ldh, float* out, int64_t ldo, void* stream);

/* the same, also writing dact_out (n, n_out, row stride ldo) = act2'(z2) expressed through the output — bitwise what
 * ngp_act_bwd(NULL, out, ...) would compute afterwards (the density head's d sigma / d x pass starts from it) */
int ngp_mlp2_fwd_dact(const float* x, int64_t ldx, const float* W1, int64_t ldw1, const float* b1, int act1,
                      const float* W2, int64_t ldw2, const float* b2, int act2, int64_t n, int n_in, int H,
                      int n_out, float* hidden, int64_t ldh, float* out, int64_t ldo, float* dact_out, void* stream);

int ngp_linear_bwd_input(const float* dz, int64_t lddz, const float* W, int64_t ldw,
                         int64_t n, int n_in, int n_out, float* dx, int64_t lddx,
                         int accumulate /* dx += instead of dx = */, void* stream);

int ngp_linear_bwd_weight(const float* dz, int64_t lddz, const float* x, int64_t ldx,
                          int64_t n, int n_in, int n_out, float* dW, int64_t ldw,
                          float* db /* may be NULL */, void* stream);

int ngp_act_bwd(const float* dy, const float* y, int64_t count, int activation,
                float* dz, void* stream);
/* the same over (n, cols <= 4) rows, which also accumulates sum_rows ||dz[row, :]||_2 into *row_norm_acc (what
 * ngp_row_norm_sum would compute from dz afterwards: the norm-bound sums of ngp_clip_decide without a launch of their own) */
int ngp_act_bwd_rows(const float* dy, const float* y_or_z, int64_t n, int cols, int activation, float* dz,
                     float* row_norm_acc, void* stream);

int ngp_mlp_hidden_bwd(const float* dOut, int64_t lddo, const float* out, int64_t ldo, int act2,
                       const float* W2, int64_t ldw2, const float* hidden, int64_t ldh, int act1,
                       int64_t n, int H, int n_out, float* dz2, int64_t lddz2,
                       float* dz1, int64_t lddz1,
                       float* dW2 /* NULL, or (n_out<=4, H): += dz2^T . hidden from the same pass */,
                       int64_t lddw2, float* db2 /* NULL or (n_out): += column sums of dz2 */, void* stream);

/* Backward of the FIRST layer of a 2-layer MLP  x -> hidden = act1(x W1^T + b1) -> out = act2(hidden W2^T + b2)
 * (n_out <= 4: xyz_net, rgb_net, norm_pred_header) with the hidden-layer gradient
 *     dz1 = act1'(hidden) * (dz2 . W2),   dz2 (n, n_out) = dL/dout * act2'(out)  [ngp_act_bwd]
 * formed inside the MFMA product while its operand tile is staged, so that dz1 is never written to or
 * read from HBM (it is n x 128 floats, and the plain route ngp_mlp_hidden_bwd -> ngp_linear_bwd_*
 * moves it three times):
 *   ngp_mlp_bwd_input : dx (n, n_in) (+)= dz1 . W1[:, :n_in]
 *   ngp_mlp_bwd_weight: dW1 (H, n_in) += dz1^T . x,  db1 (H) += column sums of dz1 (db1 may be NULL);
 *                       optionally also the SECOND layer's dW2 / db2, whose operands (dz2, hidden) this
 *                       product streams anyway */
int ngp_mlp_bwd_input(const float* dz2, int64_t lddz2, const float* W2, int64_t ldw2, const float* hidden,
                      int64_t ldh, int act1, const float* W1, int64_t ldw1, int64_t n, int n_in, int H,
                      int n_out, float* dx, int64_t lddx, int accumulate, void* stream);
int ngp_mlp_bwd_weight(const float* dz2, int64_t lddz2, const float* W2, int64_t ldw2, const float* hidden,
                       int64_t ldh, int act1, const float* x, int64_t ldx, int64_t n, int n_in, int H,
                       int n_out, float* dW1, int64_t ldw, float* db1,
                       float* dW2 /* NULL, or (n_out, H): += dz2^T . hidden from the same pass */,
                       int64_t lddw2, float* db2 /* NULL or (n_out): += column sums of dz2 */, void* stream);

/* ------------------------------------------------------------------------
 * fused Adam step (torch.optim.Adam(eps=1e-8) semantics, train.py:244) over one
 * flat fp32 tensor; optionally scales the gradient first (grad clipping /
 * 1/world_size) and zeroes it afterwards.
 * step is the 1-based step count.
 * ---------------------------------------------------------------------- */
int ngp_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t step, const float* grad_scale /* device scalar or NULL */,
                  int zero_grad, void* stream);

/* the same with the launch width named: a grid-stride sweep on `workgroups` workgroups of 256 threads (0: the default, 2 per
 * CU = 512 on MI355X).  The result does not depend on it; what does is how the sweep shares the CUs with kernels on other
 * streams — one workgroup per CU leaves room for 8-wave MLP workgroups beside it, two finish the sweep sooner; which is
 * faster per step depends on the loop around it, so the trainer measures. */
int ngp_adam_step_width(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                        float lr, float beta1, float beta2, float eps, float weight_decay,
                        int64_t step, const float* grad_scale /* device scalar or NULL */,
                        int zero_grad, int workgroups, void* stream);

/* sum of squares of a flat tensor accumulated into *out (device scalar; caller zeroes) */
int ngp_sumsq(const float* x, int64_t n, float* out, void* stream);

/* torch.nn.utils.clip_grad_norm_ coefficient (train.py: gradient_clip_val=50) on device:
 * norm = sqrt(*sumsq)*extra_scale; *coef = extra_scale * min(1, max_norm/(norm+1e-6)).
 * extra_scale carries 1/world_size for summed (not yet averaged) data-parallel gradients. */
int ngp_clip_coef(const float* sumsq, float max_norm, float extra_scale, float* coef, void* stream);

/* Gradient clipping settled from an upper bound of the norm, without reading the 0.8 GB table gradients.
 * A 2-layer MLP (hidden activation with |act'| <= 1: ReLU, Softplus) sends dfeat[s] = (act'(h) * (dz2[s] . W2)) . W1
 * into its encoder, so ||dfeat[s]||_2 <= ||dz2[s]||_2 ||W2||_F ||W1||_F, and the encoder's scatter adds, per sample,
 * a vector whose norm is at most ||dfeat[s]||_2 (trilinear weights: sum of squares <= 1).  Hence
 *   ||table gradient||_2 <= ||W1||_F ||W2||_F * sum_s ||dz2[s]||_2 .
 * ngp_row_norm_sum accumulates sum_s ||x[s, 0:cols]||_2 into *out (caller zeroes);
 * ngp_clip_decide takes those sums for two tables (row_norm_sums[0], [1]), the two weight blocks of each MLP and the
 * EXACT sum of squares of every other gradient, forms bound = sqrt(B0^2 + B1^2 + *sumsq_rest) * extra_scale and, when
 * bound * 1.001 + 1e-6 < max_norm, writes *coef = extra_scale (the exact coefficient: nothing is clipped) and
 * *need_exact = 0; otherwise *need_exact = 1 and the caller's ngp_sumsq_if / ngp_clip_coef_if launches (no-ops when the
 * flag is 0) compute the exact norm and coefficient.  NaN / inf anywhere takes the exact route. */
int ngp_row_norm_sum(const float* x, int64_t ldx, int64_t n, int cols, float* out, void* stream);
int ngp_clip_decide(const float* row_norm_sums, const float* w1_a, int64_t n1_a, const float* w2_a, int64_t n2_a,
                    const float* w1_b, int64_t n1_b, const float* w2_b, int64_t n2_b, const float* sumsq_rest,
                    float max_norm, float extra_scale, float* coef, int32_t* need_exact, void* stream);
/* ngp_clip_decide with the exact part summed by the same launch: *sumsq (in: what has been summed already, usually 0)
 * += sum of squares of rest[0:n_rest] (the MLP gradients, ~40 k entries), then the decision as above with it. */
int ngp_clip_decide_rest(const float* row_norm_sums, const float* w1_a, int64_t n1_a, const float* w2_a, int64_t n2_a,
                         const float* w1_b, int64_t n1_b, const float* w2_b, int64_t n2_b, const float* rest,
                         int64_t n_rest, float* sumsq, float max_norm, float extra_scale, float* coef,
                         int32_t* need_exact, void* stream);
int ngp_sumsq_if(const float* x, int64_t n, float* out, const int32_t* flag, void* stream);
int ngp_clip_coef_if(const float* sumsq, float max_norm, float extra_scale, float* coef, const int32_t* flag,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NGP_HIP_H */
