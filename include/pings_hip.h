/* pings_hip.h — C ABI of libpings_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary of the PINGS render + SDF-query hot path
 * (SURVEY.md §8b).  Every entry point is `extern "C"`, takes plain device
 * pointers, sizes and an opaque `hipStream_t` (passed as void*), returns an
 * integer status (0 = ok) and never throws.  All pointers are DEVICE pointers
 * unless a parameter is documented as host.  All floating point is fp32, all
 * tensors are dense row-major (the layout torch hands over).
 *
 * The reference binds this functionality through three CUDA torch extensions
 * whose sources are absent from the reference checkout (empty submodules,
 * /root/reference/.gitmodules:1-9) and through plain PyTorch code; each block
 * below cites the reference call site it replaces.
 */
#ifndef PINGS_HIP_H_
#define PINGS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#if defined(__cplusplus)
#define PINGS_EXTERN_C extern "C"
#else
#define PINGS_EXTERN_C
#endif
#define PINGS_API PINGS_EXTERN_C __attribute__((visibility("default")))

/* status codes */
#define PINGS_OK 0
#define PINGS_ERR_ARG 1      /* bad shape / null pointer / unsupported option */
#define PINGS_ERR_HIP 2      /* a HIP runtime call failed                      */
#define PINGS_ERR_CAPACITY 3 /* caller-provided buffer too small               */

/* ------------------------------------------------------------------ general */

/* ABI version of this header (bumped on any signature change).  A binding compares pings_abi_version() of the
 * library it loaded with the PINGS_ABI_VERSION of the header it was written against (pings_amd/_lib.py does). */
#define PINGS_ABI_VERSION 8
PINGS_API int pings_abi_version(void);
/* Message of the last failing call on this thread ("" if none). Host string. */
PINGS_API const char* pings_last_error(void);

/* Per-stage HIP-event timing of everything this library launches (bench.py's roofline
 * figures).  pings_prof_report synchronises the device and writes "name count total_ms"
 * lines into a HOST buffer, then clears the record.  Each recorded stage costs two event packets on the stream
 * (~10 us of idle time between dependent launches): pings_prof_only("stage") restricts the record to one stage
 * (NULL / "" = all), which is what bench.py's timed region uses for the dominant kernel. */
PINGS_API int pings_prof_enable(int on);
PINGS_API int pings_prof_only(const char* stage);
PINGS_API int pings_prof_report(char* buf, size_t cap);

/* --------------------------------------------------------------- fused SSIM
 * Replaces `fused_ssim.fused_ssim(img1, img2, train=...)`
 * (reference call sites utils/mapper.py:1243,1922,1951; arithmetic of the
 * in-tree torch predecessor gaussian_splatting/utils/loss_utils.py:189-219:
 * 11x11 Gaussian window sigma 1.5, zero "same" padding, C1=0.01^2, C2=0.03^2,
 * mean over all elements).  Images are [planes, H, W] with planes = N*C.
 */

/* Number of floats of scratch `partials` needed by pings_ssim_forward. */
PINGS_API size_t pings_ssim_partials_count(int planes, int H, int W);

/* out_mean[1] <- mean SSIM.  If train != 0 the three partial-derivative maps
 * ([planes,H,W] each) are written for the backward pass; otherwise they may be
 * NULL.  Deterministic (two-stage reduction, no atomics). */
PINGS_API int pings_ssim_forward(const float* img1, const float* img2, int planes, int H, int W,
                                 int train, float* out_mean, float* dm_dmu1,
                                 float* dm_dsigma1_sq, float* dm_dsigma12, float* partials,
                                 void* stream);

/* dL_dimg1[planes,H,W] <- gradient of (dL_dmean * mean SSIM) w.r.t. img1.
 * dL_dmean is a 1-element device buffer (the upstream scalar gradient). */
PINGS_API int pings_ssim_backward(const float* img1, const float* img2, int planes, int H, int W,
                                  const float* dL_dmean, const float* dm_dmu1,
                                  const float* dm_dsigma1_sq, const float* dm_dsigma12,
                                  float* dL_dimg1, void* stream);


/* ------------------------------------------- Gaussian(-surfel) rasteriser
 * Replaces the `diff_gaussian_surfel_rasterization` / `diff_gaussian_rasterization`
 * torch extensions (absent CUDA submodules, /root/reference/.gitmodules:1-6) at the
 * interface the reference calls them through:
 *   GaussianRasterizationSettings(...)  gaussian_renderer/__init__.py:149-166, :185-199
 *   GaussianRasterizer.markVisible      gaussian_renderer/__init__.py:215
 *   GaussianRasterizer.__call__         gaussian_renderer/__init__.py:318-326, :415-423
 *   its autograd backward               triggered by utils/mapper.py:1581
 * Semantics are those of oracle/raster_cpu.py (assumptions listed in DESIGN.md).
 *
 * Memory protocol: three opaque scratch blobs sized by the *_bytes queries and
 * allocated by the caller (torch's caching allocator in the Python wrapper);
 * the same blobs must be handed unchanged to pings_raster_backward.
 */
#define PINGS_RASTER_SURFEL 0
#define PINGS_RASTER_3DGS 1

typedef struct pings_raster_settings {
  int32_t image_height, image_width;
  int32_t mode;           /* PINGS_RASTER_SURFEL | PINGS_RASTER_3DGS                     */
  int32_t front_only;     /* config[4] of the surfel settings (surfel mode only)          */
  double tanfovx, tanfovy;
  double scale_modifier;
  const float* bg;             /* device [3]                                               */
  const float* viewmatrix;     /* device [4,4] row-major = T_cw^T   (cameras.py:214)       */
  const float* projmatrix;     /* device [4,4] = viewmatrix @ P^T   (cameras.py:216)       */
  const float* projmatrix_raw; /* device [4,4] = P^T                (cameras.py:68-70)     */
  const float* prcppoint;      /* device [2] = (cx/W, cy/H) (cameras.py:61); NULL = centre */
} pings_raster_settings;

/* present[N] (uint8) <- 1 where the point lies inside the view frustum. */
PINGS_API int pings_raster_mark_visible(const float* positions, int N,
                                        const pings_raster_settings* s, uint8_t* present,
                                        void* stream);

PINGS_API size_t pings_raster_geom_bytes(int P, int image_height, int image_width);
PINGS_API size_t pings_raster_binning_bytes(int64_t num_instances, int image_height,
                                            int image_width);
PINGS_API size_t pings_raster_image_bytes(int image_height, int image_width);

/* Stage 1: per-Gaussian projection, culling, depth sort, occlusion bound and tile counting.
 * Writes radii[P] (int32, 0 = culled) and *num_instances (HOST int64: number of
 * (Gaussian, tile) pairs that can still reach a pixel: pairs behind the depth rank at which a
 * conservative bound proves every pixel of the tile saturated are never created; set
 * PINGS_RASTER_OCCLUSION=0 in the environment to keep them all) and *footprint_class (HOST: 1 = footprints of a few
 * tiles, 2 = of many tiles; hand it unchanged to pings_raster_render / pings_raster_backward, which pick their blend
 * kernels by it: 1 -> one pixel per lane forward, Gaussian-per-lane wave scans backward; 2 -> two pixels per lane,
 * pixel-per-lane backward).  Synchronises `stream` once to return the two. */
PINGS_API int pings_raster_preprocess(const pings_raster_settings* s, int P, const float* means3D,
                                      const float* colors, const float* opacities,
                                      const float* scales, const float* rotations,
                                      void* geom_blob, int32_t* radii, int64_t* num_instances,
                                      int32_t* footprint_class, void* stream);
/* The same for a frame whose producer kernels left row counts on the DEVICE (`render`, gaussian_renderer/__init__.py:
 * 219,305,563-569,730-737 are four host synchronisations in the reference; this call is the one of the frame here):
 * of the first `dyn_rows` Gaussians only rows [0, *live_rows_dev) exist (the rest of that worst-case sized block is
 * culled without being read); rows [dyn_rows, P) are ordinary.  live_rows_dev NULL = every row exists.
 * aux_dev: HOST array of up to 8 DEVICE pointers to int32 words (NULL entries read as 0) that are copied to the HOST
 * array aux_host in the same read-back as the instance count. */
PINGS_API int pings_raster_preprocess_dyn(const pings_raster_settings* s, int P, const float* means3D,
                                          const float* colors, const float* opacities, const float* scales,
                                          const float* rotations, void* geom_blob, int32_t* radii,
                                          const int32_t* live_rows_dev, int dyn_rows,
                                          const int32_t* const* aux_dev, int aux_words, int32_t* aux_host,
                                          int64_t* num_instances, int32_t* footprint_class, void* stream);

/* Stage 2: instance binning (stable tile sort) and per-tile alpha blending.
 * Outputs are planar [C,H,W].  out_normal may be NULL in 3DGS mode.  `per_gaussian`
 * is float contributions[P] (surfel: sum of blend weights over pixels) or int32
 * n_touched[P] (3DGS: pixels where the Gaussian is seen with transmittance > 0.5).
 */
PINGS_API int pings_raster_render(const pings_raster_settings* s, int P, int64_t num_instances,
                                  void* geom_blob, void* binning_blob, void* image_blob,
                                  float* out_color, float* out_normal,
                                  float* out_depth, float* out_alpha, void* per_gaussian,
                                  int footprint_class, void* stream);

/* Scratch bytes pings_raster_backward needs (an upper bound in the instance count: the
 * gradient rows of the instances that actually blended are a data-dependent subset). */
PINGS_API size_t pings_raster_backward_bytes(int P, int64_t num_instances);

/* Backward of stage 1+2.  dL_d* of the outputs may be NULL (treated as zero).
 * bwd_blob: device scratch of pings_raster_backward_bytes(P, num_instances) bytes.
 * Gradient outputs are overwritten (not accumulated); dL_dmeans2D[P,3] receives the
 * screen-space positional gradient (xy, z = 0) the reference reads through
 * `viewspace_points`; dL_dtau[6] = [d rho, d theta] for T_cw <- SE3_exp(tau) T_cw
 * (utils/campose_utils.py:79-98).  Deterministic: no floating-point atomics. */
PINGS_API int pings_raster_backward(const pings_raster_settings* s, int P, int64_t num_instances,
                                    const float* means3D, const float* colors,
                                    const float* opacities, const float* scales,
                                    const float* rotations, const void* geom_blob,
                                    const void* binning_blob, const void* image_blob,
                                    const float* out_color, const float* out_normal,
                                    const float* out_depth, const float* out_alpha,
                                    const float* dL_dcolor, const float* dL_dnormal,
                                    const float* dL_ddepth, const float* dL_dalpha,
                                    void* bwd_blob, float* dL_dmeans3D, float* dL_dmeans2D,
                                    float* dL_dcolors, float* dL_dopacities, float* dL_dscales,
                                    float* dL_drotations, float* dL_dtau, int footprint_class,
                                    void* stream);

/* Debug / parity taps (tests only): copies of the sorted instance list and the
 * per-tile ranges out of the binning blob, and per-pixel state out of the image blob. */
PINGS_API int pings_raster_debug_lists(const void* binning_blob, int64_t num_instances,
                                       int image_height, int image_width, uint32_t* point_list,
                                       uint32_t* ranges_xy, void* stream);
PINGS_API int pings_raster_debug_image(const void* image_blob, int image_height, int image_width,
                                       float* final_T, uint32_t* n_contrib, void* stream);


/* -------------------------------------------- neural-point kNN + SDF decode
 * Replaces, for the query path, `NeuralPoints.radius_neighborhood_search`
 * (model/neural_gaussians.py:1061-1115), the search / filter / top-k part of
 * `NeuralPoints.query_feature` (:506-569) and, fused, the whole of `Mapper.sdf`
 * (utils/mapper.py:2273-2289 = query_feature + Decoder.sdf (model/decoder.py:62-104)
 * + inverse-distance weighting) with its analytic gradient (utils/tools.py:409-419).
 * All index tensors keep the reference's dtypes (int64 table / indices, int32
 * timestamps, uint8 bool masks) so the reference's own tensors are passed as they are.
 */
typedef struct pings_knn_map {
  const int64_t* table;          /* buffer_pt_index[buffer_size], -1 = empty (neural_gaussians.py:86-88) */
  int64_t buffer_size;
  const float* neural_points;    /* [Np,3] global positions                                   */
  const int32_t* point_ts_create;/* [Np]   (travel-distance window; may be NULL if !time_filtering) */
  const float* travel_dist;      /* [T]                                                        */
  int32_t cur_ts;
  int32_t time_filtering;        /* temporal_local_map_on && query_locally (:535)              */
  float diff_travel_dist_local;
  const uint8_t* free_mask;      /* [Np] free_gs_mask;  used iff use_free_mask  (:544-546)     */
  const uint8_t* valid_mask;     /* [Np] valid_gs_mask; used iff use_valid_mask (:548-550)     */
  int32_t use_free_mask, use_valid_mask;
  const int64_t* global2local;   /* [Np+1] or NULL for a global query (:553-554)               */
  const int32_t* neighbor_dx;    /* [K,3] cell offsets (:1030-1043)                            */
  int32_t K;
  int32_t nn_k;                  /* config.query_nn_k (<= 16)                                  */
  float resolution;              /* voxel size                                                 */
  float max_valid_dist2;         /* 3*((n+1)*res)^2 (:1058)                                    */
  const void* compact;           /* optional compact mirror of `table` (pings_knn_compact_build); when
                                    non-NULL the lookups go there (same results, cache resident)  */
  uint32_t compact_mask;         /* entries - 1                                                */
  const void* blocks;            /* optional cell-block index (pings_knn_blocks_build): block table ...   */
  const void* block_records;     /* ... its packed 32-byte point records ...                              */
  const int32_t* blocks_ok;      /* ... and its status word on the device: the kernels take the index only
                                    when it reads 1 and fall back to `compact` / `table` otherwise        */
  uint32_t block_mask;           /* block-table entries - 1                                               */
} pings_knn_map;

/* Compact, cache-resident mirror of the (>= 99 % empty) dense hash table: open addressing over
 * `entries` = pings_knn_compact_entries(num_points) 8-byte {slot+1, value} buckets.  A lookup
 * returns exactly table[slot] (or -1), so results are unchanged; the 81 random gathers of a
 * query then hit L2 / Infinity Cache instead of missing to HBM.  Rebuild after every change of
 * `table` (the wrapper keys it on the tensor's version counter). */
PINGS_API size_t pings_knn_compact_entries(int64_t num_points);
PINGS_API int pings_knn_compact_build(const int64_t* table, int64_t buffer_size, void* compact,
                                      size_t entries, void* stream);

/* Cell-block index: the search-side state of the map re-laid for locality.  The reference's table is keyed by a
 * hash of the CELL (neural_gaussians.py:1078-1084), so the 81 neighbour cells of a query are 81 unrelated 64-byte
 * sectors, and every live candidate costs further gathers (position, creation time, masks, local index).  The index
 * groups 4x4x4 cells into a block {packed block coordinate, 64-bit occupancy, first record} found through a small
 * open-addressing table, and keeps one 32-byte record per registered point {x, y, z, global index, global2local,
 * travel distance at creation, free / valid bits}: a query reads ~8 block entries and its ~10 live records instead
 * of ~120 sectors.  A point is registered iff table[hash(cell(point))] == point, i.e. iff the reference's lookup of
 * its own cell returns it.  Results equal the table path's exactly when (a) every non-empty table slot holds a
 * registered point and (b) no two distinct cells closer than the acceptance radius share a slot; the build verifies
 * (a) on the device and (b) on the host, writes status[0] = 1 only then, and the search kernels test that word.
 *
 * `m` carries the tensors to bake: table, buffer_size, neural_points, resolution, max_valid_dist2 and — each
 * optional, NULL = not baked, a query that needs it must then not pass the index — point_ts_create + travel_dist,
 * free_mask, valid_mask, global2local.  max_abs_dx = max |neighbor_dx| (host value).  blocks:
 * pings_knn_blocks_entries(num_points) * 32 bytes; records: max(num_points, 1) * 32 bytes; status: 8 int32
 * {ok, registered points, non-empty slots, out-of-range blocks, records, 0, 0, 0}.  Rebuild after ANY change of the
 * baked tensors. */
PINGS_API size_t pings_knn_blocks_entries(int64_t num_points);
PINGS_API int pings_knn_blocks_build(const pings_knn_map* m, int64_t num_points, int64_t num_timestamps,
                                     int32_t max_abs_dx, void* blocks, size_t entries, void* records,
                                     int32_t* status, void* stream);

/* idx[B,nn_k] (int64, -1 = none; local indices iff global2local != NULL), d2[B,nn_k]
 * (9e3 where idx = -1, :562), nn_counts[B] (int64: valid candidates over all K cells, :557).
 * Neighbours are ordered by squared distance, ties by candidate cell order.
 * global_idx[B,nn_k] (optional, may be NULL): the neighbours' indices before the
 * global2local mapping, i.e. the rows of `neural_points` the distances were measured to. */
PINGS_API int pings_knn_search(const pings_knn_map* m, const float* queries, int64_t B,
                               int64_t* idx, float* d2, int64_t* nn_counts, int64_t* global_idx,
                               void* stream);
/* `NeuralPoints.radius_neighborhood_search` itself (model/neural_gaussians.py:1061-1115): d2[B,K] (float) and idx[B,K]
 * (int64, -1 = no point) for every candidate cell, with the reference's step order (time window, python index -1 =
 * the last point, max_valid_dist2 for empty entries, collisions keep their distance).  num_points = rows of
 * neural_points.  The fused entry points never form this pair; `query_certainty` (:1117-1133) consumes it. */
PINGS_API int pings_knn_cells(const pings_knn_map* m, const float* queries, int64_t B, int64_t num_points, float* d2,
                              int64_t* idx, void* stream);

/* ---- NeuralPoints.query_feature (model/neural_gaussians.py:506-725): forward, backward, double backward ----
 * The tables are the LOCAL ones for a local query (local_geo_features [N_local+1, Fg], local_color_features,
 * local_neural_points, local_point_orientations, local_point_certainties) or the global ones; `m` carries the
 * search-side tensors (always the global neural_points / table / masks, plus global2local for a local query). */
typedef struct pings_qf_tables {
  const float* geo_features;    /* [rows, Fg] or NULL (query_geo_feature = False)                       */
  const float* color_features;  /* [rows, Fc] or NULL (query_color_feature = False / no colour table)   */
  int32_t Fg, Fc;               /* 0 when the table is NULL                                             */
  const float* points;          /* [rows - 1 .. rows, 3] positions in the index space of the features   */
  const float* orientations;    /* [.., 4] wxyz, read only when after_pgo (:627-630)                    */
  const float* certainties;     /* [..] read for the queried certainty (:691-695); may be NULL          */
  int32_t after_pgo;
  int32_t weighted_first;       /* config.weighted_first (:701-705): outputs are [B, F+3] weighted sums */
} pings_qf_tables;

/* Forward.  geo_out / color_out: [B, nn_k, F+3] (or [B, F+3] when weighted_first), NULL = not queried;
 * w_out [B, nn_k] normalised inverse-distance weights (:644-662; all 0 for a query without neighbours);
 * idx_out / gidx_out [B, nn_k] int64 (-1 = none): rows of the queried tables / of the global point array
 * (they differ for a local query: the weights are measured to the GLOBAL point, :1098-1101);
 * nn_counts [B] int64 over all K cells (:557); certainty [B] (optional).
 * Side effects of training mode (:664-689), each optional: certainty_accum[row] += w (float atomics, as the
 * reference's scatter_add_; pass a zeroed DELTA buffer distinct from t->certainties and add it to the table
 * afterwards: the queried certainty must see the values from before the accumulation, :615-620 vs :664-689),
 * ts_update[row] = max(ts_update[row], query_ts[b]) (integer atomics, exact). */
PINGS_API int pings_query_feature_forward(const pings_knn_map* m, const pings_qf_tables* t, const float* queries,
                                          int64_t B, float* geo_out, float* color_out, float* w_out,
                                          int64_t* idx_out, int64_t* gidx_out, int64_t* nn_counts,
                                          float* certainty, float* certainty_accum, const int32_t* query_ts,
                                          int32_t* ts_update, float* n_out, void* stream);
/* certainties[idx[p]] += w[p] for the n_pairs = B * nn_k pairs a forward call returned (idx < 0 skipped): the training-
 * mode accumulation of :664-689 as its own O(B nn_k) launch behind the forward kernel (stream order guarantees that the
 * queried certainty, :691-695, was computed from the values before the accumulation).  Float atomics, as the
 * reference's scatter_add_. */
PINGS_API int pings_query_feature_accumulate(const int64_t* idx, const float* w, int64_t n_pairs, float* certainties,
                                             void* stream);
/* n_out [B, nn_k, 3] (optional): the neighbour vectors on their own (they are also columns F..F+2 of the rows). */

/* Backward: upstream g_geo / g_color (shapes of the forward outputs, NULL = none) and g_w [B, nn_k] ->
 * g_x [B,3] (through the neighbour vectors AND the weights; squared distances stay in the graph, :1099-1101),
 * g_geo_features [rows, Fg] / g_color_features [rows, Fc] (dense, every row written; NULL = not wanted):
 * the deterministic row scatter-add of the upstream gradient (pings_rows_scatter_add).
 * `global_points` = m->neural_points of the forward.  scratch: pings_query_feature_scratch_bytes(B, nn_k, rows). */
PINGS_API size_t pings_query_feature_scratch_bytes(int64_t B, int nn_k, int64_t rows);
PINGS_API int pings_query_feature_backward(const pings_qf_tables* t, const float* global_points,
                                           const float* queries, int64_t B, int nn_k, const int64_t* idx,
                                           const int64_t* gidx, const float* g_geo, const float* g_color,
                                           const float* g_n, const float* g_w, int64_t rows, void* scratch,
                                           float* g_x, float* g_geo_features, float* g_color_features,
                                           void* stream);
/* Split layout (per-neighbour mode): g_n [B, nn_k, 3] != NULL carries the neighbour-vector part of the upstream
 * gradient on its own (g_geo / g_color are then ignored and the feature rows are scattered separately with
 * pings_rows_plan_build / _apply) — the autograd wrapper keeps the table path and the query path in separate graph
 * nodes so that get_gradient(x, sdf) does not pay for a table scatter nobody asked for. */

/* Backward of the backward (get_gradient(..., create_graph=True), utils/tools.py:409-419; used by
 * utils/mapper.py:874-875,:1445-1448): given gg_x [B,3] (and, optionally, gg_*_features [rows, F]) — the
 * gradients w.r.t. the backward's outputs — the gradients w.r.t. its inputs: d_g_geo / d_g_color (shapes of
 * g_geo / g_color), d_g_w [B, nn_k], d_x [B,3] (optional) and, in weighted_first mode only, the feature tables
 * d_geo_features / d_color_features [rows, F] (per-neighbour mode: the backward does not depend on them). */
PINGS_API int pings_query_feature_double_backward(
    const pings_qf_tables* t, const float* global_points, const float* queries, int64_t B, int nn_k,
    const int64_t* idx, const int64_t* gidx, const float* g_geo, const float* g_color, const float* g_w,
    const float* gg_x, const float* gg_geo_features, const float* gg_color_features, int64_t rows, void* scratch,
    float* d_g_geo, float* d_g_color, float* d_g_n, float* d_g_w, float* d_x, float* d_geo_features,
    float* d_color_features, void* stream);
/* d_g_n [B, nn_k, 3] (split layout, see above): replaces d_g_geo / d_g_color. */

/* Deterministic scatter-add of rows — the backward of a row gather `table[idx]`
 * (model/neural_gaussians.py:565-579; the reference's autograd uses atomics there):
 *   out[r, 0:F] = sum over pairs p with dst_row[p] == r, in ascending p, of w[p] * src[src_row[p] * ld + 0:F]
 * and 0 for rows nothing points at (every row of out[rows, F] is written; no memset needed).
 * dst_row[p] < 0 or >= rows skips the pair; w == NULL: weight 1; src_row == NULL: pair p reads row p.
 * F <= 64.  Counting sort by destination (integer atomics only) + one pass over the table: bitwise
 * reproducible.  `scratch`: pings_rows_scatter_add_scratch_bytes(n_pairs, rows) bytes. */
PINGS_API size_t pings_rows_scatter_add_scratch_bytes(int64_t n_pairs, int64_t rows);
PINGS_API int pings_rows_scatter_add(const int64_t* dst_row, int64_t n_pairs, const float* src, int64_t ld,
                                     int32_t F, const float* w, const int64_t* src_row, int64_t rows,
                                     void* scratch, float* out, void* stream);
/* The same in two steps for several tables sharing one destination index: `plan` (caller-owned,
 * pings_rows_plan_bytes(n_pairs, rows) bytes) is built once and applied per table; pair p reads row p of src. */
PINGS_API size_t pings_rows_plan_bytes(int64_t n_pairs, int64_t rows);
PINGS_API int pings_rows_plan_build(const int64_t* dst_row, int64_t n_pairs, int64_t rows, void* plan, void* stream);
PINGS_API int pings_rows_plan_apply(const void* plan, int64_t n_pairs, int64_t rows, const float* src, int64_t ld,
                                    int32_t F, const float* w, float* out, void* stream);

typedef struct pings_sdf_decoder {
  const float* W1;   /* [hidden, F+3] layers.0.weight (decoder.py:49) */
  const float* b1;   /* [hidden]                                      */
  const float* W2;   /* [1, hidden]   lout.weight                      */
  const float* b2;   /* [1]                                            */
  int32_t hidden;    /* <= 64                                          */
  int32_t feat_dim;  /* F (<= 61)                                      */
  float sdf_scale;   /* decoder.py:55-57                               */
  int32_t weighted_first; /* config.weighted_first (neural_gaussians.py:701) */
} pings_sdf_decoder;

/* Fused inference query: sdf[B]; optional (may be NULL) grad_x[B,3] = d sdf / d query,
 * nn_counts[B] (int64), certainty[B].  `features` is [rows, F] (local or global table matching
 * the index space of the search), `points`/`orientations`/`certainties` likewise
 * ([rows,3], [rows,4] wxyz or NULL unless after_pgo, [rows] or NULL). No side effects. */
PINGS_API int pings_sdf_forward(const pings_knn_map* m, const pings_sdf_decoder* dec,
                                const float* features, const float* points,
                                const float* orientations, const float* certainties,
                                int32_t after_pgo, const float* queries, int64_t B, float* sdf,
                                float* grad_x, int64_t* nn_counts, float* certainty,
                                int64_t* idx_out, float* w_out, float* sdf_std, int64_t* gidx_out,
                                void* stream);
/* idx_out[B,nn_k] / w_out[B,nn_k] (optional): the neighbours (in the index space of `features`) and
 * their normalised inverse-distance weights, kept for pings_sdf_backward; gidx_out[B,nn_k] (optional): the same
 * neighbours as rows of m->neural_points (what the weights were measured to), kept for pings_sdf_double_backward.  sdf_std[B] (optional): spread of the
 * per-neighbour predictions sqrt(sum_m w_m (s_m - sdf)^2), the tracker's validity filter
 * (utils/tracker.py:303-313,408); 0 in weighted_first mode. */

/* First-order backward of the fused query w.r.t. the feature table and the decoder
 * (the training path of Mapper.sdf_mapping, utils/mapper.py:822-970: loss(sdf).backward()).
 *   dL_dfeatures[rows,F]  dense, every row written (zero where no query touched the row)
 *   dL_dW1[H,F+3], dL_db1[H], dL_dW2[H], dL_db2[1]
 * One wave per query evaluates the decoder backward on the vector ALU (lane = hidden unit, weight-gradient rows
 * in registers), the per-(query, neighbour) feature-gradient rows are grouped by destination row (counting sort,
 * pings_rows_scatter_add below) and summed in ascending pair order; decoder gradients are per-workgroup partials
 * summed in fixed order.  Bitwise reproducible.  hidden <= 64.
 * `scratch` needs pings_sdf_backward_scratch_bytes(B, nn_k, F, H, feature_rows) bytes. */
PINGS_API size_t pings_sdf_backward_scratch_bytes(int64_t B, int nn_k, int feat_dim, int hidden,
                                                  int64_t feature_rows);
/* Backward of the backward: the consistency / Eikonal losses differentiate g = dS/dx (returned by
 * pings_sdf_forward as grad_x) once more (get_gradient(create_graph=True), utils/tools.py:409-419;
 * utils/mapper.py:1445-1448).  Given v[B,3] = dL/dg (times the dL/dS the first backward was called with), the
 * gradients of  sum_b <v_b, dS_b/dx>  w.r.t. the feature table and the decoder, same outputs, scratch and
 * determinism as pings_sdf_backward.  `gidx` [B,nn_k] / `global_points`: the rows of the global point array the
 * weights were measured to (pings_sdf_forward gidx_out; == idx / points for a global query).  The gradient w.r.t.
 * the query itself (a third derivative) is not produced. */
PINGS_API int pings_sdf_double_backward(const pings_sdf_decoder* dec, const float* features,
                                        int64_t feature_rows, const float* points, const float* orientations,
                                        const float* global_points, int32_t after_pgo, const float* queries,
                                        int64_t B, int nn_k, const int64_t* idx, const int64_t* gidx,
                                        const float* w, const float* v, void* scratch, float* d_features,
                                        float* d_W1, float* d_b1, float* d_W2, float* d_b2, void* stream);
PINGS_API int pings_sdf_backward(const pings_sdf_decoder* dec, const float* features,
                                 int64_t feature_rows, const float* points,
                                 const float* orientations, int32_t after_pgo, const float* queries,
                                 int64_t B, int nn_k, const int64_t* idx, const float* w,
                                 const float* dL_dsdf, void* scratch, float* dL_dfeatures,
                                 float* dL_dW1, float* dL_db1, float* dL_dW2, float* dL_db2,
                                 void* stream);


/* ------------------------------------------------------ decoder MLP (MFMA)
 * Replaces `Decoder.mlp` / `Decoder.mlp_batch` (model/decoder.py:62-98) for the one-hidden-level
 * ReLU decoders every shipped config builds (pings.py:147-172):
 *     y[N,OUT] = relu(x[N,IN] @ W1[HID,IN]^T + b1[HID]) @ W2[OUT,HID]^T + b2[OUT]
 * fp32 in / fp32 accumulate on the matrix cores (v_mfma_f32_32x32x2_f32: bitwise an fp32 fma chain).
 * Limits: IN <= 64, HID in {32,64,96,128}, OUT <= 32.
 */
PINGS_API size_t pings_mlp_backward_scratch_bytes(int IN, int HID, int OUT);
PINGS_API int pings_mlp_forward(const float* x, int64_t N, int IN, int HID, int OUT, const float* W1,
                                const float* b1, const float* W2, const float* b2, float* y,
                                void* stream);
/* Recomputes the hidden layer (nothing is saved by the forward).  dL_dx may be NULL.  Weight
 * gradients are per-workgroup partials summed in fixed order (no atomics). */
PINGS_API int pings_mlp_backward(const float* x, const float* dL_dy, int64_t N, int IN, int HID,
                                 int OUT, const float* W1, const float* b1, const float* W2,
                                 void* scratch, float* dL_dx, float* dL_dW1, float* dL_db1,
                                 float* dL_dW2, float* dL_db2, void* stream);
/* The backward of pings_mlp_backward's dL_dx output (the graph node autograd records when the mapper differentiates
 * dS/dx once more: utils/tools.py:409-419 `get_gradient(create_graph=True)`, used at utils/mapper.py:1445-1448 and
 * utils/tracker.py:317).  ddx[N,IN] = cotangent of dL_dx; outputs: d_dy[N,OUT] (cotangent of dL_dy), d_W1[HID,IN],
 * d_W2[OUT,HID]; nothing reaches x or b1 (piecewise-constant ReLU mask).  Shapes: HID = 64, OUT = 1 (`Decoder.sdf`,
 * model/decoder.py:100-104); _supported() says so, anything else is PINGS_ERR_ARG and the host wrapper composes the
 * node from device operators instead.  scratch: pings_mlp_backward_scratch_bytes(IN, HID, OUT). */
PINGS_API int pings_mlp_double_backward_supported(int IN, int HID, int OUT);
PINGS_API int pings_mlp_double_backward(const float* x, const float* ddx, const float* dL_dy, int64_t N, int IN,
                                        int HID, int OUT, const float* W1, const float* b1, const float* W2,
                                        void* scratch, float* d_dy, float* d_W1, float* d_W2, void* stream);

/* Several decoders over the SAME N rows in one launch each way (blockIdx.y = decoder): the five spawn decoders of
 * a view (gaussian_renderer/__init__.py:605-716; hidden 128, IN <= 32 — pings.py:156-160).  Results are bitwise
 * those of pings_mlp_forward / pings_mlp_backward called per decoder.  Unused directions leave their pointers NULL
 * (dL_dx may be NULL in the backward).  scratch: pings_mlp_backward_grouped_scratch_bytes(jobs, njobs). */
typedef struct pings_mlp_job {
  const float* x;                    /* [N, IN]                                   */
  int32_t IN, OUT;                   /* IN <= 32, OUT <= 32, hidden width = 128   */
  const float *W1, *b1, *W2, *b2;    /* [128, IN], [128], [OUT, 128], [OUT]       */
  float* y;                          /* forward:  [N, OUT]                        */
  const float* dL_dy;                /* backward: [N, OUT]                        */
  float* dL_dx;                      /*           [N, IN] or NULL                 */
  float *dL_dW1, *dL_db1, *dL_dW2, *dL_db2;
} pings_mlp_job;
PINGS_API int pings_mlp_forward_grouped(const pings_mlp_job* jobs, int njobs, int64_t N, void* stream);
/* N sizes the buffers and the grid; *n_rows_dev (device int32, NULL = N) rows are decoded, the rest is not touched. */
PINGS_API int pings_mlp_forward_grouped_dyn(const pings_mlp_job* jobs, int njobs, int64_t N, const int32_t* n_rows_dev,
                                            void* stream);
PINGS_API size_t pings_mlp_backward_grouped_scratch_bytes(const pings_mlp_job* jobs, int njobs);
PINGS_API int pings_mlp_backward_grouped(const pings_mlp_job* jobs, int njobs, int64_t N, void* scratch,
                                         void* stream);

/* ------------------------------------------------------- exposure correction of the rendered image
 * gaussian_renderer/__init__.py:449-461 (affine form): out[c, p] = sum_k M[c, k] img[k, p] + b[c] for the
 * [3, H W] planes — the reference's `img.permute(1,2,0).view(-1,3) @ M^T + b`, a GEMM with K = N = 3 there.
 * Backward: g_img (may be NULL), g_M [3,3], g_b [3]; sums in fp64, fixed order (bitwise reproducible).
 * scratch: pings_exposure_backward_scratch_bytes(). */
PINGS_API int pings_exposure_forward(const float* img, const float* M, const float* b, int64_t HW, float* out,
                                     void* stream);
PINGS_API size_t pings_exposure_backward_scratch_bytes(void);
PINGS_API int pings_exposure_backward(const float* img, const float* M, const float* g_out, int64_t HW,
                                      void* scratch, float* g_img, float* g_M, float* g_b, void* stream);

/* ------------------------------------------ finite-difference SDF gradient
 * The tensor code of `Mapper.get_numerical_gradient` (utils/mapper.py:2319-2370) around the fused SDF query:
 *   pings_stencil_points            x[N,3] -> [x+ex; x-ex; x+ey; x-ey; x+ez; x-ez] ([6N,3]; one-sided: 3 blocks, [3N,3])
 *   pings_sdf_forward / _backward   on the shifted points (above)
 *   pings_stencil_gradient          S[6N] -> grad[N,3] = (S+ - S-) / (2 eps)   (one-sided: (S+ - sdf_x) / eps)
 *   pings_stencil_gradient_backward dL/dgrad[N,3] -> dL/dS[6N] (one-sided also dL/dsdf_x[N], nullable)
 * Divisions are multiplications by the fp32 reciprocal of the fp32 divisor, as torch's device kernel for
 * tensor / python-float computes them. */
PINGS_API int pings_stencil_points(const float* x, int64_t N, float eps, int two_side, float* out, void* stream);
PINGS_API int pings_stencil_gradient(const float* sdf_shifted, const float* sdf_x, int64_t N, float eps, int two_side,
                                     float* grad, void* stream);
PINGS_API int pings_stencil_gradient_backward(const float* dL_dgrad, int64_t N, float eps, int two_side,
                                              float* dL_dsdf_shifted, float* dL_dsdf_x, void* stream);

/* ------------------------------------------------------- spawn_gaussians
 * Replaces the tensor code of `spawn_gaussians` around the five decoder MLPs
 * (gaussian_splatting/gaussian_renderer/__init__.py:469-778).  Call order for one view:
 *   pings_spawn_gather   -> dense per-view inputs of the MLPs          (:551-597,:672-675,:692-699)
 *   5 x pings_mlp_forward
 *   pings_spawn_plan     -> compacted row of every kept Gaussian + count (alpha > 0, scale filter; :727-761)
 *   pings_spawn_forward  -> activations + quaternion algebra, written to the compacted rows (:605-716)
 *   pings_spawn_backward -> gradients back to the raw MLP outputs (autograd of the same lines)
 * Raw MLP outputs are [n, d*k] row-major == [n*k, d]: Gaussian g = i*k + j of neural point i owns
 * floats g*d .. g*d+d-1 (the reference's .view(N*K, -1), :632,645,665,687,716).
 */
typedef struct {
  int n;                    /* neural points after the visible & valid mask */
  int k;                    /* Gaussians per neural point (Decoder.out_k) */
  int scale_dim;            /* scale-MLP columns per Gaussian (mlp_out_dim / k): 2 or 3 */
  int surfel;               /* 1: gaussian_surfel -> [s0, s1, 1e-7] (:668-670); 0: 3d_gs -> [s0, s1, s2] */
  int color_residual;       /* 1: clamp(base + 0.1 tanh(mlp), 0, 1) (:707-711); 0: sigmoid(mlp) (:714) */
  int alpha_filter_on;      /* keep tanh(alpha mlp) > 0 (:727-740) */
  int scale_filter_on;      /* keep any(scale > scale_filter_thr) (:747-761) */
  float displacement_range; /* displacement_range_ratio * resolution (:605) */
  float unit_scale;         /* unit_scale_ratio * resolution (:661) */
  float max_scale;          /* max_scale_ratio * resolution (:655) */
  float scale_filter_thr;   /* scale_filter_ratio * resolution (:751) */
} pings_spawn_params;

/* Rows `sel[n]` (int64; NULL = rows 0..n-1) of the map tensors -> pos[n,3], quat[n,4], base_color[n,3]
 * (if `color`), free_out[n] (if `free_mask`), geo_in[n, Fg + dist_concat], col_in[n, Fc + 3*view_concat],
 * view_dist[n] (nullable).  cam_origin: DEVICE [3] or NULL (then no view features).  xy_only zeroes the
 * z-component of the view vector before the norm (:592-597); the direction is rotated into the neural
 * point's frame with R(q) (apply_quaternion_rotation(quat_inverse(q), .), :695-696). */
PINGS_API int pings_spawn_gather(int n, const int64_t* sel, const float* position, const float* orientation,
                                 const float* color, const uint8_t* free_mask, const float* geo_feature, int Fg,
                                 const float* color_feature, int Fc, const float* cam_origin, int xy_only,
                                 int view_concat, int dist_concat, float* pos, float* quat, float* base_color,
                                 uint8_t* free_out, float* geo_in, float* col_in, float* view_dist,
                                 void* stream);
/* `_dyn` forms of gather / plan / forward: `n` (p->n) is the CAPACITY the buffers and grids are sized for and
 * *n_rows_dev (device int32; NULL = n) the rows the visible & valid mask actually selected (a count `render` no longer
 * reads back before spawning, gaussian_renderer/__init__.py:563-569): rows behind it are neither read nor written,
 * never kept by the plan, and the free-mask tiling (:724) uses the device count.  nan_flag (device int32, nullable):
 * zeroed by the plan, set to 1 by the forward when a spawned rotation is NaN (the reference's assert, :305-306). */
PINGS_API int pings_spawn_gather_dyn(int n, const int32_t* n_rows_dev, const int64_t* sel, const float* position,
                                     const float* orientation, const float* color, const uint8_t* free_mask,
                                     const float* geo_feature, int Fg, const float* color_feature, int Fc,
                                     const float* cam_origin, int xy_only, int view_concat, int dist_concat,
                                     float* pos, float* quat, float* base_color, uint8_t* free_out, float* geo_in,
                                     float* col_in, float* view_dist, void* stream);
PINGS_API int pings_spawn_plan_dyn(const pings_spawn_params* p, const int32_t* n_rows_dev, const float* alpha_raw,
                                   const float* scale_raw, const float* dist_ratio, void* scratch, int32_t* dest,
                                   int32_t* count, int32_t* nan_flag, void* stream);
PINGS_API int pings_spawn_forward_dyn(const pings_spawn_params* p, const int32_t* n_rows_dev, const float* xyz_raw,
                                      const float* rot_raw, const float* scale_raw, const float* alpha_raw,
                                      const float* color_raw, const float* pos, const float* quat,
                                      const float* base_color, const float* dist_ratio, const uint8_t* free_in,
                                      const int32_t* dest, float* gaussian_xyz, float* gaussian_scale,
                                      float* gaussian_rot, float* gaussian_alpha, float* gaussian_color,
                                      float* alpha_all, uint8_t* gaussian_free_mask, int32_t* nan_flag,
                                      void* stream);
/* Scatter of the per-view feature gradients (leading dimensions ldg / ldc, first Fg / Fc columns) into
 * rows `sel` of the map-sized, PRE-ZEROED gradient tensors.  Either pair may be NULL. */
PINGS_API int pings_spawn_gather_backward(int n, const int64_t* sel, const float* dL_dgeo_in, int Fg, int ldg,
                                          const float* dL_dcol_in, int Fc, int ldc, float* dL_dgeo_feature,
                                          float* dL_dcolor_feature, void* stream);
PINGS_API size_t pings_spawn_plan_scratch_bytes(int64_t num_gaussians);
/* dest[n*k]: compacted row of every Gaussian or -1 if dropped; *count (device int32) = kept Gaussians.
 * dist_ratio[n] (nullable) = view_dist / z_far when dist_adaptive_scale (:657-659). */
PINGS_API int pings_spawn_plan(const pings_spawn_params* p, const float* alpha_raw, const float* scale_raw,
                               const float* dist_ratio, void* scratch, int32_t* dest, int32_t* count,
                               void* stream);
/* dest NULL = no compaction.  Outputs: gaussian_xyz[count,3], gaussian_scale[count, surfel ? 3 : scale_dim],
 * gaussian_rot[count,4] (wxyz), gaussian_alpha[count,1], gaussian_color[count,3], alpha_all[n*k,1]
 * (pre-filter, :721), gaussian_free_mask[count] (nullable; Gaussian g takes free_in[g % n], the reference's
 * tiling at :724). */
PINGS_API int pings_spawn_forward(const pings_spawn_params* p, const float* xyz_raw, const float* rot_raw,
                                  const float* scale_raw, const float* alpha_raw, const float* color_raw,
                                  const float* pos, const float* quat, const float* base_color,
                                  const float* dist_ratio, const uint8_t* free_in, const int32_t* dest,
                                  float* gaussian_xyz, float* gaussian_scale, float* gaussian_rot,
                                  float* gaussian_alpha, float* gaussian_color, float* alpha_all,
                                  uint8_t* gaussian_free_mask, void* stream);
/* Any dL_d<output> may be NULL (= zero).  Gradients w.r.t. the raw MLP outputs, same shapes as the inputs. */
PINGS_API int pings_spawn_backward(const pings_spawn_params* p, const float* xyz_raw, const float* rot_raw,
                                   const float* scale_raw, const float* alpha_raw, const float* color_raw,
                                   const float* quat, const float* base_color, const float* dist_ratio,
                                   const int32_t* dest, const float* dL_dxyz, const float* dL_dscale,
                                   const float* dL_drot, const float* dL_dalpha, const float* dL_dcolor,
                                   const float* dL_dalpha_all, float* dL_dxyz_raw, float* dL_drot_raw,
                                   float* dL_dscale_raw, float* dL_dalpha_raw, float* dL_dcolor_raw,
                                   void* stream);

/* ------------------------------------------------- neural-point map maintenance
 * Replaces the per-frame torch bookkeeping of `NeuralPoints`:
 *   voxel_down_sample_torch   utils/tools.py:924-967
 *   NeuralPoints.update       model/neural_gaussians.py:214-375 (everything but the feature initialisation)
 *   reset_local_map           model/neural_gaussians.py:378-478
 *   assign_local_to_global    model/neural_gaussians.py:482-494
 * Indices, masks, timestamps and table entries are bit-exact with the reference (oracle/map_cpu.py, G8 vectors).
 * Where the reference's index_put_ meets duplicate targets the LAST sample wins (its CPU semantics), here
 * deterministically.  Map arrays are caller-owned with room for the appended rows (capacity >= num_points + M).
 */
PINGS_API size_t pings_voxel_downsample_scratch_bytes(int64_t N);
/* sample_idx[<= N] (int64): index of the point closest to its voxel centre, one per occupied voxel, ordered by the
 * reference's linear voxel id; *count (HOST) = number of voxels.  Synchronises `stream` once. */
PINGS_API int pings_voxel_downsample(const float* points, int64_t N, float voxel_size, void* scratch,
                                     int64_t* sample_idx, int64_t* count, void* stream);
/* `voxel_down_sample_min_value_torch(points, voxel_size, value)` (utils/tools.py:970-1009): per occupied voxel the
 * point with the smallest `value` (1000 bins of value / max(value), ties by index); value NULL = the distance to the
 * voxel centre (pings_voxel_downsample).  Same output order, same single synchronisation. */
PINGS_API int pings_voxel_downsample_min_value(const float* points, const float* value, int64_t N, float voxel_size,
                                               void* scratch, int64_t* sample_idx, int64_t* count, void* stream);
/* Loop-closure maintenance of the map (model/neural_gaussians.py:871-1010):
 *   pings_map_prune_mask  prune[i] = |travel[cur_ts] - travel[ts_update[i]]| > diff_travel && certainty[i] < threshold
 *                         (:873-880; the caller counts, compacts the rows with pings_gather_rows and rebuilds the hash)
 *   pings_map_adjust      in place: p <- R p + t, q <- quat(R) (x) q with pose_diff[ts] (ts = create, or the mid
 *                         timestamp when use_mid_ts; :911-937); pose_diff is [num_poses, 4, 4] row-major, float64 when
 *                         pose_is_f64 (the quaternion part is then evaluated in float64 and cast, as the reference's
 *                         type promotion does) else float32
 *   pings_map_rehash      table[:] = -1; table[hash(points[v_j])] = v_j with v_j = sample_idx[j] (NULL: j) for
 *                         j < M, duplicates: the last j wins (:949-1000); slot_scratch: M int64
 * Timestamps index travel_dist[num_travel] / pose_diff[num_poses] with python semantics (negative wraps); one outside
 * [-T, T) is an IndexError in the reference: the kernels never read out of bounds — they set bit 0 of the device
 * word `out_of_range` (nullable; the caller zeroes it and reads it with its next read-back) and keep / do not move
 * that point.  cur_ts outside [0, num_travel) is PINGS_ERR_ARG. */
PINGS_API int pings_map_prune_mask(int64_t N, const float* travel_dist, int64_t num_travel, int32_t cur_ts,
                                   const int32_t* point_ts_update, const float* point_certainties,
                                   float diff_travel_dist_local, float prune_certainty_thre, uint8_t* prune_mask,
                                   int32_t* out_of_range, void* stream);
PINGS_API int pings_map_adjust(int64_t N, float* neural_points, float* point_orientations, const int32_t* point_ts_create,
                               const int32_t* point_ts_update, int32_t use_mid_ts, const void* pose_diff,
                               int32_t pose_is_f64, int64_t num_poses, int32_t* out_of_range, void* stream);
PINGS_API int pings_map_rehash(const float* neural_points, const int64_t* sample_idx, int64_t M, float resolution,
                               int64_t buffer_size, int64_t* table, int64_t* slot_scratch, void* stream);
PINGS_API size_t pings_map_update_scratch_bytes(int64_t M, int64_t num_points);
/* Inserts the M voxel representatives.  table[buffer_size] int64; travel_dist NULL = no travel-distance window
 * (temporal_local_map_on False); sample_colors / point_colors nullable together.  update_mask[M] (uint8, nullable)
 * <- 1 where the sample became a new neural point; *num_new (HOST).  Appends rows num_points .. num_points+num_new-1
 * of every map array (identity orientation, ts = cur_ts, certainty 0, free = !is_reliable, valid = 1, colour +
 * colour validity) and refreshes the colour of existing points whose colour was invalid.  Synchronises once. */
PINGS_API int pings_map_update(const float* sample_points, const float* sample_colors, int64_t M, float resolution,
                               int64_t buffer_size, int64_t* table, int64_t num_points, const float* travel_dist,
                               int cur_ts, float diff_travel_dist_local, int is_reliable, float* neural_points,
                               float* point_orientations, int32_t* point_ts_create, int32_t* point_ts_update,
                               float* point_certainties, uint8_t* free_gs_mask, uint8_t* valid_gs_mask,
                               float* point_colors, uint8_t* valid_color_mask, void* scratch, uint8_t* update_mask,
                               int64_t* num_new, void* stream);
PINGS_API size_t pings_map_reset_local_scratch_bytes(int64_t num_points);
/* local_mask / sorrounding_mask [num_points+1] (uint8, last = 1), global2local[num_points+1] (int64: rank for local
 * points, 1 for the others — the reference's full_like(bool,-1) — and -1 for the padding entry),
 * local_idx[num_points+1] (int64: the local rows in order, followed by the padding row num_points),
 * *num_local (HOST).  sensor_position: DEVICE [3].  Synchronises once. */
PINGS_API int pings_map_reset_local(int64_t num_points, const float* neural_points, const int32_t* point_ts_create,
                                    const int32_t* point_ts_update, const float* travel_dist, int cur_ts,
                                    int use_mid_ts, int use_travel_dist, float diff_travel_dist_local,
                                    int diff_ts_local, const float* sensor_position, int range_filter_2d,
                                    float local_radius, float sorrounding_radius, void* scratch, uint8_t* local_mask,
                                    uint8_t* sorrounding_mask, int64_t* global2local, int64_t* local_idx,
                                    int64_t* num_local, void* stream);
/* `NeuralPoints.gather_local_data` (model/neural_gaussians.py:1135-1173) indexes nine per-point tensors with the
 * boolean surrounding mask, one nonzero + host synchronisation each in torch.  Here:
 *   pings_mask_rows         rows[<= n] (int64, ascending) = the indices where mask[i] != 0; count_and_last (HOST, two
 *                           int64): their number and mask[n - 1] (the feature tables carry a padding row that the
 *                           per-point tensors do not: :1159-1161).  Synchronises `stream` once (polled read-back).
 *   pings_gather_rows_multi dst_g[i] = src_g[idx[i]] for i < rows_g, all tensors g in ONE launch (any row width). */
typedef struct pings_gather_job {
  const void* src;
  void* dst;
  int64_t row_bytes;
  int64_t rows;
} pings_gather_job;
PINGS_API size_t pings_mask_rows_scratch_bytes(int64_t n);
PINGS_API int pings_mask_rows(const uint8_t* mask, int64_t n, void* scratch, int64_t* rows, int64_t* count_and_last,
                              void* stream);
PINGS_API int pings_gather_rows_multi(const pings_gather_job* jobs, int njobs, const int64_t* idx, void* stream);
/* dst[i] = src[idx[i]] / dst[idx[i]] = src[i] for rows of row_bytes bytes (any dtype). */
PINGS_API int pings_gather_rows(const void* src, int64_t row_bytes, const int64_t* idx, int64_t n, void* dst,
                                void* stream);
PINGS_API int pings_scatter_rows(const void* src, int64_t row_bytes, const int64_t* idx, int64_t n, void* dst,
                                 void* stream);

/* ------------------------------------------------------------ depth2normal
 * Replaces `depth2normal(depth, mask, camera, img_scale)` (gaussian_splatting/utils/point_utils.py:83-149) as `render`
 * uses it (gaussian_renderer/__init__.py:330-335), with the multiplication by the detached rendered alpha fused in:
 *   normal[3,H,W] = normalize(sum of the four neighbour cross products of the un-projected depth) * mask * alpha.
 * Visibility: `mask[H,W]` (uint8) if given, else `alpha > min_alpha`; `alpha` NULL = no weighting.  cx, cy, fx, fy are
 * the principal point / focal lengths in pixels of the (down-scaled) image.  Gradient flows to the depth only (mask
 * and alpha are detached in the reference). */
PINGS_API int pings_depth2normal_forward(const float* depth, const float* alpha, const uint8_t* mask, int H, int W,
                                         float cx, float cy, float fx, float fy, float min_alpha, float* normal,
                                         void* stream);
PINGS_API size_t pings_depth2normal_backward_scratch_bytes(int H, int W);
PINGS_API int pings_depth2normal_backward(const float* depth, const float* alpha, const uint8_t* mask, int H, int W,
                                          float cx, float cy, float fx, float fy, float min_alpha,
                                          const float* dL_dnormal, void* scratch, float* dL_ddepth, void* stream);

/* ------------------------------------------------------------ image-space losses
 * Replaces the photometric loss block of `Mapper.joint_gsdf_mapping` (utils/mapper.py:1197-1295; helpers
 * `l1_loss` / `sky_mask_loss`, gaussian_splatting/utils/loss_utils.py:17,178) — one fused pass per direction:
 *   losses[0] = |rgb - gt_rgb|.mean() over rows [v_min, v_max)                                   (:1224-1239)
 *   losses[1] = mean over (depth_min < gt_depth < depth_max) & (alpha > min_accu_alpha) of |gt - d|, or of
 *               |1/gt - 1/d| when inverse_depth                                                    (:1251-1267)
 *   losses[2] = mean over (|n| > 0) & (|m| > 0) of |m||n| - <n, m>, n = rendered normal, m = depth normal, both
 *               zeroed at sky pixels first; norms detached; consist_mode 1 detaches n, 2 detaches m (:1213-1216,1273-1295)
 *   losses[3] = alpha[sky].mean()                                                                  (:1198-1210)
 * All un-weighted (the lambdas stay with the caller).  Planes are [C,H,W] fp32, sky_mask [H,W] uint8; depth/gt_depth,
 * alpha, normal/depth_normal and sky_mask are optional (NULL); a mean over nothing is NaN as in torch.  `sums[8]`
 * (fp64: sum, count per loss) is kept for the backward pass, which takes dL/dlosses[4] from device memory and writes
 * whichever gradient planes are non-NULL. */
typedef struct {
  int H, W;
  int v_min, v_max;          /* rows of the colour loss, v_max exclusive (python slice already resolved) */
  float depth_min, depth_max, min_accu_alpha;
  int inverse_depth;
  int consist_mode;          /* 0 both sides learn, 1 gs_consist_normal_fixed, 2 gs_consist_depth_fixed */
} pings_image_loss_params;
PINGS_API size_t pings_image_losses_scratch_bytes(void);
PINGS_API int pings_image_losses_forward(const pings_image_loss_params* p, const float* rgb, const float* gt_rgb,
                                         const float* depth, const float* gt_depth, const float* alpha,
                                         const float* normal, const float* depth_normal, const uint8_t* sky_mask,
                                         void* scratch, double* sums, float* losses, void* stream);
PINGS_API int pings_image_losses_backward(const pings_image_loss_params* p, const float* rgb, const float* gt_rgb,
                                          const float* depth, const float* gt_depth, const float* alpha,
                                          const float* normal, const float* depth_normal, const uint8_t* sky_mask,
                                          const double* sums, const float* dL_dlosses, float* dL_drgb,
                                          float* dL_ddepth, float* dL_dalpha, float* dL_dnormal,
                                          float* dL_ddepth_normal, void* stream);

/* ------------------------------------------------------ colour / semantic heads of a kNN query
 * `Mesher.query_points` (utils/mesher.py:132-153) and the tracker's colour query (utils/tracker.py:322-331) apply an
 * activation to every neighbour's decoder output raw[B, k, C] and sum over the k neighbours with the IDW weights
 * weight[B, k] (NULL = `weighted_first`: k = 1, weight one):
 *   PINGS_HEAD_COLOR     out_value[B, C] = sum_j w_j sigmoid(raw_j)              (Decoder.regress_color, decoder.py:133-134)
 *   PINGS_HEAD_SEMANTIC  out_label[B] = argmax_c sum_j w_j log_softmax(raw_j)[c], first maximum as torch.argmax
 *                        (Decoder.sem_label_prob, decoder.py:119-122); out_value (nullable) receives the [B, C] sums
 * k <= 16.  One streaming pass instead of three to five torch passes per head. */
#define PINGS_HEAD_COLOR 0
#define PINGS_HEAD_SEMANTIC 1
PINGS_API int pings_head_reduce(const float* raw, const float* weight, int64_t B, int32_t k, int32_t C, int32_t mode,
                                float* out_value, int64_t* out_label, void* stream);

/* ------------------------------------------------------ tracker registration
 * Replaces the Jacobian assembly of `implicit_reg` (utils/tracker.py:608-689): with J_i = [p_i x g_i, g_i] (rotation
 * first, then translation),  N = sum_i w_i J_i^T J_i  (6x6)  and  g = -sum_i w_i r_i J_i  (6).
 * points[n,3], sdf_grad[n,3], sdf_residual[n], weight[n] -> out[42] = N row-major (36) followed by g (6), fp32,
 * accumulated in fp64 with a fixed-order two-stage reduction (bitwise reproducible). */
PINGS_API size_t pings_reg_normal_equations_scratch_bytes(void);
PINGS_API int pings_reg_normal_equations(const float* points, const float* sdf_grad, const float* sdf_residual,
                                         const float* weight, int64_t n, void* scratch, float* out, void* stream);
/* The rest of the step (utils/tracker.py:649-689, :774-783) from the 42 floats above: N += lm_lambda diag(N) in fp32,
 * t = N^-1 g in fp64 (Gaussian elimination with partial pivoting), T[4,4] = [expmap(t[:3]) | t[3:]] row-major fp64;
 * t_out[6] optional.  One tiny kernel instead of ~30 torch launches and the host sync of linalg.inv. */
PINGS_API int pings_reg_solve(const float* normal_eq, float lm_lambda, double* T_out, double* t_out, void* stream);
/* The same step with a conditioning report.  The reference's `torch.linalg.inv` (utils/tracker.py:668) raises on a
 * singular N; the kernel above cannot raise, so it leaves a bit mask in the device word `status_dev`:
 *   PINGS_REG_SINGULAR         a pivot of the damped matrix is exactly 0 or not finite (the case the reference raises on)
 *   PINGS_REG_ILL_CONDITIONED  smallest |pivot| < 1e-7 x largest |entry|: the step is rounding noise (N arrives in fp32)
 *   PINGS_REG_NONFINITE        t or T came out Inf / NaN
 * `status_host` (optional, host memory): the call waits for the kernel with a polled read-back and copies the word —
 * the one synchronisation the reference has at this line as well. */
#define PINGS_REG_SINGULAR 1
#define PINGS_REG_ILL_CONDITIONED 2
#define PINGS_REG_NONFINITE 4
PINGS_API int pings_reg_solve_checked(const float* normal_eq, float lm_lambda, double* T_out, double* t_out,
                                      int32_t* status_dev, int32_t* status_host, void* stream);

#endif /* PINGS_HIP_H_ */
