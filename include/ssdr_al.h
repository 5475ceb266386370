/* ssdr_al.h — C ABI of libssdr_al.so, the MI355X (gfx950) implementation of the SSDR-AL hot path
 *
 *     grid-subsample -> KNN pyramid -> RandLA-Net inference -> FPS-GCN / k-center selection
 *
 * Plain pointers and sizes only; no C++ or torch types.  Every entry point returns an int status
 * (SSDR_OK == 0); ssdr_last_error() gives the message for the calling thread's last failure.
 *
 * Two flavours of each op:
 *   host entry points   take host pointers, are synchronous, and are the drop-in replacements for the
 *                       reference's native modules (they copy in, run the HIP kernels, copy out);
 *   *_dev entry points  take device pointers + a HIP stream handle (void*, NULL = the library's own
 *                       stream), enqueue only, and let a caller keep tiles resident in HBM between
 *                       stages.  Outputs are valid after the stream is synchronised
 *                       (ssdr_stream_sync).
 *
 * There is no CPU fallback: without a HIP device every compute entry point fails with
 * SSDR_ERR_NO_DEVICE.
 *
 * Reference interfaces are cited relative to /root/reference/SSDR_AL_s3dis/ ("S3/").
 */
#ifndef SSDR_AL_H
#define SSDR_AL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDR_OK               0
#define SSDR_ERR_INVALID      1   /* bad argument (shape, NULL, unsupported dim/K) */
#define SSDR_ERR_NO_DEVICE    2   /* no HIP device / HIP runtime failure at init */
#define SSDR_ERR_HIP          3   /* a HIP call failed */
#define SSDR_ERR_EMPTY        4   /* empty result (reference: RuntimeError("Error"), wrapper.cpp:225-229) */
#define SSDR_ERR_UNSUPPORTED  5   /* valid for the reference but outside what the kernels cover */
#define SSDR_ERR_INTERNAL     6   /* device-side consistency flag raised (e.g. kd-tree deeper than the search stack) */

/* ---- library / device ---------------------------------------------------------------------- */
const char* ssdr_version(void);
const char* ssdr_last_error(void);
int  ssdr_init(int device);              /* idempotent; selects the HIP device, creates stream + workspace */
void ssdr_shutdown(void);
int  ssdr_stream_sync(void* stream);     /* NULL = library stream */
/* Extra streams for callers that overlap independent work (e.g. the per-room front end); every *_dev entry point
 * keeps its workspaces per stream, so calls on different streams may run concurrently. */
int  ssdr_stream_create(void** out_stream);
/* priority > 0: the device's highest stream priority, < 0: its lowest, 0: the default (hipStreamCreateWithPriority, non-blocking).  Streams of
 * another priority live on hardware queues of their own: short dependent launches are not held behind the other streams' chip-filling kernels. */
int  ssdr_stream_create_priority(void** out_stream, int priority);
int  ssdr_stream_destroy(void* stream);
int  ssdr_main_stream(void** out_stream);            /* the library's own stream (what stream == NULL means) */
int  ssdr_stream_wait(void* waiter, void* waited);   /* waiter continues after what is enqueued on waited so far */
/* events: ssdr_event_record marks a point in a stream's work; ssdr_stream_wait_event makes a stream wait for that point (issued any time later) */
int  ssdr_event_create(void** ev);
int  ssdr_event_record(void* ev, void* stream);
int  ssdr_stream_wait_event(void* stream, void* ev);
int  ssdr_event_destroy(void* ev);
/* Optional per-launch timing (HIP events on the launch stream) of the instrumented kernels; used by bench.py for
 * the roofline line.  ssdr_prof_report() synchronises and returns "name calls total_ms total_work total_work2\n" lines, where
 * total_work is the summed algorithmic FLOPs (MFMA kernels) or bytes (HBM kernels) of those launches and total_work2 the FLOPs
 * their MFMA instructions execute (0 for the other kernels). */
int ssdr_prof_enable(int on);
const char* ssdr_prof_report(void);
/* Milliseconds spent in the GPU part of the last host-flavour call (HIP events on the library stream). */
float ssdr_last_gpu_ms(void);

/* ---- KNN (replaces S3/utils/nearest_neighbors: knn_.h:4-26, knn_.cxx:22-135, knn.pyx:33-109) --
 * Exact K nearest neighbours of every query among `npts` support points, ascending squared
 * distance, with nanoflann v1.2.3's tie order (kd-tree traversal order, leaf size 10), i.e. the
 * int64 indices equal the reference's cpp_knn* output bit for bit.  dim must be 3; the host entry points also take dim 1 and 2 (answered exactly as the
 * 3-D problem with the missing coordinates at zero); dim > 3: SSDR_ERR_UNSUPPORTED.
 * If K > npts the slots >= npts hold 0 (what the reference's zero-initialised buffers leave there).
 * ssdr_knn           <-> cpp_knn / cpp_knn_omp              (knn_.cxx:22-69)
 * ssdr_knn_batch     <-> cpp_knn_batch / cpp_knn_batch_omp  (knn_.cxx:72-135)
 */
int ssdr_knn(const float* points, size_t npts, size_t dim,
             const float* queries, size_t nqueries, size_t K, int64_t* indices);
int ssdr_knn_batch(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                   const float* queries, size_t nqueries, size_t K, int64_t* batch_indices);
/* Same, but int32 output: the dtype DataProcessing.knn_search hands on (S3/helper_tool.py:173-183). */
int ssdr_knn_batch_i32(const float* batch_data, size_t batch_size, size_t npts, size_t dim,
                       const float* queries, size_t nqueries, size_t K, int32_t* batch_indices);
/* Device flavour: d_* are device pointers; out is int32 [B,nq,K]. */
int ssdr_knn_batch_dev(const float* d_batch_data, size_t batch_size, size_t npts, size_t dim,
                       const float* d_queries, size_t nqueries, size_t K, int32_t* d_indices, void* stream);
/* The device flavours (ssdr_knn_batch_dev, ssdr_knn_pyramid_dev) only enqueue work, so they cannot report what the kernels
 * found.  ssdr_knn_status waits for `stream` and returns SSDR_ERR_INTERNAL if the last KNN call issued on it overflowed one of
 * its device-side capacities (kd queue / node table / level limit, hand-over list); the host flavours check this themselves.
 * out4 (optional, host): rows handed from the grid search to the exact tree walk for K = 16 and K = 1 (tie rows, see
 * csrc/knn_grid.hip), the status bits, and the depth of the deepest tree built. */
int ssdr_knn_status(void* stream, int32_t* out4);
/* The same check without waiting: every device-flavour KNN call leaves a ticket (its counters copied to pinned memory behind its
 * kernels); this folds the tickets of the calls that HAVE finished and returns SSDR_ERR_INTERNAL if one of them overflowed.  For
 * callers that keep several batches in flight on `stream` and must not wait for the newest one (ssdr_al/pipeline.py). */
int ssdr_knn_status_poll(void* stream, int32_t* out4);

/* ---- KNN pyramid (replaces the loop of tf_map, S3/s3dis_dataset.py:156-183) ------------------
 * For level i in [0,num_layers): N_0 = npts, N_{i+1} = N_i / ratio[i] (integer division);
 *   neigh_idx[i]  int32 [B,N_i,K]       = knn(xyz[:, :N_i], xyz[:, :N_i], K)
 *   sub_idx[i]    int32 [B,N_{i+1},K]   = neigh_idx[i][:, :N_{i+1}]        (a prefix: not materialised
 *                                         separately unless d_sub_idx != NULL)
 *   interp_idx[i] int32 [B,N_i,1]       = knn(xyz[:, :N_{i+1}], xyz[:, :N_i], 1)
 * d_neigh_idx / d_sub_idx / d_interp_idx are host arrays of num_layers device pointers. */
int ssdr_knn_pyramid_dev(const float* d_xyz, size_t batch_size, size_t npts,
                         size_t num_layers, const int32_t* ratios, size_t K,
                         int32_t* const* d_neigh_idx, int32_t* const* d_sub_idx,
                         int32_t* const* d_interp_idx, void* stream);
/* Host flavour (host pointers everywhere, synchronous). */
int ssdr_knn_pyramid(const float* xyz, size_t batch_size, size_t npts,
                     size_t num_layers, const int32_t* ratios, size_t K,
                     int32_t* const* neigh_idx, int32_t* const* sub_idx, int32_t* const* interp_idx);

/* ---- grid subsampling (replaces S3/utils/cpp_wrappers/cpp_subsampling: grid_subsampling.cpp:5-106,
 *      called through wrapper.cpp:58-276) -------------------------------------------------------
 * Voxel-grid barycentres: per occupied voxel the mean position, mean feature vector (fp32, summed in
 * input order) and the majority label per label column (ties and row order as in the reference, see
 * `order`).  features / classes may be NULL (then fdim / ldim are ignored).
 * order: SSDR_ORDER_REFERENCE  rows in the reference's order (iteration order of its
 *                              std::unordered_map<size_t,...>, GCC libstdc++)  — bit-identical output
 *        SSDR_ORDER_KEY        rows by ascending voxel key (cheaper)
 * Two-step host protocol: ssdr_grid_subsample computes and keeps the result on the device and
 * returns M in *out_m; ssdr_grid_subsample_fetch copies it into caller-owned arrays of M rows. */
#define SSDR_ORDER_REFERENCE 0
#define SSDR_ORDER_KEY       1
int ssdr_grid_subsample(const float* points, size_t n,
                        const float* features, size_t fdim,
                        const int32_t* classes, size_t ldim,
                        float sampleDl, int order, size_t* out_m);
int ssdr_grid_subsample_fetch(float* out_points, float* out_features, int32_t* out_classes);
/* Device flavour: outputs must have room for n rows; *d_out_m (device int64) receives M. */
int ssdr_grid_subsample_dev(const float* d_points, size_t n,
                            const float* d_features, size_t fdim,
                            const int32_t* d_classes, size_t ldim,
                            float sampleDl, int order,
                            float* d_out_points, float* d_out_features, int32_t* d_out_classes,
                            int64_t* d_out_m, void* stream);

/* All clouds of a batch in one call (the per-cloud chain is ~45 small launches; batching divides that by the batch).
 * The clouds are concatenated: cloud r owns rows [cloud_offsets[r], cloud_offsets[r+1]) (host int64 [num_clouds+1]) of
 * d_points / d_features / d_classes; its M_r output rows are written from row cloud_offsets[r] of the output arrays and
 * M_r to d_out_m[r].  Rows by ascending voxel key (SSDR_ORDER_KEY).  At most 64 clouds per call. */
int ssdr_grid_subsample_batch_dev(const float* d_points, const float* d_features, size_t fdim, const int32_t* d_classes, size_t ldim,
                                  const int64_t* cloud_offsets, size_t num_clouds, float sampleDl,
                                  float* d_out_points, float* d_out_features, int32_t* d_out_classes, int64_t* d_out_m, void* stream);
/* The device flavours only enqueue work: what their kernels found is read here (waits for `stream`).  bit 0 = a voxel with more labels in one
 * column than the per-voxel table holds (the host flavour returns SSDR_ERR_UNSUPPORTED for the same). */
/* Which implementation ssdr_grid_subsample_batch_dev uses (process-wide).
 *   SSDR_SUBSAMPLE_AUTO (default)  rows of at most 7 words (3 + fdim + ldim <= 7): one partition pass into spatial buckets of 8 x 8 x 8 (or
 *                                  16 x 8 x 8) voxels and one workgroup per bucket that orders and reduces its voxels in LDS; clouds whose grid
 *                                  needs more than 16384 such buckets, or that hold a voxel of more than 1024 points, are reported by
 *                                  ssdr_grid_subsample_status (bits 2 / 4) and must be repeated with SSDR_SUBSAMPLE_SORT.  Other rows: the sort.
 *   SSDR_SUBSAMPLE_SORT            segmented radix sort of (voxel key, index) words: any grid, any voxel population.
 * Both give the reference's rows bit for bit (grid_subsampling.cpp:5-106), by ascending voxel key. */
#define SSDR_SUBSAMPLE_AUTO 0
#define SSDR_SUBSAMPLE_SORT 1
int ssdr_grid_subsample_set_method(int method);
int ssdr_grid_subsample_status(void* stream, int32_t* out_status);

/* ---- Semantic3D sampling loader: the cut of a whole scan into network inputs ---------------------------------------
 * split3 (SSRD_AL_semantic3d/semantic3d_dataset_sampling.py:198-236) + the merge rule of tf_map (:243-255): the cloud is halved along x and y at
 * the middle of its bounding box (float32 comparisons against float64 mid-points, as NumPy evaluates them; the z test of :224 holds for every
 * point and is reproduced), a part of more than max_size points is split again (the parts of THAT split against recurse_max_size: the
 * reference's recursive call passes the literal 800000 whatever its caller's max_size was, :233), and a part of at most merge_max points joins
 * the one before it.  d_xyz [n,3].  Outputs: d_order [n] = the points grouped by combined part (ascending index inside a leaf, leaves in the order split3
 * appends them; the reference's order inside a part is CPython's set iteration order), part_offsets (host, max_parts + 1 entries) and
 * *num_parts; d_part [n] (may be NULL) = the combined part of every point.  Synchronous (the recursion tree is decided on the host).
 * SSDR_ERR_UNSUPPORTED: a part above max_size that does not split (coincident points: the reference recurses without end). */
int ssdr_split3_dev(const float* d_xyz, size_t n, size_t max_size, size_t recurse_max_size, size_t merge_max, int32_t* d_part, int32_t* d_order, int64_t* part_offsets,
                    size_t max_parts, size_t* num_parts, void* stream);

/* ---- tile generator (spatially_regular_gen, S3/s3dis_dataset.py:115-154; data_aug, S3/helper_tool.py:185-199) --
 * From a (sub-sampled) cloud resident on the device — d_points [*,3], d_colors [*,color_dim], live row count
 * *d_m (device int64, as written by ssdr_grid_subsample_dev; n_max bounds it) — take the num_points points nearest
 * to center[3] (host), order them by the caller's shuffle d_perm (a permutation of [0,num_points)), subtract the
 * centre, and emit d_out_xyz [num_points,3], d_out_feat [num_points,3+color_dim] = [xyz_centred, colors*scale]
 * (may be NULL) and d_out_idx [num_points] = source row (may be NULL).  A cloud with fewer than num_points rows
 * is padded by duplicating row floor(d_dup_u[r] * m) of the shuffled list (d_dup_u: uniform [0,1) floats). */
int ssdr_tile_select_dev(const float* d_points, const float* d_colors, int color_dim, const int64_t* d_m, size_t n_max,
                         const float* center, size_t num_points, const int32_t* d_perm, const float* d_dup_u, float color_scale,
                         float* d_out_xyz, float* d_out_feat, int32_t* d_out_idx, void* stream);

/* Batch form: cloud r = rows [cloud_offsets[r], cloud_offsets[r+1]) with live count d_m[r]; centers host [num_clouds,3];
 * d_perm / d_dup_u [num_clouds, num_points]; outputs [num_clouds, num_points, ...].  d_out_idx = source row inside its cloud.
 * d_labels (int32, one per row, may be NULL) -> d_out_labels [num_clouds, num_points] = queried_pc_label (s3dis_dataset.py:141). */
int ssdr_tile_select_batch_dev(const float* d_points, const float* d_colors, int color_dim, const int64_t* d_m, const int64_t* cloud_offsets,
                               size_t num_clouds, const float* centers, size_t num_points, const int32_t* d_perm, const float* d_dup_u,
                               float color_scale, float* d_out_xyz, float* d_out_feat, int32_t* d_out_idx, const int32_t* d_labels,
                               int32_t* d_out_labels, void* stream);
/* Test-time variant (S3/s3dis_dataset_test.py:105-143): the same tile, plus the possibility-map update
 * possibility[idx] += (1 - dists/max(dists))^2 over the tile's (un-padded) points (float64 map, float32 dists) and,
 * optionally, min / argmin of the updated map (the next pick: :106-108). */
int ssdr_tile_select_possibility_dev(const float* d_points, const float* d_colors, int color_dim, const int64_t* d_m, size_t n_max,
                                     const float* center, size_t num_points, const int32_t* d_perm, const float* d_dup_u, float color_scale,
                                     float* d_out_xyz, float* d_out_feat, int32_t* d_out_idx,
                                     double* d_possibility, double* d_out_min_possibility, int32_t* d_out_argmin, void* stream);

/* ---- prune: the voxel grid of the superpoint partition (replaces libply_c.prune, S3/partition/ply_c/ply_c.cpp:289-383, called from
 *      partition/partition.py:126-144) ---------------------------------------------------------------------------------------
 * bin = floor((p - bbox_min) / voxel_size); per voxel: float32 position sum / count, uint32 colour sum / count truncated to uint8,
 * histogram of labels [n_labels + 1] and of objects [n_objects + 1] (uint32), rows in the order in which the voxels are first met
 * in the input.  d_rgb / d_out_rgb may be NULL; labels / objects are skipped when n_labels / n_objects is 0.  Outputs are sized
 * for n rows; *d_out_m (device int64) receives the number of voxels.  ssdr_prune_status waits for the stream and reports a
 * grid with more than 2^21 bins along an axis or an id above its declared maximum (the reference's .at() would throw). */
int ssdr_prune_dev(const float* d_xyz, size_t n, float voxel_size, const uint8_t* d_rgb, const uint8_t* d_labels, int n_labels,
                   const uint32_t* d_objects, int n_objects, float* d_out_xyz, uint8_t* d_out_rgb, uint32_t* d_out_labels,
                   uint32_t* d_out_objects, int64_t* d_out_m, void* stream);
int ssdr_prune_status(void* stream, int32_t* out_status);

/* ---- RandLA-Net inference (replaces the TF1 graph of S3/RandLANet.py:140-180, 505-585 run by
 *      model.sess.run([prob_logits, last_second_features, ...]) in S3/sampler2.py:598 / :327) ----------
 * Activations, accumulation and every non-matrix operation are fp32.  The matrix products run in one of three arithmetic
 * modes (ssdr_randla_set_precision): SSDR_PREC_F32 (default) exact f32-input MFMA; SSDR_PREC_BF16X3 both operands carried as
 * two bf16 pieces, hi*hi + lo*hi + hi*lo on the bf16 MFMA with fp32 accumulation (within the 1e-3 feature tolerance of
 * the fp32 path); SSDR_PREC_BF16 operands rounded to bf16 once (BASELINE configuration 3; tolerance reported separately).
 * Weights are handed over per layer with batch-norm already folded
 * (W [in,out] row-major, b [out]); the layer table is documented in csrc/randla_model.hip and built from the
 * reference's variable scopes by ssdr_al/randlanet.py.  Inputs are device pointers:
 *   d_features [B,N0,in_dim]   d_xyz [B,N0,3] (level i uses the first N_i points, tf_map's prefix sub-sampling)
 *   d_neigh_idx[i] int32 [B,N_i,16]  (sub_idx[i] is its prefix)   d_interp_idx[i] int32 [B,N_i,1]
 * Outputs: d_probs [B*N0,C] = softmax(logits) (RandLANet.py:84), d_feat32 [B*N0,32] = last_second_features
 * (RandLANet.py:45,175). */
int  ssdr_randla_create(int num_layers, const int32_t* d_out, int k_n, int num_classes, int in_dim, void** handle);
int  ssdr_randla_num_layers(void* handle);
int  ssdr_randla_layer_shape(void* handle, int layer, int* in, int* out, int* has_bias);
int  ssdr_randla_set_layer(void* handle, int layer, const float* W, const float* b);   /* host pointers */
#define SSDR_PREC_F32 0
#define SSDR_PREC_BF16X3 1
#define SSDR_PREC_BF16 2
int  ssdr_randla_set_precision(void* handle, int mode);
/* formulation of the K-expanded building_block halves in the two bf16 modes: 1 (default) = 32 x 32 MFMA tiles with the softmax over the
 * neighbours inside the lane (csrc/randla_lfa32.hip, every level), 0 = the 16 x 16-tile kernels of rounds 1-3 (csrc/randla_bf16.hip; their level 0
 * runs on the exact-f32 kernel).  Same results within the tolerance above; kept selectable for A/B timing and as a second check of either. */
int  ssdr_randla_set_formulation(void* handle, int tiles32);
void ssdr_randla_destroy(void* handle);
int  ssdr_randla_infer_dev(void* handle, size_t batch_size, size_t npts, const float* d_features, const float* d_xyz,
                           const int32_t* ratios, int32_t* const* d_neigh_idx, int32_t* const* d_interp_idx,
                           float* d_probs, float* d_feat32, void* stream);

/* ---- selection stage (replaces the NumPy / sklearn host code of S3/sampler2.py:12-47,102-115,262-266,313-342,
 *      612-640, S3/fps_gcn_cpu.py:12-178 and S3/kcenterGreedy.py:60-128) ---------------------------------------
 * Superpoints are CSR: sp_off int32 [S+1] into sp_pts int32 [T] (point ids) == the reference's `components`.
 * All pointers are device pointers; float64 where the reference computes in float64. */
#define SSDR_UNC_LC 0       /* 1 - max p                          (sampler2.py:29-34) */
#define SSDR_UNC_ENTROPY 1  /* -sum p log2 p                      (:35-40, :247-255)  */
#define SSDR_UNC_SB 2       /* second best / best                 (:41-44)            */
#define SSDR_REGION_MEAN 0        /* sampler2.py:13-14 */
#define SSDR_REGION_SUM_WEIGHT 1  /* :15-18 */
#define SSDR_REGION_WETSU 2       /* :19-26 */
/* compute_point_uncertainty + argmax class (sampler2.py:28-47, :602) */
int ssdr_point_uncertainty_dev(const float* d_probs, size_t n, int num_classes, int mode, float* d_unc, int32_t* d_cls, void* stream);
/* per-superpoint region uncertainty (float64), dominant predicted class and its member count (sampler2.py:612-626) */
int ssdr_region_stats_dev(const float* d_unc, const int32_t* d_cls, const int32_t* d_sp_off, const int32_t* d_sp_pts, size_t S,
                          int num_classes, int mode, double* d_region_unc, int32_t* d_dom, int32_t* d_dom_cnt, void* stream);
/* ssdr_max_dominant: per-superpoint dominant ground-truth label + purity (sampler2.py:102-106, :127-144) */
int ssdr_dominant_label_dev(const int32_t* d_labels, const int32_t* d_sp_off, const int32_t* d_sp_pts, size_t S, int num_labels,
                            int32_t* d_label, double* d_purity, void* stream);
/* add_clsbal (sampler2.py:262-266), in place on d_region_unc; n_selected == 0 is add_classbal (:256-260).  d_skip [S] (may be NULL) != 0: the
 * region is not part of the population — prediction() collects region_class over the UNLABELLED regions of at least min_size points only
 * (sampler2.py:612-627), so labelled / too small regions enter neither the histogram nor its length (their own values are scaled and never used) */
int ssdr_clsbal_dev(const int32_t* d_region_class, size_t S, const uint8_t* d_skip, const int32_t* d_selected_class_list, size_t n_selected,
                    double* d_region_unc, void* stream);
/* the same in two steps for sharded runs: local class histogram (int32[64]), then — after the caller has summed the
 * histograms of all ranks — the scaling with the global histogram / global count */
int ssdr_class_hist_dev(const int32_t* d_region_class, size_t S, const uint8_t* d_skip, const int32_t* d_selected_class_list, size_t n_selected,
                        int32_t* d_hist64, void* stream);
int ssdr_clsbal_hist_dev(const int32_t* d_region_class, size_t S, const int32_t* d_hist64, size_t total, double* d_region_unc, void* stream);
/* sorted_inds = argsort(-u) (sampler2.py:640); equal values keep ascending index */
int ssdr_rank_regions_dev(const double* d_region_unc, size_t S, int32_t* d_sorted_inds, void* stream);
/* compute_features (sampler2.py:333,339): float32 mean of feat rows over the dominant-class members of the
 * superpoints d_sel[0..nsel) (d_sel == NULL: superpoints 0..nsel-1) */
int ssdr_segment_mean_features_dev(const float* d_feat, int feat_dim, const int32_t* d_cls, const int32_t* d_dom,
                                   const int32_t* d_sp_off, const int32_t* d_sp_pts, const int32_t* d_sel, size_t nsel,
                                   float* d_out, void* stream);
/* Device-side pieces of the sharded selection's exchanges (ssdr_al/distributed.py runs the RCCL collectives
 * directly on these buffers, on the caller's stream):
 * ssdr_mask_regions_dev: out[i] = labelled[i] ? -inf : region_unc[i] for i < S, -inf for S <= i < S_padded (what every rank
 * contributes to the all-gather before the global ranking; labelled regions and padding sort last);
 * ssdr_gather_rows_dev: out[r] = in[idx[r]] for rows of row_bytes bytes (compacts the padded all-gather of candidate features). */
int ssdr_mask_regions_dev(const double* d_region_unc, const uint8_t* d_labelled, size_t S, size_t S_padded, double* d_out, void* stream);
int ssdr_gather_rows_dev(const void* d_in, const int32_t* d_idx, size_t n, size_t row_bytes, void* d_out, void* stream);
/* y0[i] = (y1[i] =) (double)x[i]: what np.concatenate / np.matmul do to the float32 features when they meet the float64
 * adjacency (sampler2.py:760-770, fps_gcn_cpu.py:162-166); y1 may be NULL. */
int ssdr_widen_f32_f64_dev(const float* d_x, size_t n, double* d_y0, double* d_y1, void* stream);
/* One cloud's block of fps_adj_all (fps_gcn_cpu.py:40-117) for its superpoints d_sel[0..nsel): bbox centres
 * [nsel,3], directed chamfer means [nsel,nsel] (cd = dir + dir^T, create_cd :12-38) and the normalised adjacency
 * (S-I)D^-1 + I [nsel,nsel]; gcn_top > 0 keeps the top entries per row (:153-160).  max_sp_size >= largest
 * superpoint in d_sel. */
int ssdr_cloud_graph_dev(const float* d_xyz, const int32_t* d_sp_off, const int32_t* d_sp_pts, const int32_t* d_sel, size_t nsel,
                         size_t max_sp_size, int gcn_top, double* d_centres, double* d_cd_dir, double* d_adj, void* stream);
/* All clouds of a batch in one call: d_sel [n_total] lists the superpoints cloud by cloud, d_coff int32 [num_clouds+1]
 * gives each cloud's row range in d_sel / d_centres, d_boff int64 [num_clouds+1] the start of its n_c x n_c block in
 * d_cd_dir / d_adj (both of sum n_c^2 elements); n_max = largest n_c.  Same results as the per-cloud call. */
int ssdr_cloud_graph_batch_dev(const float* d_xyz, const int32_t* d_sp_off, const int32_t* d_sp_pts, const int32_t* d_sel,
                               const int32_t* d_coff, const int64_t* d_boff, size_t num_clouds, size_t n_total, size_t n_max,
                               int gcn_top, double* d_centres, double* d_cd_dir, double* d_adj, void* stream);
/* Arithmetic of the chamfer term in every selection-graph entry point: 0 = float64 (S3DIS: fps_gcn_cpu.py:12-38, the default), 1 = the Semantic3D code's
 * float32 CUDA-kernel values (SSRD_AL_semantic3d/fps_gcn_cuda.py:13-30: centred coordinates rounded to float32, chamfer3D.cu's squared distances,
 * sqrt and the two means in float32, widened).  Process-wide, read at every call. */
int ssdr_select_set_chamfer_mode(int mode);
int ssdr_propagate_batch_dev(const double* d_adj, const int32_t* d_coff, const int64_t* d_boff, size_t num_clouds, size_t n_max,
                             const int32_t* d_rows, const double* d_vin, int feat_dim, double* d_vout, double* d_comb, void* stream);
/* One hop of sum_i A^i V on a block (fps_gcn_cpu.py:162-167): vout[rows] = adj * vin[rows]; comb[rows] += vout[rows] */
int ssdr_propagate_dev(const double* d_adj, size_t n, const int32_t* d_rows, const double* d_vin, int feat_dim, double* d_vout,
                       double* d_comb, void* stream);
/* gcn.create_adj (S3/gcn.py:116-191): the adjacency of the trained-GCN branch (what GCN_sampling feeds its graph convolutions, gcn.py:193-263),
 * torch float32 in the reference: rows of d_feat [N,F] L2-normalised (eps 1e-12) -> d_out_v, cosine matrix times exp(-(ED + CD)) inside a
 * cloud's block and 0 between clouds, minus I, columns scaled by the inverse column sums, plus I -> d_out_adj [N,N].  The clouds' bbox centres
 * and directed chamfer means are those of ssdr_cloud_graph_batch_dev (same d_coff / d_boff); d_rows [N] gives, cloud by cloud, the row of
 * every member in the result (the reference's ref_idx: unlabelled candidates first, then the labelled regions).  N <= 32767. */
int ssdr_create_adj_dev(const float* d_feat, size_t N, int F, const double* d_centres, const double* d_cd_dir, const int32_t* d_coff, const int64_t* d_boff,
                        size_t num_clouds, size_t n_max, const int32_t* d_rows, float* d_out_v, float* d_out_adj, void* stream);
/* The candidate rule of sampling() + GCN_FPS_sampling as ONE enqueue-only call (S3/sampler2.py:533-552 create_file_top_and_all, :745-753 the
 * "first 2 x selected_num regions of a cloud" rule, :313-342 / :736-781 GCN_FPS_sampling; fps_gcn_cpu.py:60-178): the ranking goes in, the selected
 * candidates come out, and no host decision sits in between — the row counts the rule finds stay on the device and every kernel behind it reads them
 * there.  d_order [S] = the regions by descending uncertainty (ssdr_rank_regions_dev); d_labelled [S] != 0: the region is labelled and never competes;
 * the superpoints of cloud c are d_sp_base[c] .. d_sp_base[c+1]-1; d_lab_off [num_clouds+1] / d_lab_sp: the labelled regions cloud by cloud
 * (ascending superpoint id), n_lab of them in all.  batch_size = sampling_batch.  selector 0: farthest_features_sample from candidate `start`
 * (fps_gcn_cpu.py:119-147); 1: kCenterGreedy over candidates + labelled rows seeded with the labelled ones (kcenterGreedy.py:84-128).  Capacities the caller sizes the run by (upper bounds, computable from the static tables):
 * cap_rows >= candidates + labelled regions, cap_nmax >= the largest cloud's share of them, cap_sq >= the sum over clouds of its share squared,
 * cap_unl >= candidates (cap_rows <= 2^22: the reference's own round, 20 000 candidates + 4 000 labelled rows of 272 clouds, is one call), max_select = min(batch_size, unlabelled regions) = the number of picks.
 * d_result (int32): [0..7] n_unl, n_lab, rows, largest block, picks, status (bit 0: rows > cap_rows, bit 1: blocks > cap_sq: nothing was selected),
 * block elements (int64 in two words); [8 .. 8+max_select) the picks (indices into the candidate list); then [cap_rows] the candidate list
 * (superpoint ids, cloud by cloud, descending uncertainty inside a cloud) followed by the labelled regions.  feat_dim = 32.
 * d_cls / d_dom: predicted class per point / dominant predicted class per region — the candidates' dominant_point_ids (sampler2.py:625-626);
 * d_lab_cls / d_lab_dom (both or neither; NULL = the predicted pair): GROUND-TRUTH class per point / dominant ground-truth class per region
 * (ssdr_dominant_label_dev) — the labelled regions' dominant_point_ids (get_labeled_selection_cloudname_spidx_pointidx, sampler2.py:288-291). */
int ssdr_gcn_fps_sampling_dev(const float* d_feat, int feat_dim, const int32_t* d_cls, const int32_t* d_dom, const int32_t* d_lab_cls, const int32_t* d_lab_dom,
                              const float* d_xyz, const int32_t* d_sp_off, const int32_t* d_sp_pts, const int32_t* d_order, size_t S, const uint8_t* d_labelled, const int32_t* d_sp_base, size_t num_clouds,
                              const int32_t* d_lab_off, const int32_t* d_lab_sp, size_t n_lab, size_t batch_size, int gcn_number, int gcn_top, int selector, int start,
                              size_t cap_rows, size_t cap_nmax, size_t cap_sq, size_t cap_unl, size_t max_select, int32_t* d_result, void* stream);
/* The propagated rows of the last ssdr_gcn_fps_sampling_dev call on `stream`: device pointer to [cap_rows][32] float64, the candidates first, then the
 * labelled regions (sum_i A^i V, fps_gcn_cpu.py:162-167 — what GCN_FPS_sampling hands to farthest_features_sample, :169-170).  Valid until the
 * next selection call on that stream. */
int ssdr_gcn_fps_sampling_rows(void* stream, const double** d_rows, size_t* cap_rows);
/* The same for the SHARDED run (one process per GPU), again without a host decision: two enqueue-only calls around the all-gather of the
 * candidates' propagated features.  Global region id = rank * Smax + local id (padding counts as labelled), global cloud = rank * Bmax + local cloud.
 * ssdr_gcn_fps_sharded_local_dev: the candidate rule over the global ranking d_gorder [Sg = world * Smax] (ssdr_rank_regions_dev over the all-gathered masked
 * uncertainties) for the clouds of ALL ranks — d_gbase [world * Bmax + 1]: global cloud c spans d_gbase[c] .. d_gbase[c+1]-1 — then this rank's share of
 * GCN_FPS_sampling for its own clouds.  d_comb_out [nu_max (+ nl_max), 32] float64: its candidates' propagated features in candidate order (what the all-gather
 * sends; nu_max >= the candidates of any rank; nl_max = 0 for the FPS selector).  d_plan (int32, 16 + world + 2 * world * nu_max words): [0..7] as d_result above for this rank, [4] the picks
 * of all ranks, [8] the candidates of all ranks, [9] != 0: a rank offers more than nu_max; [16 .. 16 + world) candidates per rank; then the rows of the gathered
 * array in global candidate order; then the global candidate list (global region ids, rank by rank, cloud by cloud, descending uncertainty inside a cloud).
 * ssdr_fps_gathered_dev: d_gathered [world, nu_max, 32] -> the candidates' rows in order (d_glob [cap_rows, 32], cap_rows >= repeat x the candidates of all
 * ranks, which never exceed 2 x the picks) -> farthest_features_sample from candidate `start` (the replicated global FPS, fps_gcn_cpu.py:169-170); max_select
 * picks (indices into the global candidate list).  repeat > 1 (measurement only): the rows are taken `repeat` times — the chain's load at `repeat` x the ranks. */
int ssdr_gcn_fps_sharded_local_dev(const float* d_feat, int feat_dim, const int32_t* d_cls, const int32_t* d_dom, const int32_t* d_lab_cls, const int32_t* d_lab_dom,
                                   const float* d_xyz, const int32_t* d_sp_off, const int32_t* d_sp_pts, const int32_t* d_lab_off, const int32_t* d_lab_sp, size_t n_lab, size_t num_clouds,
                                   const int32_t* d_gorder, size_t Sg, const uint8_t* d_glabelled, const int32_t* d_gbase, int rank, int world, size_t Smax, size_t Bmax,
                                   size_t batch_size, int gcn_number, int gcn_top, size_t cap_rows, size_t cap_nmax, size_t cap_sq, size_t nu_max, size_t nl_max,
                                   double* d_comb_out, int32_t* d_plan, void* stream);
int ssdr_fps_gathered_dev(const double* d_gathered, const int32_t* d_plan, int world, size_t nu_max, size_t cap_rows, int repeat, int start, size_t max_select, double* d_glob,
                          int32_t* d_out, void* stream);
/* The k-center selector of the sharded run (BASELINE configuration 4: "a single all-gather of per-superpoint features before the global k-center step"; sampler2.py's
 * "kcenter" branch, kcenterGreedy.py:84-128): ssdr_gcn_fps_sharded_local_dev with nl_max > 0 (>= the labelled regions of any rank) also writes this rank's labelled
 * regions' propagated rows behind its candidates (d_comb_out [nu_max + nl_max, 32]); after the all-gather of those rows, ssdr_kcenter_gathered_dev builds
 * [every rank's candidates | every rank's labelled regions] in d_glob [cap_rows, 32] (cap_rows > n_lab_total + the candidates of all ranks), marks the labelled rows as
 * already selected (d_already [n_lab_total], scratch) and runs kCenterGreedy for max_select picks (indices into the global candidate list) — nothing is read back.
 * d_nlab_off [world + 1]: prefix of the ranks' labelled counts (static). */
int ssdr_kcenter_gathered_dev(const double* d_gathered, const int32_t* d_plan, int world, size_t nu_max, size_t nl_max, const int32_t* d_nlab_off, size_t n_lab_total,
                              size_t cap_rows, size_t max_select, double* d_glob, int32_t* d_already, int32_t* d_out, void* stream);
/* What the enqueue-only selection calls issued on `stream` found and could not return (bit 0: a cooperative multi-workgroup FPS / k-center
 * launch — ssdr_fps_dev / ssdr_kcenter_dev above 1536 / 4096 rows — was not co-resident: a workgroup waited for one that never arrived, the chain
 * stopped and its remaining picks read -1).  Waits for the stream; SSDR_ERR_INTERNAL when a bit is set; clears the word. */
int ssdr_select_status(void* stream, int32_t* out_status);
/* farthest_features_sample (fps_gcn_cpu.py:119-147); `start` is the reference's np.random.randint draw */
int ssdr_fps_dev(const double* d_feat, size_t n, int feat_dim, int start, size_t count, int32_t* d_out, void* stream);
/* farthest_superpoint_sample (sampler2.py:49-80, "edcd" branch) over one cloud's superpoints, from the centres and
 * directed chamfer means ssdr_cloud_graph_dev produced: distance = |centre_i - centre_c|^2 + CD(i,c); n <= 8192 */
int ssdr_fps_superpoint_dev(const double* d_centres, const double* d_cd_dir, size_t n, int start, size_t count, int32_t* d_out, void* stream);
/* kCenterGreedy.select_batch_ (kcenterGreedy.py:84-128) with direct float64 Euclidean distances */
int ssdr_kcenter_dev(const double* d_feat, size_t n, int feat_dim, const int32_t* d_already_selected, size_t n_already, size_t count,
                     int32_t* d_out, void* stream);

/* ---- chamfer_3D.forward (Semantic3D variant: SSRD_AL_semantic3d/chamfer3D/chamfer3D.cu:12-152, chamfer_cuda.cpp:30-33,
 *      dist_chamfer_3D.py:29-81).  xyz1 [b,n,3], xyz2 [b,m,3] -> squared distances dist1 [b,n], dist2 [b,m] and
 *      nearest indices idx1 [b,n] (into xyz2), idx2 [b,m] (into xyz1); fp32; lowest index on exact ties. */
int ssdr_chamfer3d_forward_dev(const float* d_xyz1, const float* d_xyz2, size_t batch, size_t n, size_t m,
                               float* d_dist1, float* d_dist2, int32_t* d_idx1, int32_t* d_idx2, void* stream);

/* ---- evaluation tail ("next" row N2: S3/RandLANet.py:326-334, 353-411; S3/helper_tool.py:237-262) ------------------
 * ssdr_vote_smooth_dev: test_probs[point_idx] = smooth*test_probs[point_idx] + (1-smooth)*probs with NumPy's rule for
 *   repeated indices (old row on the right-hand side, last occurrence wins).  d_owner_scratch: int32, one entry per
 *   row of test_probs, initialised to -1 by the caller once (the call restores it).
 * ssdr_confusion_dev: preds = argmax(probs[proj_idx[i]]) (proj_idx may be NULL: row i), confusion[label][pred] += 1
 *   accumulated into d_confusion uint64 [C,C] (caller zeroes it), optional d_pred int32 [n] and d_iou float64 [C] =
 *   DP.IoU_from_confusions of the accumulated matrix.  Re-projection to the raw cloud is proj_idx = ssdr_knn(sub_xyz,
 *   raw_xyz, K=1) (utils/data_prepare_s3dis.py:69). */
int ssdr_vote_smooth_dev(float* d_test_probs, const int32_t* d_point_idx, const float* d_probs, size_t n, int num_classes, double smooth,
                         int32_t* d_owner_scratch, void* stream);
int ssdr_confusion_dev(const float* d_probs, int num_classes, const int32_t* d_proj_idx, const int32_t* d_labels, size_t n, int32_t* d_pred,
                       uint64_t* d_confusion, double* d_iou, void* stream);

/* ---- plain device memory for callers without their own allocator (tests, the ctypes mirror) ---------- */
int ssdr_dev_alloc(size_t bytes, void** d_ptr);
int ssdr_dev_free(void* d_ptr);
int ssdr_memcpy_h2d(void* d_dst, const void* src, size_t bytes);
int ssdr_memcpy_d2h(void* dst, const void* d_src, size_t bytes);
/* the same, ordered on `stream` (NULL = library stream) and waiting for that stream only: a selection that runs on its own stream
 * uploads its index tables and reads its result back without waiting for what the other streams hold */
int ssdr_memcpy_h2d_on(void* d_dst, const void* src, size_t bytes, void* stream);
int ssdr_memcpy_d2h_on(void* dst, const void* d_src, size_t bytes, void* stream);

/* cpp_knn_batch_distance_pick (knn_.h:21-23, knn_.cxx:136-203, knn.pyx:111-149): nqueries "least used first" query points
 * per batch element and their K neighbours.  The reference seeds std::mt19937 with time(0); here the seed is an argument
 * (the same seed gives the reference's result).  batch_queries [B][nqueries][dim], batch_indices [B][nqueries][K]. */
int ssdr_knn_batch_distance_pick(const float* batch_data, size_t batch_size, size_t npts, size_t dim, float* batch_queries,
                                 size_t nqueries, size_t K, int64_t* batch_indices, uint32_t seed);

/* ---- superpoint-graph inputs (the partition itself, cut-pursuit, stays out of scope) -----------------
 * compute_graph_nn_2 (partition/graphs.py:23-70, voronoi == 0): sklearn's exact float64 k-NN.  source/target/distances
 * [n*k_nn1] of the adjacency graph (neighbour j of point i at i*k_nn1 + j, the point itself dropped), target2 [n*k_nn2].
 * Equidistant neighbours (sklearn leaves their order open) come in the order the kd-tree walk meets them. */
int ssdr_knn_graph_dev(const float* d_xyz, size_t n, size_t k_nn1, size_t k_nn2, uint32_t* d_source, uint32_t* d_target,
                       float* d_distances, uint32_t* d_target2, void* stream);
/* libply_c.compute_geof (partition/ply_c/ply_c.cpp:385-455): [n,4] = linearity, planarity, scattering, verticality of
 * every point and its k_nn neighbours d_target[n*k_nn]. */
int ssdr_geof_dev(const float* d_xyz, size_t n, const uint32_t* d_target, size_t k_nn, float* d_geof, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSDR_AL_H */
