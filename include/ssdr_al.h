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
/* Milliseconds spent in the GPU part of the last host-flavour call (HIP events on the library stream). */
float ssdr_last_gpu_ms(void);

/* ---- KNN (replaces S3/utils/nearest_neighbors: knn_.h:4-26, knn_.cxx:22-135, knn.pyx:33-109) --
 * Exact K nearest neighbours of every query among `npts` support points, ascending squared
 * distance, with nanoflann v1.2.3's tie order (kd-tree traversal order, leaf size 10), i.e. the
 * int64 indices equal the reference's cpp_knn* output bit for bit.  dim must be 3.
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

/* ---- RandLA-Net inference (replaces the TF1 graph of S3/RandLANet.py:140-180, 505-585 run by
 *      model.sess.run([prob_logits, last_second_features, ...]) in S3/sampler2.py:598 / :327) ----------
 * fp32 throughout (exact-f32 MFMA).  Weights are handed over per layer with batch-norm already folded
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
void ssdr_randla_destroy(void* handle);
int  ssdr_randla_infer_dev(void* handle, size_t batch_size, size_t npts, const float* d_features, const float* d_xyz,
                           const int32_t* ratios, int32_t* const* d_neigh_idx, int32_t* const* d_interp_idx,
                           float* d_probs, float* d_feat32, void* stream);

/* ---- plain device memory for callers without their own allocator (tests, the ctypes mirror) ---------- */
int ssdr_dev_alloc(size_t bytes, void** d_ptr);
int ssdr_dev_free(void* d_ptr);
int ssdr_memcpy_h2d(void* d_dst, const void* src, size_t bytes);
int ssdr_memcpy_d2h(void* dst, const void* d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* SSDR_AL_H */
