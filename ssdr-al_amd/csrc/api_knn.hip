// C-ABI entry points of the KNN ops (include/ssdr_al.h), host and device flavours.
#include "ssdr_internal.hpp"
#include <random>

namespace ssdr {
namespace {

struct KnnState {
    KdForest forest;
    DevBuf pts, qry, out;
};
KnnState& st() { static KnnState s; return s; }

int check_knn_args(const void* p, size_t npts, size_t dim, const void* q, size_t nq, size_t K, const void* out) {
    if (dim != 3) { set_error("dim=%zu: only dim == 3 is implemented", dim); return SSDR_ERR_UNSUPPORTED; }
    if ((!p && npts) || (!q && nq) || (!out && nq && K)) { set_error("NULL pointer argument"); return SSDR_ERR_INVALID; }
    if (npts > 0x3fffffff || nq > 0x3fffffff) { set_error("too many points"); return SSDR_ERR_INVALID; }
    return SSDR_OK;
}

// device-resident batch: B trees of npts points, nq queries each
int knn_batch_device(const float* d_pts, size_t B, size_t npts, const float* d_q, size_t nq, size_t K,
                     void* d_out, bool i64, bool self_order, hipStream_t s) {
    KnnState& S = st();
    std::vector<KdTreeDesc> trees(B);
    for (size_t b = 0; b < B; ++b) { trees[b].pts = d_pts + b * npts * 3; trees[b].n = (int)npts; }
    SSDR_TRY(kd_build(S.forest, trees, s));
    SSDR_TRY(kd_search(S.forest, 0, (int)B, d_q, nq * 3, (int)nq, (int)K, self_order ? 0 : -1, d_out, i64, nq * K, s));
    return SSDR_OK;
}

int knn_batch_host(const float* pts, size_t B, size_t npts, size_t dim, const float* q, size_t nq, size_t K,
                   void* out, bool i64) {
    SSDR_TRY(check_knn_args(pts, npts, dim, q, nq, K, out));
    SSDR_TRY(ensure_init());
    if (B == 0 || nq == 0 || K == 0) return SSDR_OK;
    KnnState& S = st(); Context& c = ctx(); hipStream_t s = c.stream;
    const size_t esz = i64 ? 8 : 4;
    SSDR_TRY(S.pts.reserve(B * npts * 12 + 16)); SSDR_TRY(S.qry.reserve(B * nq * 12 + 16)); SSDR_TRY(S.out.reserve(B * nq * K * esz + 16));
    const bool same = (pts == q && npts == nq);
    SSDR_HIP(hipMemcpyAsync(S.pts.p, pts, B * npts * 12, hipMemcpyHostToDevice, s));
    const float* d_q = S.pts.as<float>();
    if (!same) { SSDR_HIP(hipMemcpyAsync(S.qry.p, q, B * nq * 12, hipMemcpyHostToDevice, s)); d_q = S.qry.as<float>(); }
    SSDR_HIP(hipEventRecord(c.ev0, s));
    SSDR_TRY(knn_batch_device(S.pts.as<float>(), B, npts, d_q, nq, K, S.out.p, i64, same, s));
    SSDR_HIP(hipEventRecord(c.ev1, s));
    SSDR_HIP(hipMemcpyAsync(out, S.out.p, B * nq * K * esz, hipMemcpyDeviceToHost, s));
    SSDR_TRY(kd_check(S.forest, s));
    SSDR_HIP(hipEventElapsedTime(&c.last_ms, c.ev0, c.ev1));
    return SSDR_OK;
}

}  // namespace
}  // namespace ssdr

using namespace ssdr;

extern "C" {

int ssdr_knn(const float* points, size_t npts, size_t dim, const float* queries, size_t nqueries, size_t K, int64_t* indices) {
    return knn_batch_host(points, 1, npts, dim, queries, nqueries, K, indices, true);
}
int ssdr_knn_batch(const float* batch_data, size_t batch_size, size_t npts, size_t dim, const float* queries,
                   size_t nqueries, size_t K, int64_t* batch_indices) {
    return knn_batch_host(batch_data, batch_size, npts, dim, queries, nqueries, K, batch_indices, true);
}
int ssdr_knn_batch_i32(const float* batch_data, size_t batch_size, size_t npts, size_t dim, const float* queries,
                       size_t nqueries, size_t K, int32_t* batch_indices) {
    return knn_batch_host(batch_data, batch_size, npts, dim, queries, nqueries, K, batch_indices, false);
}
int ssdr_knn_batch_dev(const float* d_batch_data, size_t batch_size, size_t npts, size_t dim, const float* d_queries,
                       size_t nqueries, size_t K, int32_t* d_indices, void* stream) {
    SSDR_TRY(check_knn_args(d_batch_data, npts, dim, d_queries, nqueries, K, d_indices));
    SSDR_TRY(ensure_init());
    if (batch_size == 0 || nqueries == 0 || K == 0) return SSDR_OK;
    return knn_batch_device(d_batch_data, batch_size, npts, d_queries, nqueries, K, d_indices, false,
                            d_batch_data == d_queries && npts == nqueries, pick_stream(stream));
}

int ssdr_knn_batch_distance_pick(const float* batch_data, size_t batch_size, size_t npts, size_t dim, float* batch_queries, size_t nqueries,
                                 size_t K, int64_t* batch_indices, uint32_t seed) {
    if (dim != 3) { set_error("dim=%zu: only dim == 3 is implemented", dim); return SSDR_ERR_UNSUPPORTED; }
    if (!batch_data || !batch_queries || !batch_indices || npts == 0 || npts > 0x3fffffff) { set_error("knn_batch_distance_pick: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (batch_size == 0 || nqueries == 0 || K == 0) return SSDR_OK;
    KnnState& S = st(); Context& c = ctx(); hipStream_t s = c.stream;
    std::mt19937 gen(seed);                                   // the reference: mt19937 mt_rand(time(0)), one draw per query (:141, :168)
    std::vector<uint32_t> rnd(batch_size * nqueries);
    for (auto& r : rnd) r = (uint32_t)gen();
    DevBuf d_rnd, d_used, d_q, d_idx;
    SSDR_TRY(S.pts.reserve(batch_size * npts * 12 + 16));
    SSDR_TRY(d_rnd.reserve(4 * rnd.size())); SSDR_TRY(d_used.reserve(4 * batch_size * npts));
    SSDR_TRY(d_q.reserve(12 * batch_size * nqueries)); SSDR_TRY(d_idx.reserve(8 * batch_size * nqueries * K));
    SSDR_HIP(hipMemcpyAsync(S.pts.p, batch_data, batch_size * npts * 12, hipMemcpyHostToDevice, s));
    SSDR_HIP(hipMemcpyAsync(d_rnd.p, rnd.data(), 4 * rnd.size(), hipMemcpyHostToDevice, s));
    std::vector<KdTreeDesc> trees(batch_size);
    for (size_t b = 0; b < batch_size; ++b) { trees[b].pts = S.pts.as<float>() + b * npts * 3; trees[b].n = (int)npts; }
    SSDR_TRY(kd_build(S.forest, trees, s));
    SSDR_TRY(kd_distance_pick(S.forest, (int)batch_size, (int)npts, d_rnd.as<uint32_t>(), (int)nqueries, (int)K, d_used.as<int>(), d_q.as<float>(),
                              d_idx.as<int64_t>(), s));
    SSDR_HIP(hipMemcpyAsync(batch_queries, d_q.p, 12 * batch_size * nqueries, hipMemcpyDeviceToHost, s));
    SSDR_HIP(hipMemcpyAsync(batch_indices, d_idx.p, 8 * batch_size * nqueries * K, hipMemcpyDeviceToHost, s));
    SSDR_HIP(hipStreamSynchronize(s));
    d_rnd.release(); d_used.release(); d_q.release(); d_idx.release();
    return kd_check(S.forest, s);
}

int ssdr_knn_pyramid_dev(const float* d_xyz, size_t B, size_t npts, size_t num_layers, const int32_t* ratios, size_t K,
                         int32_t* const* d_neigh_idx, int32_t* const* d_sub_idx, int32_t* const* d_interp_idx, void* stream) {
    if (!d_xyz || !ratios || !d_neigh_idx || !d_interp_idx || num_layers == 0 || num_layers > 16) { set_error("bad pyramid arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream);
    std::vector<size_t> N(num_layers + 1); N[0] = npts;
    for (size_t l = 0; l < num_layers; ++l) {
        if (ratios[l] <= 0) { set_error("ratio must be positive"); return SSDR_ERR_INVALID; }
        N[l + 1] = N[l] / (size_t)ratios[l];
    }
    // tree (l, b) = prefix N_l of tile b; stored level-major so one level's trees are contiguous
    KnnState& S = st();
    std::vector<KdTreeDesc> trees((num_layers + 1) * B);
    for (size_t l = 0; l <= num_layers; ++l)
        for (size_t b = 0; b < B; ++b) { trees[l * B + b].pts = d_xyz + b * npts * 3; trees[l * B + b].n = (int)N[l]; }
    SSDR_TRY(kd_build(S.forest, trees, s));
    for (size_t l = 0; l < num_layers; ++l) {
        // neigh_idx[l] = knn(xyz_l, xyz_l, K)          (s3dis_dataset.py:165)
        SSDR_TRY(kd_search(S.forest, (int)(l * B), (int)B, d_xyz, npts * 3, (int)N[l], (int)K, (int)(l * B),
                           d_neigh_idx[l], false, N[l] * K, s));
        // interp_idx[l] = knn(xyz_{l+1}, xyz_l, 1)     (s3dis_dataset.py:170)
        SSDR_TRY(kd_search(S.forest, (int)((l + 1) * B), (int)B, d_xyz, npts * 3, (int)N[l], 1, (int)(l * B),
                           d_interp_idx[l], false, N[l], s));
        // sub_idx[l] = neigh_idx[l][:, :N_{l+1}]       (s3dis_dataset.py:168)
        if (d_sub_idx && d_sub_idx[l] && N[l + 1] > 0)
            SSDR_HIP(hipMemcpy2DAsync(d_sub_idx[l], N[l + 1] * K * 4, d_neigh_idx[l], N[l] * K * 4, N[l + 1] * K * 4, B,
                                      hipMemcpyDeviceToDevice, s));
    }
    return SSDR_OK;
}

int ssdr_knn_pyramid(const float* xyz, size_t B, size_t npts, size_t num_layers, const int32_t* ratios, size_t K,
                     int32_t* const* neigh_idx, int32_t* const* sub_idx, int32_t* const* interp_idx) {
    if (!xyz || !ratios || !neigh_idx || !interp_idx || num_layers == 0 || num_layers > 16) { set_error("bad pyramid arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    Context& c = ctx(); hipStream_t s = c.stream; KnnState& S = st();
    std::vector<size_t> N(num_layers + 1); N[0] = npts;
    for (size_t l = 0; l < num_layers; ++l) { if (ratios[l] <= 0) { set_error("ratio must be positive"); return SSDR_ERR_INVALID; } N[l + 1] = N[l] / (size_t)ratios[l]; }
    size_t tot = 0; std::vector<size_t> off_n(num_layers), off_i(num_layers);
    for (size_t l = 0; l < num_layers; ++l) { off_n[l] = tot; tot += B * N[l] * K; off_i[l] = tot; tot += B * N[l]; }
    SSDR_TRY(S.pts.reserve(B * npts * 12 + 16)); SSDR_TRY(S.out.reserve(tot * 4 + 16));
    SSDR_HIP(hipMemcpyAsync(S.pts.p, xyz, B * npts * 12, hipMemcpyHostToDevice, s));
    std::vector<int32_t*> dn(num_layers), di(num_layers);
    for (size_t l = 0; l < num_layers; ++l) { dn[l] = S.out.as<int32_t>() + off_n[l]; di[l] = S.out.as<int32_t>() + off_i[l]; }
    SSDR_HIP(hipEventRecord(c.ev0, s));
    SSDR_TRY(ssdr_knn_pyramid_dev(S.pts.as<float>(), B, npts, num_layers, ratios, K, dn.data(), nullptr, di.data(), s));
    SSDR_HIP(hipEventRecord(c.ev1, s));
    for (size_t l = 0; l < num_layers; ++l) {
        SSDR_HIP(hipMemcpyAsync(neigh_idx[l], dn[l], B * N[l] * K * 4, hipMemcpyDeviceToHost, s));
        SSDR_HIP(hipMemcpyAsync(interp_idx[l], di[l], B * N[l] * 4, hipMemcpyDeviceToHost, s));
        if (sub_idx && sub_idx[l] && N[l + 1] > 0)
            SSDR_HIP(hipMemcpy2DAsync(sub_idx[l], N[l + 1] * K * 4, dn[l], N[l] * K * 4, N[l + 1] * K * 4, B, hipMemcpyDeviceToHost, s));
    }
    SSDR_TRY(kd_check(S.forest, s));
    SSDR_HIP(hipEventElapsedTime(&c.last_ms, c.ev0, c.ev1));
    return SSDR_OK;
}

}
