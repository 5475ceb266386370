// C-ABI entry points of the KNN ops (include/ssdr_al.h), host and device flavours.
#include "ssdr_internal.hpp"
#include <random>
#include <map>

namespace ssdr {
namespace {

// One state per stream: calls on different streams own different forests / grids / scratch and may run concurrently.
// what a finished device-flavour call found, copied behind its kernels: [0..9] the grid counters, [10..12] the forest's status words
struct KnnTicket { hipEvent_t ev = nullptr; int32_t* host = nullptr; bool pending = false; };
constexpr int TICKETS = 16, TICKET_WORDS = 16;
struct KnnState {
    KdForest forest;
    GridForest grid;
    DevBuf pts, qry, out;
    KnnTicket ticket[TICKETS]; int next_ticket = 0;
    int32_t folded[4] = {0, 0, 0, 0};       // hand-over rows (K = 16, K = 1) of the last finished call, status bits so far, deepest tree
    ~KnnState() { for (auto& t : ticket) { if (t.ev) (void)hipEventDestroy(t.ev); if (t.host) (void)hipHostFree(t.host); } }
};
KnnState& st(hipStream_t s = nullptr) { return per_stream<KnnState>(s); }

// leaves the counters of the call just enqueued on s where ssdr_knn_status_poll finds them
int fold_ticket(KnnState& S, KnnTicket& t) {
    S.folded[0] = t.host[0]; S.folded[1] = t.host[1]; S.folded[2] |= t.host[2] | t.host[11]; S.folded[3] = t.host[12];
    t.pending = false;
    return SSDR_OK;
}
int leave_ticket(KnnState& S, hipStream_t s) {
    KnnTicket& t = S.ticket[S.next_ticket];
    S.next_ticket = (S.next_ticket + 1) % TICKETS;
    if (!t.ev) { SSDR_HIP(hipEventCreateWithFlags(&t.ev, hipEventDisableTiming)); SSDR_HIP(hipHostMalloc(reinterpret_cast<void**>(&t.host), 4 * TICKET_WORDS)); }
    if (t.pending) { SSDR_HIP(hipEventSynchronize(t.ev)); fold_ticket(S, t); }      // sixteen calls ago: long finished
    for (int i = 0; i < TICKET_WORDS; ++i) t.host[i] = 0;
    if (S.grid.need.p && S.grid.nsets > 0) SSDR_HIP(hipMemcpyAsync(t.host, S.grid.counters(), 4 * 10, hipMemcpyDeviceToHost, s));
    if (S.forest.counters.p) SSDR_HIP(hipMemcpyAsync(t.host + 10, S.forest.counters.p, 4 * 3, hipMemcpyDeviceToHost, s));
    SSDR_HIP(hipEventRecord(t.ev, s));
    t.pending = true;
    return SSDR_OK;
}

constexpr int GRID_TARGET_PTS = 22;      // the measured cell radius holds about this many points (K + 1 = 17 and a margin)

// grid search of a job table split as [jobs16 | jobs1], then the tree walk for the rows the grid handed over
int grid_knn(KnnState& S, const std::vector<GridDesc>& sets, const std::vector<GridJob>& jobs16, const std::vector<GridJob>& jobs1, bool i64, hipStream_t s) {
    // squared radius of a handed-over row's ball / its (K+1)-th squared distance (see the tree hand-over below); <= 0: complete trees at once
    const char* bs = getenv("SSDR_KNN_BALL_SCALE");
    const float ball_scale = bs ? (float)atof(bs) : 9.f;
    S.grid.ball_scale = ball_scale > 0.f ? ball_scale : 1.f;
    const char* tp = getenv("SSDR_KNN_TARGET_PTS");      // tuning knob of the cell size (development)
    SSDR_TRY(grid_build(S.grid, sets, tp ? std::max(1, atoi(tp)) : GRID_TARGET_PTS, s));
    std::vector<GridJob> jobs(jobs16); jobs.insert(jobs.end(), jobs1.begin(), jobs1.end());
    SSDR_TRY(grid_set_jobs(S.grid, jobs, s));
    int mq16 = 0, mq1 = 0;
    for (auto& j : jobs16) mq16 = std::max(mq16, j.nq);
    for (auto& j : jobs1) mq1 = std::max(mq1, j.nq);
    {
        double bytes = 0;      // algorithmic bytes: support + query coordinates read once, indices written once
        for (auto& j : jobs16) bytes += 24.0 * j.nq + (i64 ? 8.0 : 4.0) * 16 * j.nq;
        SSDR_TRY(grid_search(S.grid, 0, (int)jobs16.size(), mq16, 16, i64, s, bytes));
    }
    {
        double bytes = 0;
        for (auto& j : jobs1) bytes += 24.0 * j.nq + (i64 ? 8.0 : 4.0) * j.nq;
        bool fused = !jobs16.empty() && !jobs1.empty();       // every K = 1 job rides on a K = 16 scan: only the second pass runs for them
        for (auto& j : jobs16) fused = fused && j.job1 >= 0;
        SSDR_TRY(grid_search(S.grid, (int)jobs16.size(), (int)jobs1.size(), fused ? 0 : mq1, 1, i64, s, bytes));
    }
    // exact nanoflann trees only for the support sets that own handed-over rows (flags set by the searches above)
    std::vector<KdTreeDesc> trees(sets.size());
    for (size_t i = 0; i < sets.size(); ++i) { trees[i].pts = sets[i].pts; trees[i].n = sets[i].n; }
    ProfScope prof("knn_tree_handover", s, 0.0);
    // the trees are split only where the balls of the handed-over rows reach; a walk that leaves them (it may, while it has not found
    // K points yet) puts its row on the fall-back list, and those rows are answered on complete trees
    KdBalls balls{S.grid.ball_q(), S.grid.ball_tree(), S.grid.counters() + 6, GRID_BALL_CAP};
    const bool cut = ball_scale > 0.f;
    SSDR_TRY(kd_build(S.forest, trees, s, S.grid.need.as<int>(), cut ? &balls : nullptr));
    int* ctr = S.grid.counters();
    if (!jobs16.empty()) SSDR_TRY(kd_search_worklist(S.forest, S.grid.jobs.as<GridJob>(), S.grid.work_list(0), ctr + 0, S.grid.work_cap, 16, i64, s, S.grid.work_list(2), ctr + 8, S.grid.need2()));
    if (!jobs1.empty()) SSDR_TRY(kd_search_worklist(S.forest, S.grid.jobs.as<GridJob>(), S.grid.work_list(1), ctr + 1, S.grid.work_cap, 1, i64, s, S.grid.work_list(3), ctr + 9, S.grid.need2()));
    if (cut) {
        SSDR_TRY(kd_rebuild(S.forest, S.grid.need2(), s));
        if (!jobs16.empty()) SSDR_TRY(kd_search_worklist(S.forest, S.grid.jobs.as<GridJob>(), S.grid.work_list(2), ctr + 8, S.grid.work_cap, 16, i64, s));
        if (!jobs1.empty()) SSDR_TRY(kd_search_worklist(S.forest, S.grid.jobs.as<GridJob>(), S.grid.work_list(3), ctr + 9, S.grid.work_cap, 1, i64, s));
    }
    SSDR_TRY(leave_ticket(S, s));
    return SSDR_OK;
}

int check_knn_args(const void* p, size_t npts, size_t dim, const void* q, size_t nq, size_t K, const void* out) {
    if (dim != 3) { set_error("dim=%zu: only dim <= 3 is implemented (1 and 2 through the host entry points)", dim); return SSDR_ERR_UNSUPPORTED; }
    if ((!p && npts) || (!q && nq) || (!out && nq && K)) { set_error("NULL pointer argument"); return SSDR_ERR_INVALID; }
    if (npts > 0x3fffffff || nq > 0x3fffffff) { set_error("too many points"); return SSDR_ERR_INVALID; }
    return SSDR_OK;
}

// device-resident batch: B trees of npts points, nq queries each
int knn_batch_device(const float* d_pts, size_t B, size_t npts, const float* d_q, size_t nq, size_t K,
                     void* d_out, bool i64, bool self_order, hipStream_t s) {
    KnnState& S = st(s);
    if ((K == 16 || K == 1) && !getenv("SSDR_KNN_TREE_ONLY")) {
        std::vector<GridDesc> sets(B); std::vector<GridJob> jobs(B);
        const size_t esz = i64 ? 8 : 4;
        for (size_t b = 0; b < B; ++b) {
            sets[b] = GridDesc{}; sets[b].pts = d_pts + b * npts * 3; sets[b].n = (int)npts;
            jobs[b] = GridJob{(int)b, self_order ? (int)b : -1, (int)nq, -1, d_q + b * nq * 3, reinterpret_cast<char*>(d_out) + b * nq * K * esz, 0, 0};
        }
        return K == 16 ? grid_knn(S, sets, jobs, {}, i64, s) : grid_knn(S, sets, {}, jobs, i64, s);
    }
    std::vector<KdTreeDesc> trees(B);
    for (size_t b = 0; b < B; ++b) { trees[b].pts = d_pts + b * npts * 3; trees[b].n = (int)npts; }
    SSDR_TRY(kd_build(S.forest, trees, s));
    SSDR_TRY(kd_search(S.forest, 0, (int)B, d_q, nq * 3, (int)nq, (int)K, self_order ? 0 : -1, d_out, i64, nq * K, s));
    S.grid.nsets = 0;           // no grid in this call: the ticket carries the forest's words only
    return leave_ticket(S, s);
}

int knn_batch_host(const float* pts, size_t B, size_t npts, size_t dim, const float* q, size_t nq, size_t K,
                   void* out, bool i64) {
    if ((dim == 1 || dim == 2) && pts && q) {
        // knn_.cxx is dim-generic.  One and two dimensions are the 3-D problem with the missing coordinates at zero, EXACTLY: a zero-span dimension is never
        // the cut dimension (nanoflann.hpp:898-937: its span does not exceed (1 - 1e-5) x the largest, and with every span zero the cut stays on dimension 0 as it
        // does in the dim-generic code), the metric adds (0 - 0)^2 = +0 behind the other terms, the boxes and the far-branch bounds never see it.  dim > 3 is refused.
        std::vector<float> p3(B * npts * 3, 0.f), q3(B * nq * 3, 0.f);
        for (size_t i = 0; i < B * npts; ++i) for (size_t d = 0; d < dim; ++d) p3[3 * i + d] = pts[dim * i + d];
        for (size_t i = 0; i < B * nq; ++i) for (size_t d = 0; d < dim; ++d) q3[3 * i + d] = q[dim * i + d];
        return knn_batch_host(p3.data(), B, npts, 3, (pts == q && npts == nq) ? p3.data() : q3.data(), nq, K, out, i64);
    }
    SSDR_TRY(check_knn_args(pts, npts, dim, q, nq, K, out));
    SSDR_TRY(ensure_init());
    if (B == 0 || nq == 0 || K == 0) return SSDR_OK;
    Context& c = ctx(); hipStream_t s = c.stream; KnnState& S = st(s);
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

int ssdr_knn_status(void* stream, int32_t* out4) {
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream);
    KnnState& S = st(s);
    int32_t h[4] = {0, 0, 0, 0};
    SSDR_HIP(hipStreamSynchronize(s));
    // everything has finished: the earlier calls' tickets are folded first (their status bits must not be lost: the direct read below
    // only sees the counters of the NEWEST call), then the direct read
    for (int i = 0; i < TICKETS; ++i) if (S.ticket[i].pending) fold_ticket(S, S.ticket[i]);
    h[2] = S.folded[2];
    S.folded[2] = 0;
    if (S.grid.need.p && S.grid.nsets > 0) {
        int g[10] = {0};
        SSDR_HIP(hipMemcpy(g, S.grid.counters(), sizeof(g), hipMemcpyDeviceToHost));
        h[0] = g[0]; h[1] = g[1]; h[2] |= g[2];
        if (getenv("SSDR_KNN_DEBUG")) {
            fprintf(stderr, "knn grid: second pass for %d (K=16) + %d (K=1) rows; handed over to the tree %d + %d rows, %d of them unsettled; %d + %d again on complete trees\n", g[4], g[5], g[0], g[1], g[3], g[8], g[9]);
            std::vector<GridDesc> d(S.grid.nsets); std::vector<int> need(S.grid.nsets);
            SSDR_HIP(hipMemcpy(d.data(), S.grid.desc.p, sizeof(GridDesc) * d.size(), hipMemcpyDeviceToHost));
            SSDR_HIP(hipMemcpy(need.data(), S.grid.need.p, 4 * need.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < d.size(); i += std::max<size_t>(1, d.size() / 12))
                fprintf(stderr, "  set %zu: n=%d c=%.4f dims=%dx%dx%d cells=%d (%.1f pts/cell) need_tree=%d\n", i, d[i].n, d[i].c, d[i].nx, d[i].ny, d[i].nz, d[i].ncell,
                        (double)d[i].n / d[i].ncell, need[i]);
        }
    }
    if (S.forest.counters.p) {
        int k[3] = {0, 0, 0};
        SSDR_HIP(hipMemcpy(k, S.forest.counters.p, sizeof(k), hipMemcpyDeviceToHost));
        h[2] |= k[1]; h[3] = k[2];
    }
    if (out4) for (int i = 0; i < 4; ++i) out4[i] = h[i];
    if (h[2]) {
        set_error("KNN device status 0x%x (1 = kd queue overflow, 2 = kd node overflow, 4 = kd tree deeper than its level limit, 8 = hand-over list overflow): "
                  "the neighbour lists of the last call on this stream are not trustworthy", h[2]);
        return SSDR_ERR_INTERNAL;
    }
    return SSDR_OK;
}

int ssdr_knn_status_poll(void* stream, int32_t* out4) {
    SSDR_TRY(ensure_init());
    KnnState& S = st(pick_stream(stream));
    for (int i = 0; i < TICKETS; ++i) {
        KnnTicket& t = S.ticket[i];
        if (t.pending && hipEventQuery(t.ev) == hipSuccess) fold_ticket(S, t);
    }
    if (out4) for (int i = 0; i < 4; ++i) out4[i] = S.folded[i];
    if (S.folded[2]) {
        set_error("KNN device status 0x%x (1 = kd queue overflow, 2 = kd node overflow, 4 = kd tree deeper than its level limit, 8 = hand-over list overflow, "
                  "16 = walk into an unbuilt node): the neighbour lists of an earlier call on this stream are not trustworthy", S.folded[2]);
        S.folded[2] = 0;
        return SSDR_ERR_INTERNAL;
    }
    return SSDR_OK;
}

int ssdr_knn_batch_distance_pick(const float* batch_data, size_t batch_size, size_t npts, size_t dim, float* batch_queries, size_t nqueries,
                                 size_t K, int64_t* batch_indices, uint32_t seed) {
    if (dim != 3) { set_error("dim=%zu: only dim == 3 is implemented", dim); return SSDR_ERR_UNSUPPORTED; }
    if (!batch_data || !batch_queries || !batch_indices || npts == 0 || npts > 0x3fffffff) { set_error("knn_batch_distance_pick: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (batch_size == 0 || nqueries == 0 || K == 0) return SSDR_OK;
    Context& c = ctx(); hipStream_t s = c.stream; KnnState& S = st(s);
    std::mt19937 gen(seed);                                   // the reference: mt19937 mt_rand(time(0)), one draw per query (:141, :168)
    std::vector<uint32_t> rnd(batch_size * nqueries);
    for (auto& r : rnd) r = (uint32_t)gen();
    struct Scratch { DevBuf rnd, used, q, idx; ~Scratch() { rnd.release(); used.release(); q.release(); idx.release(); } } sc;      // freed on every return path
    DevBuf &d_rnd = sc.rnd, &d_used = sc.used, &d_q = sc.q, &d_idx = sc.idx;
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
    // set (l, b) = prefix N_l of tile b; stored level-major so one level's sets are contiguous
    KnnState& S = st(s);
    if (K == 16 && !getenv("SSDR_KNN_TREE_ONLY")) {
        std::vector<GridDesc> sets((num_layers + 1) * B);
        for (size_t l = 0; l <= num_layers; ++l)
            for (size_t b = 0; b < B; ++b) { GridDesc& d = sets[l * B + b]; d = GridDesc{}; d.pts = d_xyz + b * npts * 3; d.n = (int)N[l]; }
        std::vector<GridJob> j16, j1;
        for (size_t l = 0; l < num_layers; ++l)
            for (size_t b = 0; b < B; ++b) {
                // neigh_idx[l] = knn(xyz_l, xyz_l, K) (s3dis_dataset.py:165); interp_idx[l] = knn(xyz_{l+1}, xyz_l, 1) (:170)
                // the K = 1 job (table position num_layers * B + its own index) is answered inside the K = 16 scan where that is final
                j16.push_back(GridJob{(int)(l * B + b), (int)(l * B + b), (int)N[l], (int)(num_layers * B + j1.size()), d_xyz + b * npts * 3,
                                      d_neigh_idx[l] + b * N[l] * K, (int)N[l + 1], 0});
                j1.push_back(GridJob{(int)((l + 1) * B + b), (int)(l * B + b), (int)N[l], -1, d_xyz + b * npts * 3, d_interp_idx[l] + b * N[l], 0, 0});
            }
        SSDR_TRY(grid_knn(S, sets, j16, j1, false, s));
        for (size_t l = 0; l < num_layers; ++l)      // sub_idx[l] = neigh_idx[l][:, :N_{l+1}] (s3dis_dataset.py:168)
            if (d_sub_idx && d_sub_idx[l] && N[l + 1] > 0)
                SSDR_HIP(hipMemcpy2DAsync(d_sub_idx[l], N[l + 1] * K * 4, d_neigh_idx[l], N[l] * K * 4, N[l + 1] * K * 4, B, hipMemcpyDeviceToDevice, s));
        return SSDR_OK;
    }
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
    S.grid.nsets = 0;
    return leave_ticket(S, s);
}

int ssdr_knn_pyramid(const float* xyz, size_t B, size_t npts, size_t num_layers, const int32_t* ratios, size_t K,
                     int32_t* const* neigh_idx, int32_t* const* sub_idx, int32_t* const* interp_idx) {
    if (!xyz || !ratios || !neigh_idx || !interp_idx || num_layers == 0 || num_layers > 16) { set_error("bad pyramid arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    Context& c = ctx(); hipStream_t s = c.stream; KnnState& S = st(s);
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
