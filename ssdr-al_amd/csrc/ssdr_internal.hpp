// Internal helpers shared by the HIP translation units of libssdr_al.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <string>
#include <vector>
#include <algorithm>
#include <map>
#include <mutex>
#include "../../include/ssdr_al.h"

#ifndef HIPEMU
#define SSDR_DYN_SHARED(type, name) extern __shared__ type name[]
#endif

// register budget of a kernel: at least n waves per SIMD (the allocator's own choice is sometimes one short of the next occupancy step)
#ifndef HIPEMU
#define SSDR_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#else
#define SSDR_WAVES_PER_EU(n)
#endif

namespace ssdr {

void set_error(const char* fmt, ...);

#define SSDR_HIP(call)                                                                        \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            ssdr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return SSDR_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define SSDR_TRY(call)                  \
    do {                                \
        int s__ = (call);               \
        if (s__ != SSDR_OK) return s__; \
    } while (0)

// A named, grow-only device buffer.  Workspaces are kept across calls so steady-state calls do not
// touch hipMalloc (Guideline 9: no allocation in the launch path once warm).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
    ~DevBuf() { release(); }
    int reserve(size_t bytes);
    void release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Context {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    int num_cu = 256;
};
Context& ctx();
int ensure_init();
inline hipStream_t pick_stream(void* s) { return s ? reinterpret_cast<hipStream_t>(s) : ctx().stream; }

// ---- per-stream scratch state --------------------------------------------------------------------
// Every *_dev entry point keeps its scratch (grow-only buffers, tickets, forests) per stream, so that calls on different streams run
// concurrently.  Host threads may make their first call on different streams at the same time (the reference calls knn_search from
// DataLoader workers): the lookup is serialised, the returned reference stays valid (node-based map) until ssdr_stream_destroy forgets
// the stream — which releases the state, so a recycled stream handle starts clean.  The maps live on the heap and are never destroyed:
// their buffers must not be freed after the runtime has shut down.
void register_stream_forgetter(void (*f)(hipStream_t));
void forget_stream(hipStream_t s);          // context.hip: called by ssdr_stream_destroy
template <class T>
T& per_stream(hipStream_t s) {
    static std::mutex mu;
    static std::map<hipStream_t, T>* m = nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!m) {
        m = new std::map<hipStream_t, T>;
        register_stream_forgetter([](hipStream_t st) { std::lock_guard<std::mutex> lk2(mu); m->erase(st); });
    }
    return (*m)[s ? s : ctx().stream];
}

// ---- in-library kernel timing (context.hip): HIP events on the launch stream around instrumented launches.
// Off by default; bench.py switches it on for the roofline leg.  `work` is the launch's algorithmic FLOPs or bytes, `work2` the
// FLOPs its MFMA instructions execute (padding, both product orientations and the three split-bf16 products included).
struct ProfScope {
    int slot = -1;
    ProfScope(const char* name, hipStream_t s, double work, double work2 = 0.0);      // work2: FLOPs the matrix cores execute (MFMA kernels)
    ~ProfScope();
    hipStream_t stream = nullptr;
};

// ---- radix sort (radix_sort.hip) ----------------------------------------------------------------
// Stable sort of (u64 key, u32 value) pairs, in place (result in keys/vals).  d_n (optional) is a device
// int holding the live count (<= n_host); n_host sizes the launches.
constexpr int RADIX_TILE = 2048;       // pairs per sort tile; segment offsets must be multiples of it
constexpr int RADIX_MAX_SEG = 64;
struct RadixSorter {
    DevBuf k1, v1, hist, andor;
    int nblocks_max = 0;
    bool wide_high = false;      // keys whose bits above 32 vary (a ranking by float value): wide passes up there too when the set is large, not the one-workgroup kernel
    int reserve(size_t slots);
    // key_bits: upper bound on the significant key bits when the caller knows one (skips launching higher passes)
    // input_in_alt: the caller wrote the unsorted pairs into alt_keys() / alt_vals() (valid after reserve()) instead of keys / vals;
    // with an odd number of executed digit passes the result then lands in keys / vals without the final copy
    int sort(uint64_t* keys, uint32_t* vals, int n_host, const int* d_n, hipStream_t s, int key_bits = 64, bool input_in_alt = false);
    uint64_t* alt_keys() { return k1.as<uint64_t>(); }
    uint32_t* alt_vals() { return v1.as<uint32_t>(); }
    // nseg independent segments in the same launches: segment i lives in slots [off[i], off[i+1]) (tile-aligned), holds
    // n_host[i] pairs (d_cnt[i] when given, device side).  off has nseg + 1 entries.
    // d_andor (optional, device, [nseg][2]): AND and OR of every segment's keys when the producer knows them (any AND' subset of the true AND and
    // OR' superset of the true OR will do: a digit is skipped when AND' and OR' agree on it) — saves the pass that reads all keys to find out
    // vals == nullptr: keys only.  shift: the sort key is bits [shift, shift + key_bits) of the word — whatever sits below travels with it
    // (a stable sort of (key << b | index) words needs no value array at all).
    int sort_segments(uint64_t* keys, uint32_t* vals, int nseg, const int* off, const int* n_host, const int* d_cnt, hipStream_t s, int key_bits = 64,
                      bool input_in_alt = false, const unsigned long long* d_andor = nullptr, int shift = 0);
};

// ---- kd-tree forest (kdtree.hip) ----------------------------------------------------------------
// Descriptor tables travel to the device through pinned staging buffers.  A RING of them (an event per slot guards its reuse): with ONE buffer the host
// waited, at every call, for the previous call's copy to have executed on the stream — i.e. for the stream to have come that far — and a caller that keeps
// several stages in flight (pipeline.ALRound: 43 of 48 ms of host time inside the pyramid's enqueue) was throttled to the GPU's progress on that stream.
template <class T> struct StagingRing {
    static constexpr int SLOTS = 4;
    T* buf[SLOTS] = {nullptr, nullptr, nullptr, nullptr}; size_t cap[SLOTS] = {0, 0, 0, 0}; hipEvent_t ev[SLOTS] = {nullptr, nullptr, nullptr, nullptr}; int next = 0;
    // the slot to fill for `n` entries (waits only if the copy of SLOTS calls ago has not executed yet)
    int acquire(size_t n, T** out) {
        const int i = next; next = (next + 1) % SLOTS;
        if (cap[i] < n) {
            if (ev[i]) { if (hipEventSynchronize(ev[i]) != hipSuccess) return -1; }
            if (buf[i]) (void)hipHostFree(buf[i]);
            cap[i] = n * 2;
            if (hipHostMalloc(reinterpret_cast<void**>(&buf[i]), sizeof(T) * cap[i]) != hipSuccess) { buf[i] = nullptr; cap[i] = 0; return -1; }
        }
        if (!ev[i]) { if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return -1; }
        else if (hipEventSynchronize(ev[i]) != hipSuccess) return -1;
        *out = buf[i];
        return i;
    }
    int release(int i, hipStream_t s) { return hipEventRecord(ev[i], s) == hipSuccess ? 0 : -1; }
    ~StagingRing() { for (int i = 0; i < SLOTS; ++i) { if (buf[i]) (void)hipHostFree(buf[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); } }
};

struct KdTreeDesc {      // one independent support set
    const float* pts;    // device pointer, n x 3 row-major
    int n;
    int voff;            // offset of this tree's slice in `sorted` (position-ordered point records: x, y, z, index)
    int root;            // node id of the root
    float lo[3], hi[3];  // tight root box
};

struct KdForest {
    DevBuf desc, sorted, node_a, node_box, node_tree, queue, counters, tmp;     // node_a: 32-byte records {int4 range+children; float4 divlow, divhigh, dim, -}
    StagingRing<KdTreeDesc> staging;
    int ntrees = 0;
    int total_pts = 0;
    int node_cap = 0;
    int queue_cap = 0;
    int max_n = 0;
    KdForest() = default;
    KdForest(const KdForest&) = delete;
    KdForest& operator=(const KdForest&) = delete;

};

// Queries that will walk the forest, known before it is built: (x, y, z, r^2) and the tree each belongs to.  A build that is given
// them splits only the nodes whose box one of the balls reaches; the other nodes stay closed, and a walk that wants to enter a
// closed node reports its row instead of answering it (kd_search_worklist's fall-back list).
struct KdBalls { const float4* q = nullptr; const int* tree = nullptr; const int* count = nullptr; int cap = 0; };

// Builds `ntrees` trees described by host descriptors (pts/n filled in; voff/root/lo/hi are computed).
// d_need (optional, device int[ntrees]): only trees whose flag is non-zero when the build runs are built.
// balls (optional): see KdBalls; more than balls->cap balls: everything is built.
int kd_build(KdForest& f, const std::vector<KdTreeDesc>& trees, hipStream_t s, const int* d_need = nullptr, const KdBalls* balls = nullptr);
// Builds the flagged trees of the forest of the last kd_build again, completely.
int kd_rebuild(KdForest& f, const int* d_need, hipStream_t s);
// For every tree t in [tree0, tree0+ntrees): queries q = d_queries + (t-tree0)*q_stride floats, nq each.
// qorder_tree >= 0: visit queries in the vind order of forest tree (qorder_tree0 + (t-tree0)) (must have n == nq).
// out: int32 or int64 [ntrees][nq][K].
int kd_search(const KdForest& f, int tree0, int ntrees, const float* d_queries, size_t q_stride, int nq, int K,
              int qorder_tree0, void* d_out, bool out_i64, size_t out_stride, hipStream_t s);
// knn_.cxx:136-203 on a built forest of ntrees trees of npts points each: d_rnd holds ntrees*nq std::mt19937 draws
int kd_distance_pick(const KdForest& f, int ntrees, int npts, const uint32_t* d_rnd, int nq, int K, int* d_used, float* d_out_q,
                     int64_t* d_out_idx, hipStream_t s);
// float64 distances / bounds (sklearn semantics, partition/graphs.py): ids by ascending distance + squared distances
int kd_search_f64(const KdForest& f, int tree0, int ntrees, const float* d_queries, size_t q_stride, int nq, int K,
                  int qorder_tree0, int32_t* d_out, double* d_out_d2, size_t out_stride, hipStream_t s);
// Reads back the device status flags of the forest (synchronises the stream).
int kd_check(const KdForest& f, hipStream_t s);

// ---- uniform-grid search with hand-over to the kd walk (knn_grid.hip) -----------------------------------------------
struct GridDesc {        // one support set
    const float* pts;    // device pointer, n x 3 row-major
    int n;
    int pt_off;          // offset of this set's slice in `sorted` / `rank`
    int cell_off;        // offset of its cell table (ncell + 1 ints)
    int cell_cap;        // most cells the table may hold
    float c, inv_c;      // cell size (measured on the device)
    int nx, ny, nz, ncell;
    float lo[3], hi[3];  // bounding box; lo is the grid origin
    float slack;         // absolute bound on the rounding of a face position lo + k c against a point's binning floor((v - lo) / c): both carry
                         // a few ulps of the coordinates' magnitude and of k c, whatever the size of the gap to the face itself
    int pad_;
};
struct GridJob {         // one search: queries of one batch element against one support set, one output block [nq][K]
    int sup;             // support set
    int ord;             // >= 0: the queries are the points of this set and are taken in its cell order; < 0: natural order
    int nq;
    int job1;            // K = 16 self-searches of a pyramid: the K = 1 job answered in the same scan (-1: none).  That job's support set
                         // must be the first n1 points of this job's support set (tf_map's prefix sub-sampling) and its queries the same
    const float* qpts;   // query coordinates [nq][3]
    void* out;           // [nq][K] int32 / int64
    int n1; int pad;
};
constexpr int GRID_BALL_CAP = 2048;
struct GridForest {
    DevBuf desc, cell, rank, sorted, bsum, work, need, jobs, balls;
    StagingRing<GridDesc> staging; StagingRing<GridJob> jstaging;
    int nsets = 0, total_pts = 0, max_n = 0, max_blk = 0, work_cap = 0;
    // device ints behind the per-set `need` flags: [0], [1] hand-over list lengths (K = 16, K = 1), [2] status, [3] unsettled rows,
    // [4], [5] retry list lengths, [6] balls, [8], [9] rows the partly built trees could not answer (fall-back lists);
    // work_list(0..1): hand-over lists, work_list(2..3): retry lists, afterwards the fall-back lists
    int* counters() const { return need.as<int>() + nsets; }
    int* need2() const { return need.as<int>() + nsets + 16; }          // per set: its tree must be built completely
    float4* ball_q() const { return balls.as<float4>(); }
    int* ball_tree() const { return reinterpret_cast<int*>(balls.as<float4>() + GRID_BALL_CAP); }
    float ball_scale = 9.f;     // squared radius of a hand-over row's ball / its (K+1)-th squared distance
    int* work_list(int which) const { return work.as<int>() + (size_t)which * 2 * work_cap; }
    GridForest() = default;
    GridForest(const GridForest&) = delete;
    GridForest& operator=(const GridForest&) = delete;
    ~GridForest() {
    }
};
// Bins every set (pts / n filled in by the caller).  target_pts: number of points the measured cell radius should hold.
int grid_build(GridForest& g, const std::vector<GridDesc>& sets, int target_pts, hipStream_t s);
int grid_set_jobs(GridForest& g, const std::vector<GridJob>& jobs, hipStream_t s);
int grid_search(const GridForest& g, int job0, int njobs, int max_nq, int K, bool out_i64, hipStream_t s, double prof_bytes = 0.0);
// Answers the rows of a work list (pairs job id, query) by the exact tree walk; tree ids == set ids.  d_fallback (optional): rows whose
// walk met a closed node of a partly built tree are appended there (and their tree flagged in d_need2) instead of being written.
int kd_search_worklist(const KdForest& f, const GridJob* d_jobs, const int* d_work, const int* d_count, int work_cap, int K, bool out_i64, hipStream_t s,
                       int* d_fallback = nullptr, int* d_fallback_count = nullptr, int* d_need2 = nullptr);

}  // namespace ssdr
