// The Semantic3D sampling loader's partition of a whole cloud into the parts the network is fed with
// (SSRD_AL_semantic3d/semantic3d_dataset_sampling.py:198-253): split3 halves the cloud along x and y at the middle of its bounding box —
// and, as written, puts EVERY point on the first side of z (the test is `z < z_max + 0.5 * z_len`, :224, true for all points: reproduced,
// not fixed) — appends the eight parts in (x, y, z) order, recursing into a part of more than max_size points; tf_map then folds every
// part of at most merge_max (2000) points into the one before it (:243-251).
//
// Arithmetic as NumPy does it: the bounds are float32 minima / maxima widened to Python floats, `x_min + 0.5 * x_len` is float64, and the
// comparison of the float32 column with that Python float runs in float32 (the scalar is cast: NumPy 1.16's value-based rule and NumPy
// 2's weak scalars agree).  The recursion tree is small (a handful of nodes), so it lives on the host: per level one pass for the open
// nodes' boxes and one that moves their points to the children; the grouping of the points by part is a stable radix sort of the leaves'
// depth-first ranks.  Inside a part the points keep ascending index: the reference's own order is the iteration order of a CPython
// set of ints (`list(x & y & z)`, :232) — a property of the interpreter's hash table, not of the algorithm.
#include "ssdr_internal.hpp"

namespace ssdr {
namespace {

constexpr int S3_OPEN_MAX = 64;      // open (splitting) nodes of one level

__device__ __forceinline__ unsigned f2o(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }      // order-preserving
inline float o2f(unsigned o) {          // (host: the boxes come back as ordered integers)
    const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    union { unsigned u; float f; } c; c.u = u; return c.f;
}

// box[slot][0..3] = min x, max x, min y, max y of the points whose node is open (slot_of[node] >= 0), as ordered integers
__global__ __launch_bounds__(256) void s3_boxes(const float* __restrict__ xyz, const int* __restrict__ node, int n, const int* __restrict__ slot_of, int nopen, unsigned* box) {
    __shared__ unsigned s_b[S3_OPEN_MAX * 4];
    for (int i = threadIdx.x; i < nopen * 4; i += 256) s_b[i] = (i & 1) ? 0u : 0xffffffffu;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int sl = slot_of[node[i]];
        if (sl < 0) continue;
        const unsigned x = f2o(xyz[3 * (size_t)i]), y = f2o(xyz[3 * (size_t)i + 1]);
        atomicMin(&s_b[4 * sl], x); atomicMax(&s_b[4 * sl + 1], x); atomicMin(&s_b[4 * sl + 2], y); atomicMax(&s_b[4 * sl + 3], y);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nopen * 4; i += 256) { if (i & 1) atomicMax(&box[i], s_b[i]); else atomicMin(&box[i], s_b[i]); }
}

// points of open nodes move to child (x side) * 2 + (y side) of their node; cnt[slot][4] counts the children
__global__ __launch_bounds__(256) void s3_assign(const float* __restrict__ xyz, int* node, int n, const int* __restrict__ slot_of, int nopen, const float* __restrict__ thr,
                                                 const int* __restrict__ child_base, int* cnt) {
    __shared__ int s_c[S3_OPEN_MAX * 4];
    for (int i = threadIdx.x; i < nopen * 4; i += 256) s_c[i] = 0;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int sl = slot_of[node[i]];
        if (sl < 0) continue;
        const int q = (xyz[3 * (size_t)i] < thr[2 * sl] ? 0 : 2) + (xyz[3 * (size_t)i + 1] < thr[2 * sl + 1] ? 0 : 1);
        node[i] = child_base[sl] + q;
        atomicAdd(&s_c[4 * sl + q], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nopen * 4; i += 256) if (s_c[i]) atomicAdd(&cnt[i], s_c[i]);
}

__global__ __launch_bounds__(256) void s3_keys(const int* __restrict__ node, int n, const int* __restrict__ rank_of, const int* __restrict__ part_of, uint64_t* keys, uint32_t* vals, int* part) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int nd = node[i];
        keys[i] = (uint64_t)(unsigned)rank_of[nd]; vals[i] = (uint32_t)i;
        if (part) part[i] = part_of[nd];
    }
}
__global__ __launch_bounds__(256) void s3_fill(int* p, int n, int v) { for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = v; }

struct S3State { DevBuf node, tab, keys, vals; RadixSorter sorter; };

struct S3Node { int parent_slot; long count; int child[4]; bool split; };

}  // namespace
}  // namespace ssdr

using namespace ssdr;

extern "C" int ssdr_split3_dev(const float* d_xyz, size_t n, size_t max_size, size_t recurse_max_size, size_t merge_max, int32_t* d_part, int32_t* d_order, int64_t* part_offsets,
                               size_t max_parts, size_t* num_parts, void* stream) {
    if (!d_xyz || !d_order || !part_offsets || !num_parts || n == 0 || n > 0x3fffffff || max_size == 0 || recurse_max_size == 0 || max_parts == 0) { set_error("split3: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream); S3State& Q = per_stream<S3State>(s);
    const int N = (int)n, G = std::max(1, std::min((N + 255) / 256, ctx().num_cu * 8));
    SSDR_TRY(Q.node.reserve(4 * n));
    int* node = Q.node.as<int>();
    hipLaunchKernelGGL(s3_fill, dim3(G), dim3(256), 0, s, node, N, 0);
    std::vector<S3Node> T(1);
    T[0] = {-1, (long)n, {-1, -1, -1, -1}, true};           // the top-level call always splits (semantic3d_dataset_sampling.py:242)
    std::vector<int> open = {0};
    for (int level = 0; !open.empty(); ++level) {
        if (level > 40 || (int)open.size() > S3_OPEN_MAX) { set_error("split3: more than %d parts above max_size on one level (or 40 levels: coincident points)", S3_OPEN_MAX); return SSDR_ERR_UNSUPPORTED; }
        const int nopen = (int)open.size(), nn = (int)T.size();
        // device tables of the level: slot_of [nn], child_base [nopen], box [nopen][4], thr [nopen][2], cnt [nopen][4]
        const size_t words = (size_t)nn + 11 * (size_t)nopen;
        SSDR_TRY(Q.tab.reserve(4 * words));
        int* slot_of = Q.tab.as<int>(); int* child_base = slot_of + nn; unsigned* box = reinterpret_cast<unsigned*>(child_base + nopen);
        float* thr = reinterpret_cast<float*>(box + 4 * nopen); int* cnt = reinterpret_cast<int*>(thr + 2 * nopen);
        std::vector<int> h(words, 0);
        for (int i = 0; i < nn; ++i) h[i] = -1;
        for (int k = 0; k < nopen; ++k) {
            h[open[k]] = k; h[nn + k] = (int)T.size();
            for (int q = 0; q < 4; ++q) { T[open[k]].child[q] = (int)T.size(); T.push_back({k, 0, {-1, -1, -1, -1}, false}); }
            for (int q = 0; q < 4; ++q) h[nn + nopen + 4 * k + q] = (q & 1) ? 0 : -1;      // box: min slots all ones, max slots zero
        }
        SSDR_HIP(hipMemcpyAsync(Q.tab.p, h.data(), 4 * words, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(s3_boxes, dim3(G), dim3(256), 0, s, d_xyz, node, N, slot_of, nopen, box);
        std::vector<unsigned> hb(4 * nopen);
        SSDR_HIP(hipMemcpyAsync(hb.data(), box, 16 * (size_t)nopen, hipMemcpyDeviceToHost, s));
        SSDR_HIP(hipStreamSynchronize(s));
        std::vector<float> ht(2 * nopen);
        for (int k = 0; k < nopen; ++k)
            for (int d = 0; d < 2; ++d) {
                const double lo = (double)o2f(hb[4 * k + 2 * d]), hi = (double)o2f(hb[4 * k + 2 * d + 1]);
                ht[2 * k + d] = (float)(lo + 0.5 * (hi - lo));        // float(np.min) + 0.5 * (float(np.max) - float(np.min)), then the float32 comparison
            }
        SSDR_HIP(hipMemcpyAsync(thr, ht.data(), 8 * (size_t)nopen, hipMemcpyHostToDevice, s));
        // (the children tables were sized before the pushes above: node ids of the new children are < T.size(), slot_of only covers the old ones)
        hipLaunchKernelGGL(s3_assign, dim3(G), dim3(256), 0, s, d_xyz, node, N, slot_of, nopen, thr, child_base, cnt);
        std::vector<int> hc(4 * nopen);
        SSDR_HIP(hipMemcpyAsync(hc.data(), cnt, 16 * (size_t)nopen, hipMemcpyDeviceToHost, s));
        SSDR_HIP(hipStreamSynchronize(s));
        std::vector<int> next;
        for (int k = 0; k < nopen; ++k)
            for (int q = 0; q < 4; ++q) {
                S3Node& c = T[T[open[k]].child[q]];
                c.count = hc[4 * k + q];
                if ((size_t)c.count > (level == 0 ? max_size : recurse_max_size)) {          // (:233: the recursive call passes its own literal)
                    if (c.count == T[open[k]].count) { set_error("split3: a part of %ld points above max_size does not split (coincident points): the reference recurses without end", c.count); return SSDR_ERR_UNSUPPORTED; }
                    c.split = true; next.push_back(T[open[k]].child[q]);
                }
            }
        open.swap(next);
    }
    // leaves in the order split3 appends them (depth first, children (x1,y1), (x1,y2), (x2,y1), (x2,y2)), then tf_map's merge rule (:243-251).  Every
    // child is followed by its z twin, which holds no points: an empty part only ever merges into the one before it — or, first in the list, is
    // joined by the parts after it and dropped if it stays empty (:255) — so empty parts change nothing and are skipped here.
    std::vector<int> rank_of(T.size(), 0), part_of(T.size(), -1);
    std::vector<long> part_cnt;
    int rank = 0;
    std::vector<std::pair<int, int>> st;            // (node, next child)
    st.push_back({0, 0});
    while (!st.empty()) {
        if (st.back().second == 4) { st.pop_back(); continue; }
        const int c = T[st.back().first].child[st.back().second++];
        if (T[c].split) { st.push_back({c, 0}); continue; }
        rank_of[c] = rank++;
        if (T[c].count == 0) continue;
        if (T[c].count > (long)merge_max || part_cnt.empty()) part_cnt.push_back(T[c].count);
        else part_cnt.back() += T[c].count;
        part_of[c] = (int)part_cnt.size() - 1;
    }
    size_t np_ = 0;
    np_ = part_cnt.size();
    if (np_ > max_parts) { set_error("split3: %zu parts, the caller has room for %zu", np_, max_parts); return SSDR_ERR_INVALID; }
    part_offsets[0] = 0;
    for (size_t p = 0; p < np_; ++p) part_offsets[p + 1] = part_offsets[p] + part_cnt[p];
    *num_parts = np_;
    // group the points: stable sort of the leaves' ranks (ascending index inside a leaf, leaves in append order)
    const size_t nt = T.size();
    SSDR_TRY(Q.tab.reserve(8 * nt));
    int* d_rank = Q.tab.as<int>(); int* d_partof = d_rank + nt;
    SSDR_HIP(hipMemcpyAsync(d_rank, rank_of.data(), 4 * nt, hipMemcpyHostToDevice, s));
    SSDR_HIP(hipMemcpyAsync(d_partof, part_of.data(), 4 * nt, hipMemcpyHostToDevice, s));
    SSDR_TRY(Q.keys.reserve(8 * n)); SSDR_TRY(Q.vals.reserve(4 * n));
    hipLaunchKernelGGL(s3_keys, dim3(G), dim3(256), 0, s, node, N, d_rank, d_partof, Q.keys.as<uint64_t>(), Q.vals.as<uint32_t>(), d_part);
    int bits = 1; while ((1 << bits) < std::max(rank, 2)) ++bits;
    SSDR_TRY(Q.sorter.sort(Q.keys.as<uint64_t>(), Q.vals.as<uint32_t>(), N, nullptr, s, bits));
    SSDR_HIP(hipMemcpyAsync(d_order, Q.vals.p, 4 * n, hipMemcpyDeviceToDevice, s));
    SSDR_HIP(hipStreamSynchronize(s));           // (rank_of / part_of are host vectors the copies above read)
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}
