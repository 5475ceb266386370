// Tile generator for gfx950: the step between the sub-sampled cloud and the KNN pyramid.
//
// Reference: spatially_regular_gen (S3/s3dis_dataset.py:115-154): query the num_points nearest points of a picked
// centre (sklearn KDTree.query(pick_point, k=num_points)), shuffle them, subtract the centre, and pad a cloud that
// is smaller than num_points by duplicating randomly chosen points (DP.data_aug, S3/helper_tool.py:185-199).
// Here: squared distance keys -> stable radix sort (ascending distance, ties by index) -> gather through a
// caller-supplied shuffle permutation.  Randomness (shuffle, duplicate choice) comes from the caller as arrays,
// as the reference draws it from np.random on the host.
#include "ssdr_internal.hpp"
#include "block_prims.hpp"
#include <map>
#include <vector>

namespace ssdr {
namespace {

__device__ __forceinline__ void tile_keys_body(const float* __restrict__ pts, const long long* __restrict__ d_m, int n_host, float cx, float cy, float cz,
                                                 uint64_t* keys, uint32_t* vals, int* d_count) {
    const int m = (int)min((long long)n_host, *d_m);
    if (blockIdx.x == 0 && threadIdx.x == 0) *d_count = m;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
        const float dx = pts[3 * (size_t)i] - cx, dy = pts[3 * (size_t)i + 1] - cy, dz = pts[3 * (size_t)i + 2] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        // non-negative floats order like their bit patterns.  Without a value array the index rides in the low word of the sort word
        // (distance bits << 32 | index): the sorter then moves 8 bytes per point and the gather reads the low words
        if (vals) { keys[i] = (uint64_t)__float_as_uint(d); vals[i] = (uint32_t)i; }
        else keys[i] = ((uint64_t)__float_as_uint(d) << 32) | (uint64_t)(uint32_t)i;
    }
}

// A cloud smaller than num_points (avail = m < num_points): the entries of perm below avail, in order, shuffle its points; `padmap` holds them compacted
// (padmap[k] = the k-th such entry) so that a row finds its source with one load.  One workgroup of 256 threads per cloud: every thread counts its
// stretch of perm, an exclusive scan of the 256 counts, every thread writes its stretch.  (Until round 6 every ROW scanned perm for its entry: num_points^2
// steps — 13.6 ms of a 22 ms step in the Semantic3D flavour with 65 536-point tiles over rooms of ~50 k points, tools/sem3d_probe.py; S3DIS rooms below
// 40 960 subsampled points take the same path.)
__device__ __forceinline__ void tile_padmap_body(int m, int num_points, const int* __restrict__ perm, int* __restrict__ padmap, unsigned* s_part) {
    const int avail = min(m, num_points), tid = threadIdx.x;
    if (avail == num_points) return;                       // (uniform)
    const int per = (num_points + 255) / 256, q0 = min(tid * per, num_points), q1 = min(q0 + per, num_points);
    unsigned cnt = 0;
    for (int q = q0; q < q1; ++q) cnt += perm[q] < avail ? 1u : 0u;
    __syncthreads();
    s_part[tid] = cnt;
    __syncthreads();
    unsigned incl = cnt;
    for (int o = 1; o < 256; o <<= 1) {
        const unsigned y = tid >= o ? s_part[tid - o] : 0u;
        __syncthreads();
        incl += y; s_part[tid] = incl;
        __syncthreads();
    }
    unsigned at = incl - cnt;
    for (int q = q0; q < q1; ++q) { const int v = perm[q]; if (v < avail) padmap[at++] = v; }
}

// out row r takes sorted position perm[r] when that position exists (< min(m, num_points)); a cloud smaller than
// num_points is padded: rows >= m duplicate point floor(dup_u[r] * m) of the *shuffled* list (data_aug).
__device__ __forceinline__ void tile_gather_body(const float* __restrict__ pts, const float* __restrict__ colors, int cdim,
                                                   const uint32_t* __restrict__ sorted, const int* __restrict__ d_count,
                                                   const int* __restrict__ perm, const float* __restrict__ dup_u, int num_points,
                                                   float cx, float cy, float cz, float color_scale,
                                                   float* out_xyz, float* out_feat, int* out_idx, const int* __restrict__ padmap, int stride = 1, int bx = -1,
                                                   const int* __restrict__ labels = nullptr, int* out_lab = nullptr) {
    const int m = *d_count;
    const int avail = min(m, num_points);
    for (int r = (bx < 0 ? (int)blockIdx.x : bx) * 256 + threadIdx.x; r < num_points; r += gridDim.x * 256) {
        int pos;
        if (avail == num_points) pos = perm[r];
        else {
            // small cloud: perm is a permutation of [0,num_points); its entries < avail, in order, shuffle the avail points (padmap, above):
            // row r < avail takes the r-th of them; row r >= avail duplicates one
            int want = r < avail ? r : (int)(dup_u[r] * (float)avail);
            if (want >= avail) want = avail - 1;
            pos = want >= 0 ? padmap[want] : 0;
        }
        const uint32_t id = sorted[(size_t)pos * stride];        // stride 2: the low words of 64-bit sort words
        const float x = pts[3 * (size_t)id] - cx, y = pts[3 * (size_t)id + 1] - cy, z = pts[3 * (size_t)id + 2] - cz;
        out_xyz[3 * (size_t)r] = x; out_xyz[3 * (size_t)r + 1] = y; out_xyz[3 * (size_t)r + 2] = z;
        if (out_feat) {
            float* f = out_feat + (size_t)r * (3 + cdim);
            f[0] = x; f[1] = y; f[2] = z;
            for (int c = 0; c < cdim; ++c) f[3 + c] = colors[(size_t)id * cdim + c] * color_scale;
        }
        if (out_idx) out_idx[r] = (int)id;
        if (out_lab) out_lab[r] = labels[id];              // queried_pc_label = input_label[queried_idx] (s3dis_dataset.py:141)
    }
}


// kernel entry points: one cloud, or all clouds of a batch (blockIdx.y = cloud)
__global__ __launch_bounds__(256) void tile_keys(const float* __restrict__ pts, const long long* __restrict__ d_m, int n_host, float cx, float cy, float cz, uint64_t* keys, uint32_t* vals, int* d_count) { tile_keys_body(pts, d_m, n_host, cx, cy, cz, keys, vals, d_count); }
__global__ __launch_bounds__(256) void tile_padmap(const int* __restrict__ d_count, int num_points, const int* __restrict__ perm, int* padmap) { __shared__ unsigned s_part[256]; tile_padmap_body(*d_count, num_points, perm, padmap, s_part); }
__global__ __launch_bounds__(256) void tile_gather(const float* __restrict__ pts, const float* __restrict__ colors, int cdim, const uint32_t* __restrict__ sorted, const int* __restrict__ d_count, const int* __restrict__ perm, const float* __restrict__ dup_u, int num_points, float cx, float cy, float cz, float color_scale, float* out_xyz, float* out_feat, int* out_idx, const int* __restrict__ padmap) { tile_gather_body(pts, colors, cdim, sorted, d_count, perm, dup_u, num_points, cx, cy, cz, color_scale, out_xyz, out_feat, out_idx, padmap); }

struct TileTab { int nr; int off[RADIX_MAX_SEG + 1]; int toff[RADIX_MAX_SEG + 1]; float cx[RADIX_MAX_SEG], cy[RADIX_MAX_SEG], cz[RADIX_MAX_SEG]; };

__global__ __launch_bounds__(256) void tile_keys_b(TileTab t, const float* __restrict__ pts, const long long* __restrict__ d_m, uint64_t* keys, uint32_t* vals, int* d_count) {
    const int r = blockIdx.y;
    tile_keys_body(pts + 3 * (size_t)t.off[r], d_m + r, t.off[r + 1] - t.off[r], t.cx[r], t.cy[r], t.cz[r], keys + t.toff[r], vals ? vals + t.toff[r] : nullptr, d_count + r);
}
__global__ __launch_bounds__(256) void tile_gather_b(TileTab t, const float* __restrict__ pts, const float* __restrict__ colors, int cdim, const uint32_t* __restrict__ sorted,
                                                     const int* __restrict__ d_count, const int* __restrict__ perm, const float* __restrict__ dup_u, int num_points,
                                                     float color_scale, float* out_xyz, float* out_feat, int* out_idx, int stride,
                                                     const int* __restrict__ labels, int* out_lab, const int* __restrict__ padmap) {
    int bx, r; xcd_tile_map(bx, r);            // the rows of a room are gathered at random: one room per XCD's L2
    const size_t o = (size_t)t.off[r], q = (size_t)r * num_points;
    tile_gather_body(pts + 3 * o, colors ? colors + o * cdim : nullptr, cdim, sorted + (size_t)t.toff[r] * stride, d_count + r, perm + q, dup_u + q, num_points, t.cx[r], t.cy[r], t.cz[r],
                     color_scale, out_xyz + 3 * q, out_feat ? out_feat + q * (3 + cdim) : nullptr, out_idx ? out_idx + q : nullptr, padmap + q, stride, bx,
                     labels ? labels + o : nullptr, out_lab ? out_lab + q : nullptr);
}

// ---- batch flavour: only the rows that can be among the num_points nearest are sorted ------------------------------------------
// A room holds ~4 x num_points rows; sorting all of them by distance (a segmented radix sort: a dozen launch-bound digit passes) was a third
// of the front end.  Non-negative float distances order like their bit patterns, so a histogram over the top TS_BITS bits of the pattern
// finds the first bin T whose cumulative count reaches num_points; rows with bin <= T (all num_points nearest and the rest of bin T, ties
// included) are the candidates.  They are written straight to their bin's slots (the histogram's prefix = cursors), and ranges of about a
// thousand consecutive candidates, cut at bin boundaries, are sorted in LDS.
constexpr int TS_BITS = 14, TS_BINS = 1 << TS_BITS, TS_SHIFT = 31 - TS_BITS;       // bit 31 (sign) is clear: 8 exponent + 6 mantissa bits

__device__ __forceinline__ float tile_dist(const float* __restrict__ pts, int i, float cx, float cy, float cz) {
    const float dx = pts[3 * (size_t)i] - cx, dy = pts[3 * (size_t)i + 1] - cy, dz = pts[3 * (size_t)i + 2] - cz;
    // the sign bit cleared: a sum of squares has none, but a NaN coordinate can carry one, and the bins below are indexed by the bit pattern
    // (0xFFC00000 >> 17 lies past the histogram; a positive NaN lands in the last bins like any far point)
    return __uint_as_float(__float_as_uint((dx * dx + dy * dy) + dz * dz) & 0x7fffffffu);
}
__global__ __launch_bounds__(256) void tile_hist_b(TileTab t, const float* __restrict__ pts, const long long* __restrict__ d_m, unsigned* hist, int* d_count) {
    __shared__ unsigned s_h[TS_BINS];
    const int r = blockIdx.y, n_host = t.off[r + 1] - t.off[r];
    const int m = (int)min((long long)n_host, d_m[r]);
    if (blockIdx.x == 0 && threadIdx.x == 0) d_count[r] = m;
    if ((int)blockIdx.x * 256 >= m) return;
    for (int b = threadIdx.x; b < TS_BINS; b += 256) s_h[b] = 0u;
    __syncthreads();
    const float* P = pts + 3 * (size_t)t.off[r];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256)
        atomicAdd(&s_h[__float_as_uint(tile_dist(P, i, t.cx[r], t.cy[r], t.cz[r])) >> TS_SHIFT], 1u);
    __syncthreads();
    for (int b = threadIdx.x; b < TS_BINS; b += 256) if (s_h[b]) atomicAdd(&hist[(size_t)r * TS_BINS + b], s_h[b]);
}
// one workgroup per room: first bin T whose cumulative count reaches num_points (the last non-empty bin when the room is smaller); the start
// of every bin <= T among the room's candidates (the histogram becomes the bins' write cursors; bins beyond T are cleared); the ranges the
// sort works on: range k = the bins that START in [k TS_RSTEP, (k + 1) TS_RSTEP)
constexpr int TS_RCAP = 4096, TS_RSTEP = 1024;
__global__ __launch_bounds__(256) void tile_thresh_b(TileTab t, unsigned* hist, const int* __restrict__ d_count, int num_points, unsigned* thr, int* d_cand, unsigned* rstart, unsigned* rcur, int rstride,
                                                     const int* __restrict__ perm, int* padmap) {
    __shared__ unsigned s_h[TS_BINS + TS_BINS / 32];          // the room's histogram (coalesced in, coalesced out); one pad word per 32 bins: a thread's stretch starts in its own bank pair
    __shared__ unsigned s_part[256];
    __shared__ unsigned s_T;
    const int r = blockIdx.x, tid = threadIdx.x;
    unsigned* h = hist + (size_t)r * TS_BINS;
    unsigned* RS = rstart + (size_t)r * rstride;
    constexpr int PER = TS_BINS / 256;
    auto at = [](int b) { return b + (b >> 5); };
    const unsigned want = (unsigned)min(num_points, d_count[r]);
    for (int b = tid; b < TS_BINS; b += 256) s_h[at(b)] = h[b];
    for (int k = tid; k < rstride; k += 256) { RS[k] = 0xffffffffu; rcur[(size_t)r * rstride + k] = 0u; }
    __syncthreads();
    unsigned tot = 0;
    for (int k = 0; k < PER; ++k) tot += s_h[at(tid * PER + k)];
    s_part[tid] = tot;
    __syncthreads();
    // exclusive prefix of the per-thread totals (256 entries: Hillis-Steele in LDS)
    unsigned incl = tot;
    for (int o = 1; o < 256; o <<= 1) {
        const unsigned y = tid >= o ? s_part[tid - o] : 0u;
        __syncthreads();
        incl += y; s_part[tid] = incl;
        __syncthreads();
    }
    const unsigned before = incl - tot;
    if (want > 0 && before < want && incl >= want) {          // the bin where the cumulative count reaches `want` lies in this thread's stretch
        unsigned run = before; int k = 0;
        while (k < PER - 1 && run + s_h[at(tid * PER + k)] < want) { run += s_h[at(tid * PER + k)]; ++k; }
        s_T = (unsigned)(tid * PER + k);
    }
    if (want == 0 && tid == 0) s_T = 0;
    __syncthreads();
    const unsigned T = s_T;
    unsigned pos = before, ncand = 0;
    for (int k = 0; k < PER; ++k) {
        const int b = tid * PER + k;
        const unsigned cb = s_h[at(b)];
        if ((unsigned)b <= T) {
            s_h[at(b)] = pos;
            if (cb) atomicMin(&RS[pos / TS_RSTEP], pos);
            pos += cb; ncand = pos;
        } else s_h[at(b)] = 0u;
    }
    if ((unsigned)(tid * PER) <= T && (unsigned)(tid * PER + PER - 1) >= T) { thr[r] = T; d_cand[r] = (int)ncand; }
    __syncthreads();
    for (int b = tid; b < TS_BINS; b += 256) h[b] = s_h[at(b)];
    tile_padmap_body(d_count[r], num_points, perm + (size_t)r * num_points, padmap + (size_t)r * num_points, s_part);      // (a room smaller than a tile only)
}
// candidates (bin <= T) to their RANGE's slots (the sort orders a range completely, so the order inside it is free): a workgroup counts its
// candidates per range in LDS (the range of bin b is (start of b) / TS_RSTEP: the histogram now holds the starts), reserves each range's share
// with ONE global atomic and writes.  A global atomic per candidate (0.66 M per step on ~130 k cursors) was the cost of this kernel.
constexpr int TS_CPT = 12;            // rows per thread (held in registers between the two passes)
constexpr int TS_RMAX = 2048;         // ranges a workgroup keeps counters for; beyond: the per-bin cursors (tile_compact_bins_b)
__global__ __launch_bounds__(256) void tile_compact_b(TileTab t, const float* __restrict__ pts, const int* __restrict__ d_count, const unsigned* __restrict__ thr,
                                                      const unsigned* __restrict__ hist, const unsigned* __restrict__ rstart, unsigned* rcur, int rstride, uint64_t* keys) {
    __shared__ unsigned s_cnt[TS_RMAX], s_base[TS_RMAX];
    const int r = blockIdx.y, m = d_count[r], tid = threadIdx.x;
    const float* P = pts + 3 * (size_t)t.off[r];
    const unsigned T = thr[r];
    const unsigned* h = hist + (size_t)r * TS_BINS;
    const unsigned* RS = rstart + (size_t)r * rstride;
    unsigned* RC = rcur + (size_t)r * rstride;
    uint64_t* K = keys + t.toff[r];
    for (int i0 = blockIdx.x * 256 * TS_CPT; i0 < m; i0 += gridDim.x * 256 * TS_CPT) {          // uniform over the workgroup
        for (int k = tid; k < rstride; k += 256) s_cnt[k] = 0u;
        __syncthreads();
        unsigned bits[TS_CPT], loc[TS_CPT]; int rid[TS_CPT];
#pragma unroll
        for (int u = 0; u < TS_CPT; ++u) {
            const int i = i0 + u * 256 + tid;
            rid[u] = -1; bits[u] = 0u; loc[u] = 0u;
            if (i < m) {
                bits[u] = __float_as_uint(tile_dist(P, i, t.cx[r], t.cy[r], t.cz[r]));
                const unsigned b = bits[u] >> TS_SHIFT;
                if (b <= T) { rid[u] = (int)(h[b] / TS_RSTEP); loc[u] = atomicAdd(&s_cnt[rid[u]], 1u); }
            }
        }
        __syncthreads();
        for (int k = tid; k < rstride; k += 256) { const unsigned c = s_cnt[k]; if (c) s_base[k] = RS[k] + atomicAdd(&RC[k], c); }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TS_CPT; ++u)
            if (rid[u] >= 0) K[s_base[rid[u]] + loc[u]] = ((uint64_t)bits[u] << 32) | (uint64_t)(uint32_t)(i0 + u * 256 + tid);
        __syncthreads();
    }
}
// the same with one global atomic per candidate on its bin's cursor (rooms of more than TS_RMAX ranges)
__global__ __launch_bounds__(256) void tile_compact_bins_b(TileTab t, const float* __restrict__ pts, const int* __restrict__ d_count, const unsigned* __restrict__ thr, unsigned* hist, uint64_t* keys) {
    const int r = blockIdx.y, m = d_count[r];
    const float* P = pts + 3 * (size_t)t.off[r];
    const unsigned T = thr[r];
    unsigned* cur = hist + (size_t)r * TS_BINS;
    uint64_t* K = keys + t.toff[r];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
        const unsigned bits = __float_as_uint(tile_dist(P, i, t.cx[r], t.cy[r], t.cz[r]));
        const unsigned b = bits >> TS_SHIFT;
        if (b <= T) K[atomicAdd(&cur[b], 1u)] = ((uint64_t)bits << 32) | (uint64_t)(uint32_t)i;      // (one atomic per bin and wave for rows that share a bin: measured slower, the rows of a wave rarely do)
    }
}
// clears what tile_compact_b left in the cursors (bins <= T), for the next call's histogram
__global__ __launch_bounds__(256) void tile_clear_b(unsigned* hist) {
    for (int b = blockIdx.x * 256 + threadIdx.x; b < TS_BINS; b += gridDim.x * 256) hist[(size_t)blockIdx.y * TS_BINS + b] = 0u;
}
// one workgroup per range: bitonic sort of its words (distance bits << 32 | index: ascending distance, ties by index).  A range of at most
// TS_RCAP words is sorted in LDS; a larger one (a single bin of more than TS_RSTEP candidates: rows crowded into 1/64 of a binade of
// distance) in place in global memory.
__global__ __launch_bounds__(256) void tile_binsort_b(TileTab t, const unsigned* __restrict__ rstart, int rstride, const int* __restrict__ d_cand, uint64_t* keys) {
    __shared__ uint64_t s_k[TS_RCAP];
    const int r = blockIdx.y, tid = threadIdx.x;
    const unsigned* RS = rstart + (size_t)r * rstride;
    uint64_t* K = keys + t.toff[r];
    const unsigned cand = (unsigned)d_cand[r];
    const int nk = (int)((cand + TS_RSTEP - 1) / TS_RSTEP);
    for (int q = blockIdx.x; q < nk; q += gridDim.x) {
        const unsigned s0 = RS[q];
        if (s0 == 0xffffffffu) continue;                         // no bin starts here (it lies inside the previous range's last bin)
        unsigned e0 = cand;
        for (int q2 = q + 1; q2 < nk; ++q2) if (RS[q2] != 0xffffffffu) { e0 = RS[q2]; break; }
        const unsigned n = e0 - s0;
        if (n <= 1) continue;
        unsigned N = 2; while (N < n) N <<= 1;
        // bitonic network with ascending comparators only (the first step of every merge pairs i with its mirror image in the block): positions
        // >= n then simply count as +infinity and are never touched, so n need not be a power of two and nothing is padded
        const bool lds = n <= (unsigned)TS_RCAP;
        uint64_t* A = lds ? s_k : K + s0;
        if (N <= 2048u && N >= 16u) {
            // the usual case (~1000-1450 words), eight consecutive words per thread: the steps with partners less than eight apart run in registers
            // (no index arithmetic, no LDS, no barrier: 30 of the 66 steps of a 2048-word network), the others through LDS four comparators per thread;
            // the words beyond n are +infinity here (a plain network on N words)
            for (unsigned i = tid; i < N; i += 256) s_k[i] = i < n ? K[s0 + i] : ~0ull;
            __syncthreads();
            const bool act = (unsigned)tid < N / 8;
            uint64_t v[8];
            auto ce = [&](int a, int b) { const uint64_t x = v[a], y = v[b]; const bool sw = x > y; v[a] = sw ? y : x; v[b] = sw ? x : y; };
            auto ld8 = [&]() {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = s_k[8 * tid + e];
            };
            auto st8 = [&]() {
#pragma unroll
                for (int e = 0; e < 8; ++e) s_k[8 * tid + e] = v[e];
            };
            auto tail3 = [&]() { ce(0, 4); ce(1, 5); ce(2, 6); ce(3, 7); ce(0, 2); ce(1, 3); ce(4, 6); ce(5, 7); ce(0, 1); ce(2, 3); ce(4, 5); ce(6, 7); };
            if (act) {
                ld8();
                ce(0, 1); ce(2, 3); ce(4, 5); ce(6, 7);                                        // k = 2
                ce(0, 3); ce(1, 2); ce(4, 7); ce(5, 6); ce(0, 1); ce(2, 3); ce(4, 5); ce(6, 7);        // k = 4: mirror step, then 1
                ce(0, 7); ce(1, 6); ce(2, 5); ce(3, 4); ce(0, 2); ce(1, 3); ce(4, 6); ce(5, 7); ce(0, 1); ce(2, 3); ce(4, 5); ce(6, 7);      // k = 8
                st8();
            }
            __syncthreads();
            for (unsigned k = 16; k <= N; k <<= 1) {
                for (unsigned j = k >> 1; j >= 8; j >>= 1) {
                    const int lj = __ffsll((long long)j) - 1;
                    if (act) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const unsigned i = 4u * tid + c;
                            unsigned lo, hi;
                            if (j == (k >> 1)) { const unsigned blk = i >> lj, off = i & (j - 1); lo = blk * k + off; hi = blk * k + (k - 1 - off); }
                            else { lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)); hi = lo | j; }
                            const uint64_t a = s_k[lo], b = s_k[hi];
                            if (a > b) { s_k[lo] = b; s_k[hi] = a; }
                        }
                    }
                    __syncthreads();
                }
                if (act) { ld8(); tail3(); st8(); }
                __syncthreads();
            }
            for (unsigned i = tid; i < n; i += 256) K[s0 + i] = s_k[i];
            __syncthreads();
            continue;
        }
        if (lds) { for (unsigned i = tid; i < n; i += 256) s_k[i] = K[s0 + i]; __syncthreads(); }
        for (unsigned k = 2; k <= N; k <<= 1) {
            for (unsigned j = k >> 1; j > 0; j >>= 1) {
                for (unsigned i = tid; i < N / 2; i += 256) {
                    unsigned lo, hi;
                    if (j == (k >> 1)) { const unsigned blk = i / j, off = i % j; lo = blk * k + off; hi = blk * k + (k - 1 - off); }
                    else { lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)); hi = lo | j; }
                    if (hi < n) { const uint64_t a = A[lo], b = A[hi]; if (a > b) { A[lo] = b; A[hi] = a; } }
                }
                if (!lds) __threadfence_block();
                __syncthreads();
            }
        }
        if (lds) { for (unsigned i = tid; i < n; i += 256) K[s0 + i] = s_k[i]; __syncthreads(); }
    }
}

// Possibility map of the test-time generator (S3/s3dis_dataset_test.py:140-143): for the `avail` points of the tile,
// dists = (dx*dx + dy*dy) + dz*dz in float32, delta = (1 - dists / max(dists))^2, possibility[idx] += delta (float64).
// The tile's points are the first `avail` entries of the distance-sorted list, so max(dists) is the key of the last one.
__global__ __launch_bounds__(256) void tile_possibility(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ sorted, const int* __restrict__ d_count,
                                                        int num_points, double* possibility) {
    const int avail = min(*d_count, num_points);
    if (avail <= 0) return;
    const float dmax = __uint_as_float((unsigned)keys[avail - 1]);
    for (int r = blockIdx.x * 256 + threadIdx.x; r < avail; r += gridDim.x * 256) {
        const float d = __uint_as_float((unsigned)keys[r]);
        const float q = 1 - d / dmax;
        possibility[sorted[r]] += (double)(q * q);
    }
}

// min / argmin of the possibility map (np.min, np.argmin: first minimum), one workgroup
__global__ __launch_bounds__(1024) void possibility_min(const double* __restrict__ possibility, const long long* __restrict__ d_m, int n_host, double* out_min, int* out_arg) {
    __shared__ double s_v[1024];
    __shared__ int s_i[1024];
    const int m = (int)min((long long)n_host, *d_m), tid = threadIdx.x;
    double bv = 1.0e300; int bi = 0x7fffffff;
    for (int i = tid; i < m; i += 1024) { const double v = possibility[i]; if (v < bv || (v == bv && i < bi)) { bv = v; bi = i; } }
    s_v[tid] = bv; s_i[tid] = bi;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o && (s_v[tid + o] < s_v[tid] || (s_v[tid + o] == s_v[tid] && s_i[tid + o] < s_i[tid]))) { s_v[tid] = s_v[tid + o]; s_i[tid] = s_i[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { *out_min = s_v[0]; *out_arg = s_i[0]; }
}

struct TileState { RadixSorter sorter; DevBuf keys, vals, count, hist, thr, cand, rstart, rcur, padmap; bool hist_clear = false; };
TileState& tst(hipStream_t st) { return per_stream<TileState>(st); }

}  // namespace
}  // namespace ssdr

using namespace ssdr;

extern "C" int ssdr_tile_select_possibility_dev(const float*, const float*, int, const int64_t*, size_t, const float*, size_t, const int32_t*, const float*, float,
                                                float*, float*, int32_t*, double*, double*, int32_t*, void*);

extern "C" int ssdr_tile_select_dev(const float* d_points, const float* d_colors, int color_dim, const int64_t* d_m, size_t n_max,
                                    const float* center, size_t num_points, const int32_t* d_perm, const float* d_dup_u, float color_scale,
                                    float* d_out_xyz, float* d_out_feat, int32_t* d_out_idx, void* stream) {
    return ssdr_tile_select_possibility_dev(d_points, d_colors, color_dim, d_m, n_max, center, num_points, d_perm, d_dup_u, color_scale, d_out_xyz, d_out_feat,
                                            d_out_idx, nullptr, nullptr, nullptr, stream);
}

extern "C" int ssdr_tile_select_possibility_dev(const float* d_points, const float* d_colors, int color_dim, const int64_t* d_m, size_t n_max,
                                                const float* center, size_t num_points, const int32_t* d_perm, const float* d_dup_u, float color_scale,
                                                float* d_out_xyz, float* d_out_feat, int32_t* d_out_idx,
                                                double* d_possibility, double* d_out_min_possibility, int32_t* d_out_argmin, void* stream) {
    if (!d_points || !d_m || !center || !d_perm || !d_dup_u || !d_out_xyz || n_max == 0 || num_points == 0 || n_max > 0x3fffffff) { set_error("tile_select: bad arguments"); return SSDR_ERR_INVALID; }
    if (d_out_feat && color_dim > 0 && !d_colors) { set_error("tile_select: colors missing"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream); TileState& T = tst(s);
    SSDR_TRY(T.keys.reserve(8 * n_max)); SSDR_TRY(T.vals.reserve(4 * n_max)); SSDR_TRY(T.count.reserve(16));
    const int g = (int)std::max<size_t>(1, std::min<size_t>((n_max + 255) / 256, (size_t)ctx().num_cu * 8));
    hipLaunchKernelGGL(tile_keys, dim3(g), dim3(256), 0, s, d_points, (const long long*)d_m, (int)n_max, center[0], center[1], center[2],
                       T.keys.as<uint64_t>(), T.vals.as<uint32_t>(), T.count.as<int>());
    SSDR_TRY(T.sorter.sort(T.keys.as<uint64_t>(), T.vals.as<uint32_t>(), (int)n_max, T.count.as<int>(), s, 32));   // float bit patterns
    const int g2 = (int)std::max<size_t>(1, std::min<size_t>((num_points + 255) / 256, 1024));
    SSDR_TRY(T.padmap.reserve(4 * num_points));
    hipLaunchKernelGGL(tile_padmap, dim3(1), dim3(256), 0, s, T.count.as<int>(), (int)num_points, d_perm, T.padmap.as<int>());
    hipLaunchKernelGGL(tile_gather, dim3(g2), dim3(256), 0, s, d_points, d_colors, d_colors ? color_dim : 0, T.vals.as<uint32_t>(), T.count.as<int>(),
                       d_perm, d_dup_u, (int)num_points, center[0], center[1], center[2], color_scale, d_out_xyz, d_out_feat, d_out_idx, T.padmap.as<int>());
    if (d_possibility) {
        hipLaunchKernelGGL(tile_possibility, dim3(g2), dim3(256), 0, s, T.keys.as<uint64_t>(), T.vals.as<uint32_t>(), T.count.as<int>(), (int)num_points, d_possibility);
        if (d_out_min_possibility && d_out_argmin)
            hipLaunchKernelGGL(possibility_min, dim3(1), dim3(1024), 0, s, d_possibility, (const long long*)d_m, (int)n_max, d_out_min_possibility, d_out_argmin);
    }
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

extern "C" int ssdr_tile_select_batch_dev(const float* d_points, const float* d_colors, int color_dim, const int64_t* d_m, const int64_t* cloud_offsets, size_t num_clouds,
                                          const float* centers, size_t num_points, const int32_t* d_perm, const float* d_dup_u, float color_scale,
                                          float* d_out_xyz, float* d_out_feat, int32_t* d_out_idx, const int32_t* d_labels, int32_t* d_out_labels, void* stream) {
    if (d_out_labels && !d_labels) { set_error("tile_select_batch: labels missing"); return SSDR_ERR_INVALID; }
    if (!d_points || !d_m || !cloud_offsets || !centers || !d_perm || !d_dup_u || !d_out_xyz || num_clouds == 0 || num_clouds > RADIX_MAX_SEG || num_points == 0) { set_error("tile_select_batch: bad arguments (1..%d clouds)", RADIX_MAX_SEG); return SSDR_ERR_INVALID; }
    if (d_out_feat && color_dim > 0 && !d_colors) { set_error("tile_select_batch: colors missing"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream); TileState& T = tst(s);
    TileTab t; t.nr = (int)num_clouds;
    int toff = 0, maxn = 0; std::vector<int> n_host(num_clouds);
    for (size_t r = 0; r < num_clouds; ++r) {
        const long n = (long)(cloud_offsets[r + 1] - cloud_offsets[r]);
        if (n <= 0 || cloud_offsets[r + 1] > 0x3fffffff) { set_error("tile_select_batch: bad cloud offsets"); return SSDR_ERR_INVALID; }
        t.off[r] = (int)cloud_offsets[r]; t.toff[r] = toff; n_host[r] = (int)n; maxn = std::max(maxn, (int)n);
        t.cx[r] = centers[3 * r]; t.cy[r] = centers[3 * r + 1]; t.cz[r] = centers[3 * r + 2];
        toff += ((int)n + RADIX_TILE - 1) / RADIX_TILE * RADIX_TILE;
    }
    t.off[num_clouds] = (int)cloud_offsets[num_clouds]; t.toff[num_clouds] = toff;
    SSDR_TRY(T.keys.reserve(8 * (size_t)toff + 16)); SSDR_TRY(T.count.reserve(4 * num_clouds + 16));
    const int rstride = (maxn + TS_RSTEP - 1) / TS_RSTEP + 1;
    SSDR_TRY(T.hist.reserve(4 * (size_t)TS_BINS * RADIX_MAX_SEG)); SSDR_TRY(T.thr.reserve(4 * RADIX_MAX_SEG)); SSDR_TRY(T.cand.reserve(4 * RADIX_MAX_SEG));
    SSDR_TRY(T.rstart.reserve(4 * (size_t)rstride * num_clouds)); SSDR_TRY(T.rcur.reserve(4 * (size_t)rstride * num_clouds));
    SSDR_TRY(T.padmap.reserve(4 * num_clouds * num_points));
    if (!T.hist_clear) { SSDR_HIP(hipMemsetAsync(T.hist.p, 0, 4 * (size_t)TS_BINS * RADIX_MAX_SEG, s)); T.hist_clear = true; }      // tile_clear_b leaves it clear
    const unsigned R = (unsigned)num_clouds;
    const int g = std::max(1, std::min((maxn + 255) / 256, 64));
    ProfScope prof("tile_select", s, (12.0 + 4.0 * (12 + 4 * (d_out_feat ? color_dim + 3 : 0))) * 0.0 + 40.0 * (double)num_clouds * (double)num_points);      // 40 B per tile point
    hipLaunchKernelGGL(tile_hist_b, dim3(g, R), dim3(256), 0, s, t, d_points, (const long long*)d_m, T.hist.as<unsigned>(), T.count.as<int>());
    hipLaunchKernelGGL(tile_thresh_b, dim3(R), dim3(256), 0, s, t, T.hist.as<unsigned>(), T.count.as<int>(), (int)num_points, T.thr.as<unsigned>(), T.cand.as<int>(), T.rstart.as<unsigned>(), T.rcur.as<unsigned>(), rstride,
                       d_perm, T.padmap.as<int>());
    if (rstride <= TS_RMAX)
        hipLaunchKernelGGL(tile_compact_b, dim3(std::max(1, std::min((maxn + 256 * TS_CPT - 1) / (256 * TS_CPT), 256)), R), dim3(256), 0, s, t, d_points, T.count.as<int>(), T.thr.as<unsigned>(),
                           T.hist.as<unsigned>(), T.rstart.as<unsigned>(), T.rcur.as<unsigned>(), rstride, T.keys.as<uint64_t>());
    else hipLaunchKernelGGL(tile_compact_bins_b, dim3(g, R), dim3(256), 0, s, t, d_points, T.count.as<int>(), T.thr.as<unsigned>(), T.hist.as<unsigned>(), T.keys.as<uint64_t>());
    hipLaunchKernelGGL(tile_clear_b, dim3(8, R), dim3(256), 0, s, T.hist.as<unsigned>());
    hipLaunchKernelGGL(tile_binsort_b, dim3(std::min(rstride, 64), R), dim3(256), 0, s, t, T.rstart.as<unsigned>(), rstride, T.cand.as<int>(), T.keys.as<uint64_t>());
    const int g2 = (int)std::max<size_t>(1, std::min<size_t>((num_points + 255) / 256, 256));
    hipLaunchKernelGGL(tile_gather_b, dim3(g2, R), dim3(256), 0, s, t, d_points, d_colors, d_colors ? color_dim : 0, reinterpret_cast<const uint32_t*>(T.keys.as<uint64_t>()), T.count.as<int>(),
                       d_perm, d_dup_u, (int)num_points, color_scale, d_out_xyz, d_out_feat, d_out_idx, 2, d_labels, d_out_labels, T.padmap.as<int>());
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}
