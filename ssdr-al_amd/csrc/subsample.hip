// Voxel-grid barycentre subsampling for gfx950.
//
// Reference: S3/utils/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106 — one
// thread walks the cloud and accumulates into an std::unordered_map keyed by the voxel index.  What makes
// the result reproducible bit for bit is (a) the fp32 key arithmetic (:27-31, :53-56), (b) that every
// voxel's sums are taken in *input order* (:59-70) and (c) the finalisation `sum * (float)(1.0/count)` for
// positions but `sum / (float)count` for features (:87-95).  The map itself is irrelevant, so here:
//
//   keys     one pass computes the size_t voxel key of every point (same fp32 ops, contraction off)
//   sort     stable radix sort of (key, point index): members of a voxel become contiguous and stay in
//            input order (radix_sort.hip)
//   heads    ballot + prefix-sum compaction of segment heads -> M voxels, seg_start[M+1]
//   reduce   one lane per voxel adds its members sequentially (the order the reference adds them in),
//            and replays the label histogram's first-maximum rule (:97-101), including the iteration
//            order of the reference's per-voxel unordered_map<int,int> (13 -> 29 buckets, most recently
//            first-seen label first)
//   order    SSDR_ORDER_KEY: rows by ascending key.  SSDR_ORDER_REFERENCE: rows permuted into the
//            iteration order of the reference's unordered_map<size_t,...> (subsample_order.hip).
#include "ssdr_internal.hpp"
#include <atomic>
#include "block_prims.hpp"
#include "voxel_label.hpp"
#include "subsample_types.hpp"
#include <map>
#include <cstring>

namespace ssdr {
namespace {

__device__ __forceinline__ void gs_params_body(const float* partial, int nparts, float dl, GsParams* prm) {
    __shared__ float s_mm[(BS / 64) * 6];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = threadIdx.x; i < nparts; i += BS) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], partial[6 * i + d]); mx[d] = fmaxf(mx[d], partial[6 * i + 3 + d]); }
    }
    block_minmax3(mn, mx, s_mm);
    if (threadIdx.x == 0) {
        // grid_subsampling.cpp:27-31
        const float inv = 1 / dl;
        float org[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { org[d] = floorf(mn[d] * inv) * dl; prm->org[d] = org[d]; }
        prm->dl = dl;
        prm->nx = (unsigned long long)(long long)floorf((mx[0] - org[0]) / dl) + 1ull;
        prm->ny = (unsigned long long)(long long)floorf((mx[1] - org[1]) / dl) + 1ull;
        prm->m = 0; prm->status = 0;
        // keys are ix + nx (iy + ny iz) with ix < nx, iy < ny, iz < nz (the same rounded quotients as the key kernel's): all below nx ny nz
        const unsigned long long nz = (unsigned long long)(long long)floorf((mx[2] - org[2]) / dl) + 1ull;
        unsigned long long top = prm->nx * prm->ny * nz - 1ull, m = 0;
        if (prm->nx == 0 || prm->ny == 0 || nz == 0 || (prm->nx * prm->ny) / prm->ny != prm->nx || (prm->nx * prm->ny * nz) / nz != prm->nx * prm->ny) m = ~0ull;      // wrapped: anything goes
        else while (m < top) m = (m << 1) | 1ull;
        prm->key_and = 0ull; prm->key_or = m;
    }
}

// With `rec` the kernel also packs every point's row (xyz, features, labels; at most REC_W words) into one 32-byte
// record: the per-voxel reduction then gathers ONE memory sector per point in sorted order instead of one from each of
// three arrays (PMC: the three-array reduction fetched 6-11x its algorithmic bytes, a random 4-12 byte read pays for a
// whole sector).
// idx_bits > 0: the sort word is (voxel key << idx_bits) | point index and no value array is written — the stable sort on the upper part keeps
// the indices of a voxel ascending, and the sorter moves 8 bytes per point instead of 12.  A voxel key that does not leave room for the
// index (a grid of more than 2^(64 - idx_bits) cells) sets status bit 1: that cloud needs the (key, value) flavour.
__device__ __forceinline__ void gs_keys_body(const float* __restrict__ P, int n, GsParams* prm, uint64_t* keys, uint32_t* vals,
                                              const float* __restrict__ F, int fdim, const int* __restrict__ cls, int ldim, uint32_t* rec, int idx_bits = 0) {
    const float ox = prm->org[0], oy = prm->org[1], oz = prm->org[2], dl = prm->dl;
    const unsigned long long nx = prm->nx, ny = prm->ny;
    for (int i = blockIdx.x * BS + threadIdx.x; i < n; i += gridDim.x * BS) {
        const float x = P[3 * (size_t)i], y = P[3 * (size_t)i + 1], z = P[3 * (size_t)i + 2];
        // grid_subsampling.cpp:53-56 (size_t arithmetic wraps)
        const unsigned long long ix = (unsigned long long)(long long)floorf((x - ox) / dl);
        const unsigned long long iy = (unsigned long long)(long long)floorf((y - oy) / dl);
        const unsigned long long iz = (unsigned long long)(long long)floorf((z - oz) / dl);
        const unsigned long long key = ix + nx * iy + nx * ny * iz;
        if (idx_bits > 0) {
            if (key >> (64 - idx_bits)) atomicOr(&prm->status, 2);
            keys[i] = (key << idx_bits) | (unsigned long long)i;
        } else { keys[i] = key; vals[i] = (uint32_t)i; }
        if (rec) {
            uint32_t w[REC_W];
            w[0] = __float_as_uint(x); w[1] = __float_as_uint(y); w[2] = __float_as_uint(z);
#pragma unroll
            for (int k = 3; k < REC_W; ++k) {
                uint32_t v = 0;
                if (k - 3 < fdim) v = __float_as_uint(F[(size_t)i * fdim + (k - 3)]);
                else if (k - 3 - fdim < ldim) v = (uint32_t)cls[(size_t)i * ldim + (k - 3 - fdim)];
                w[k] = v;
            }
            uint4* r4 = reinterpret_cast<uint4*>(rec + (size_t)i * REC_W);
            r4[0] = make_uint4(w[0], w[1], w[2], w[3]); r4[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
}

// ---- segment heads: 3-step compaction -------------------------------------------------------------
// shift > 0: composite words (see gs_keys_body) — the voxel key is the part above `shift`; the kernel then also unpacks the point indices
// into `vals_out` for the reduction, which is written against a value array
__device__ __forceinline__ void gs_heads_count_body(const uint64_t* __restrict__ ks, int n, int* bsum, int shift = 0, uint32_t* vals_out = nullptr) {
    __shared__ int s_sum[(BS / 64) * 2];
    const int base = blockIdx.x * CHUNK;
    int c = 0, z = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int i = base + u * BS + threadIdx.x;
        if (i < n) {
            const uint64_t k = ks[i];
            c += (i == 0) || ((k >> shift) != (ks[i - 1] >> shift));
            if (vals_out) vals_out[i] = (uint32_t)(k & ((1ull << shift) - 1ull));
        }
    }
    block_sum2(c, z, s_sum);
    if (threadIdx.x == 0) bsum[blockIdx.x] = c;
}

__device__ __forceinline__ void gs_heads_scan_body(int* bsum, int nb, GsParams* prm, int* seg_start, int n) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int e = base + tid;
        const int x = e < nb ? bsum[e] : 0;
        int incl = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { int y = __shfl_up(incl, off); if (lane >= off) incl += y; }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { int c = wsum[w]; if (w < wid) wbase += c; tot += c; }
        const int carry = carry_s;
        if (e < nb) bsum[e] = carry + wbase + incl - x;
        __syncthreads();
        if (tid == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (tid == 0) { prm->m = carry_s; seg_start[carry_s] = n; }
}

__device__ __forceinline__ void gs_heads_write_body(const uint64_t* __restrict__ ks, int n, const int* bsum, int* seg_start, int shift = 0) {
    __shared__ int s_w[2][U][BS / 64];
    const int base = blockIdx.x * CHUNK, off = bsum[blockIdx.x];
    const int hi = min(n, base + CHUNK);
    block_compact(base, hi, [&](int i) { return (i == 0) || ((ks[i] >> shift) != (ks[i - 1] >> shift)); },
                  [&](int k, int i) { seg_start[off + k] = i; }, s_w);
}

// ---- per-voxel reduction ----------------------------------------------------------------------------
// majority label with the reference's tie rule: voxel_label.hpp (labels through a getter, in input order)
__device__ int voxel_label(const int* __restrict__ cls, int ldim, int col, const uint32_t* __restrict__ vs, int s, int e, int* status) {
    return voxel_label_t([&](int j) { return cls[(size_t)vs[j] * ldim + col]; }, s, e, status);
}
__device__ __forceinline__ int voxel_label_fast(const int* __restrict__ cls, int ldim, int col, const uint32_t* __restrict__ vs, int s, int e) {
    return voxel_label_fast_t([&](int j) { return cls[(size_t)vs[j] * ldim + col]; }, s, e);
}

__device__ __forceinline__ void gs_reduce_labels_body(const int* __restrict__ cls, int ldim, const uint32_t* __restrict__ vs, const int* __restrict__ seg_start,
                                                       GsParams* prm, const int* __restrict__ row_of_voxel, int* out_c) {
    const long long total = (long long)prm->m * ldim;
    for (long long e = (long long)blockIdx.x * BS + threadIdx.x; e < total; e += (long long)gridDim.x * BS) {
        const int v = (int)(e / ldim), col = (int)(e % ldim);
        const int s = seg_start[v], en = seg_start[v + 1];
        const int row = row_of_voxel ? row_of_voxel[v] : v;
        int lab = voxel_label_fast(cls, ldim, col, vs, s, en);
        if (lab < 0) lab = voxel_label(cls, ldim, col, vs, s, en, &prm->status);
        out_c[(size_t)row * ldim + col] = lab;
    }
}

__device__ __forceinline__ void gs_reduce_body(const float* __restrict__ P, const float* __restrict__ F, int fdim,
                                                const int* __restrict__ cls, int ldim,
                                                const uint32_t* __restrict__ vs, const int* __restrict__ seg_start, GsParams* prm,
                                                const int* __restrict__ row_of_voxel,
                                                float* out_p, float* out_f, int* out_c, long long* out_m, uint64_t* out_first, const uint64_t* ks) {
    const int m = prm->m;
    if (blockIdx.x == 0 && threadIdx.x == 0 && out_m) *out_m = m;
    // one lane per (voxel, output channel): the per-voxel sums stay sequential in input order (what makes them
    // bit-identical to the reference), the channels of a voxel run side by side and share the index loads
    const int CH = 3 + (F ? fdim : 0);       // labels have their own kernel (gs_reduce_labels): different work per lane
    const long long total = (long long)m * CH;
    for (long long e = (long long)blockIdx.x * BS + threadIdx.x; e < total; e += (long long)gridDim.x * BS) {
        const int v = (int)(e / CH), c = (int)(e % CH);
        const int s = seg_start[v], en = seg_start[v + 1];
        const int count = en - s;
        const int row = row_of_voxel ? row_of_voxel[v] : v;
        if (c < 3) {
            float sum = 0.f;
            for (int j0 = s; j0 < en; j0 += GS_UNROLL) {               // GS_UNROLL gathers in flight, then the adds in input order
                float wv[GS_UNROLL];
#pragma unroll
                for (int u = 0; u < GS_UNROLL; ++u) wv[u] = P[3 * (size_t)vs[min(j0 + u, en - 1)] + c];
#pragma unroll
                for (int u = 0; u < GS_UNROLL; ++u) if (j0 + u < en) sum += wv[u];
            }
            const float a = (float)(1.0 / (double)count);          // cloud.h:120 via grid_subsampling.cpp:87
            out_p[3 * (size_t)row + c] = sum * a;
            if (c == 0 && out_first) out_first[v] = ks[s];
        } else {
            const int f = c - 3;
            float acc = 0.f;
            for (int j0 = s; j0 < en; j0 += GS_UNROLL) {
                float wv[GS_UNROLL];
#pragma unroll
                for (int u = 0; u < GS_UNROLL; ++u) wv[u] = F[(size_t)vs[min(j0 + u, en - 1)] * fdim + f];
#pragma unroll
                for (int u = 0; u < GS_UNROLL; ++u) if (j0 + u < en) acc += wv[u];
            }
            out_f[(size_t)row * fdim + f] = acc / (float)count;      // :90-94
        }
    }
}


// Reduction from packed records: 8 lanes per voxel, lane c owns word c of the record (coordinates and features are summed
// in input order, label columns vote).  The vote of a label lane is a packed byte-counter add per point (labels 0..12, at
// most 255 points); only a voxel whose maximum is shared by two labels (the tie goes by first-seen order), or one
// outside those limits, is re-scanned by the exact routines above.
// The reduction of one voxel (lane c = word c of the record) from a record source `get(j, c)`, j in [s, en): shared by the
// staged kernel below (records in LDS) and its fallback (records gathered from global memory).
template <class Get>
__device__ __forceinline__ void gs_reduce_voxel(Get get, int s, int en, int c, int fdim, int ldim, int row, const uint32_t* __restrict__ rec,
                                                const uint32_t* __restrict__ vs, int gs0, int gen, GsParams* prm, float* out_p, float* out_f, int* out_c) {
    const int CH = 3 + fdim, count = en - s;
    const bool is_label = c >= CH;
    float sum = 0.f;
    unsigned long long pk0 = 0ull, pk1 = 0ull;          // byte counters of labels 0..7 / 8..15
    bool exact = count > 255;                           // needs the exact routines
    for (int j0 = s; j0 < en; j0 += GS_UNROLL) {
        uint32_t wv[GS_UNROLL];
#pragma unroll
        for (int u = 0; u < GS_UNROLL; ++u) wv[u] = get(min(j0 + u, en - 1), c);
#pragma unroll
        for (int u = 0; u < GS_UNROLL; ++u) {
            if (j0 + u < en) {
                if (!is_label) sum += __uint_as_float(wv[u]);
                else {
                    const unsigned L = wv[u];
                    const unsigned long long inc = 1ull << ((L & 7u) * 8u);
                    exact |= L >= 13u;
                    pk0 += L < 8u ? inc : 0ull; pk1 += (L >= 8u && L < 16u) ? inc : 0ull;
                }
            }
        }
    }
    if (c < 3) {
        const float a = (float)(1.0 / (double)count);          // cloud.h:120 via grid_subsampling.cpp:87
        out_p[3 * (size_t)row + c] = sum * a;
    } else if (!is_label) {
        out_f[(size_t)row * fdim + (c - 3)] = sum / (float)count;      // :90-94
    } else {
        // labels outside [0,13) or more than 255 points, or a shared maximum: re-scanned from the record source (same values in the same order)
        const int best = voxel_label_packed(pk0, pk1, exact, [&](int j) { return (int)get(j, c); }, s, en, &prm->status);
        out_c[(size_t)row * ldim + (c - CH)] = best;
    }
}

// Staged version of the packed reduction.  A workgroup owns the voxels that START inside one tile of GS_T sorted
// positions; it first gathers all their points' records into LDS with every thread loading (a flat, fully parallel
// random gather runs at 2x the rate of per-voxel loops that chase seg_start -> index -> record), then the per-voxel
// sequential sums read LDS.  A tile whose voxels span more than GS_CAP points falls back to global gathers.
constexpr int GS_T = 256, GS_CAP = 512, GS_LD = REC_W + 1;
// first_ge[t] = first voxel that starts at or after position t * GS_T (t = 0 .. ntiles; first_ge[ntiles] = m): voxel v writes the
// entries of the tile boundaries in (start of v-1, start of v], so every entry has exactly one writer
__device__ __forceinline__ void gs_tile_index_body(const int* __restrict__ seg_start, const GsParams* prm, int n, int* first_ge) {
    const int m = prm->m, ntiles = (n + GS_T - 1) / GS_T;
    for (int v = blockIdx.x * BS + threadIdx.x; v <= m; v += gridDim.x * BS) {
        const int cur = v < m ? seg_start[v] : n + GS_T, prev = v > 0 ? seg_start[v - 1] : -1;     // v == m: sentinel past the end
        for (int t = prev / GS_T + (prev >= 0 ? 1 : 0); t <= ntiles && t * GS_T <= cur; ++t) if (t * GS_T > prev) first_ge[t] = v;
    }
}
__device__ __forceinline__ void gs_reduce_staged_body(const uint32_t* __restrict__ rec, int fdim, int ldim, int n, const int* __restrict__ first_ge,
                                                       const uint32_t* __restrict__ vs, const int* __restrict__ seg_start, GsParams* prm,
                                                       const int* __restrict__ row_of_voxel,
                                                       float* out_p, float* out_f, int* out_c, long long* out_m) {
    __shared__ uint32_t s_rec[GS_CAP * GS_LD];
    __shared__ int s_seg[GS_T + 2];
    const int m = prm->m, tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0 && out_m) *out_m = m;
    const int ntiles = (n + GS_T - 1) / GS_T;
    for (int b = blockIdx.x; b < ntiles; b += gridDim.x) {
        const int vfirst = first_ge[b], vend = first_ge[b + 1], nv = vend - vfirst;
        const int p0 = seg_start[vfirst], np = seg_start[vend] - p0;
        if (nv > 0 && np <= GS_CAP) {
            for (int i = tid; i <= nv; i += BS) s_seg[i] = seg_start[vfirst + i] - p0;
            uint4 r0[GS_CAP / BS], r1[GS_CAP / BS];
#pragma unroll
            for (int u = 0; u < GS_CAP / BS; ++u) {                 // all record gathers of the tile in flight together
                const int i = tid + u * BS;
                if (i < np) {
                    const uint4* src = reinterpret_cast<const uint4*>(rec + (size_t)vs[p0 + i] * REC_W);
                    r0[u] = src[0]; r1[u] = src[1];
                }
            }
#pragma unroll
            for (int u = 0; u < GS_CAP / BS; ++u) {
                const int i = tid + u * BS;
                if (i < np) {
                    uint32_t* d = s_rec + i * GS_LD;
                    d[0] = r0[u].x; d[1] = r0[u].y; d[2] = r0[u].z; d[3] = r0[u].w; d[4] = r1[u].x; d[5] = r1[u].y; d[6] = r1[u].z; d[7] = r1[u].w;
                }
            }
            __syncthreads();
            for (int e = tid; e < nv * REC_W; e += BS) {
                const int vl = e / REC_W, c = e % REC_W;
                if (c >= 3 + fdim + ldim) continue;
                const int v = vfirst + vl, sl = s_seg[vl], el = s_seg[vl + 1];
                gs_reduce_voxel([&](int j, int cc) { return s_rec[j * GS_LD + cc]; }, sl, el, c, fdim, ldim, row_of_voxel ? row_of_voxel[v] : v,
                                rec, vs, p0 + sl, p0 + el, prm, out_p, out_f, out_c);
            }
        } else if (nv > 0) {                                         // a tile with a very large voxel: gather from global memory
            for (int e = tid; e < nv * REC_W; e += BS) {
                const int vl = e / REC_W, c = e % REC_W;
                if (c >= 3 + fdim + ldim) continue;
                const int v = vfirst + vl, sg = seg_start[v], eg = seg_start[v + 1];
                gs_reduce_voxel([&](int j, int cc) { return rec[(size_t)vs[j] * REC_W + cc]; }, sg, eg, c, fdim, ldim, row_of_voxel ? row_of_voxel[v] : v,
                                rec, vs, sg, eg, prm, out_p, out_f, out_c);
            }
        }
        __syncthreads();
    }
}

// ---- kernel entry points: one cloud, or all clouds of a batch (blockIdx.y = cloud) -----------------------------
__global__ __launch_bounds__(BS) void gs_minmax_partial(const float* __restrict__ P, int n, float* partial) { gs_minmax_partial_body(P, n, partial); }
__global__ __launch_bounds__(BS) void gs_params(const float* partial, int nparts, float dl, GsParams* prm) { gs_params_body(partial, nparts, dl, prm); }
__global__ __launch_bounds__(BS) void gs_keys(const float* __restrict__ P, int n, GsParams* prm, uint64_t* keys, uint32_t* vals, const float* __restrict__ F, int fdim, const int* __restrict__ cls, int ldim, uint32_t* rec) { gs_keys_body(P, n, prm, keys, vals, F, fdim, cls, ldim, rec); }
__global__ __launch_bounds__(BS) void gs_tile_index(const int* __restrict__ seg_start, const GsParams* prm, int n, int* first_ge) { gs_tile_index_body(seg_start, prm, n, first_ge); }
__global__ __launch_bounds__(BS) void gs_reduce_packed(const uint32_t* __restrict__ rec, int fdim, int ldim, int n, const int* __restrict__ first_ge, const uint32_t* __restrict__ vs, const int* __restrict__ seg_start, GsParams* prm, const int* __restrict__ row_of_voxel, float* out_p, float* out_f, int* out_c, long long* out_m) { gs_reduce_staged_body(rec, fdim, ldim, n, first_ge, vs, seg_start, prm, row_of_voxel, out_p, out_f, out_c, out_m); }
__global__ __launch_bounds__(BS) void gs_heads_count(const uint64_t* __restrict__ ks, int n, int* bsum) { gs_heads_count_body(ks, n, bsum); }
__global__ __launch_bounds__(1024) void gs_heads_scan(int* bsum, int nb, GsParams* prm, int* seg_start, int n) { gs_heads_scan_body(bsum, nb, prm, seg_start, n); }
__global__ __launch_bounds__(BS) void gs_heads_write(const uint64_t* __restrict__ ks, int n, const int* bsum, int* seg_start) { gs_heads_write_body(ks, n, bsum, seg_start); }
__global__ __launch_bounds__(BS) void gs_reduce_labels(const int* __restrict__ cls, int ldim, const uint32_t* __restrict__ vs, const int* __restrict__ seg_start, GsParams* prm, const int* __restrict__ row_of_voxel, int* out_c) { gs_reduce_labels_body(cls, ldim, vs, seg_start, prm, row_of_voxel, out_c); }
__global__ __launch_bounds__(BS) void gs_reduce(const float* __restrict__ P, const float* __restrict__ F, int fdim, const int* __restrict__ cls, int ldim, const uint32_t* __restrict__ vs, const int* __restrict__ seg_start, GsParams* prm, const int* __restrict__ row_of_voxel, float* out_p, float* out_f, int* out_c, long long* out_m, uint64_t* out_first, const uint64_t* ks) { gs_reduce_body(P, F, fdim, cls, ldim, vs, seg_start, prm, row_of_voxel, out_p, out_f, out_c, out_m, out_first, ks); }

__global__ __launch_bounds__(BS) void gs_minmax_partial_b(CloudTab t, const float* __restrict__ P, float* partial) {
    const int r = blockIdx.y;
    gs_minmax_partial_body(P + 3 * (size_t)t.off[r], t.off[r + 1] - t.off[r], partial + (size_t)r * PB * 6);
}
__global__ __launch_bounds__(BS) void gs_params_b(const float* partial, float dl, GsParams* prm, int idx_bits) {
    gs_params_body(partial + (size_t)blockIdx.x * PB * 6, PB, dl, prm + blockIdx.x);
    if (threadIdx.x == 0 && idx_bits > 0) {      // composite sort words: the voxel key sits above the point index (gs_keys_body)
        const unsigned long long m = prm[blockIdx.x].key_or;
        prm[blockIdx.x].key_or = (m >> (64 - idx_bits)) ? ~0ull : (m << idx_bits);
    }
}
__global__ __launch_bounds__(BS) void gs_keys_b(CloudTab t, const float* __restrict__ P, GsParams* prm, uint64_t* keys, uint32_t* vals,
                                                const float* __restrict__ F, int fdim, const int* __restrict__ cls, int ldim, uint32_t* rec, int idx_bits) {
    const int r = blockIdx.y; const size_t o = (size_t)t.off[r];
    gs_keys_body(P + 3 * o, t.off[r + 1] - t.off[r], prm + r, keys + t.toff[r], vals ? vals + t.toff[r] : nullptr,
                 F ? F + o * fdim : nullptr, fdim, cls ? cls + o * ldim : nullptr, ldim, rec ? rec + o * REC_W : nullptr, idx_bits);
}
__global__ __launch_bounds__(BS) void gs_tile_index_b(CloudTab t, const int* __restrict__ seg_start, const GsParams* prm, int* first_ge) {
    const int r = blockIdx.y;
    gs_tile_index_body(seg_start + t.toff[r] + r, prm + r, t.off[r + 1] - t.off[r], first_ge + t.toff[r] / GS_T + 2 * r);
}
__global__ __launch_bounds__(BS) void gs_reduce_packed_b(CloudTab t, const uint32_t* __restrict__ rec, int fdim, int ldim, const int* __restrict__ first_ge,
                                                         const uint32_t* __restrict__ vs,
                                                         const int* __restrict__ seg_start, GsParams* prm, float* out_p, float* out_f, int* out_c, long long* out_m) {
    const int r = blockIdx.y; const size_t o = (size_t)t.off[r];
    gs_reduce_staged_body(rec + o * REC_W, fdim, ldim, t.off[r + 1] - t.off[r], first_ge + t.toff[r] / GS_T + 2 * r, vs + t.toff[r], seg_start + t.toff[r] + r, prm + r, nullptr,
                          out_p + 3 * o, out_f ? out_f + o * fdim : nullptr, out_c ? out_c + o * ldim : nullptr, out_m ? out_m + r : nullptr);
}
__global__ __launch_bounds__(BS) void gs_heads_count_b(CloudTab t, const uint64_t* __restrict__ ks, int* bsum, int nb_max, int shift, uint32_t* vals_out) {
    const int r = blockIdx.y, n = t.off[r + 1] - t.off[r];
    if ((int)blockIdx.x * CHUNK >= n) { if (threadIdx.x == 0) bsum[(size_t)r * nb_max + blockIdx.x] = 0; return; }
    gs_heads_count_body(ks + t.toff[r], n, bsum + (size_t)r * nb_max, shift, vals_out ? vals_out + t.toff[r] : nullptr);
}
__global__ __launch_bounds__(1024) void gs_heads_scan_b(CloudTab t, int* bsum, int nb_max, GsParams* prm, int* seg_start) {
    const int r = blockIdx.x, n = t.off[r + 1] - t.off[r];
    gs_heads_scan_body(bsum + (size_t)r * nb_max, (n + CHUNK - 1) / CHUNK, prm + r, seg_start + t.toff[r] + r, n);
}
__global__ __launch_bounds__(BS) void gs_heads_write_b(CloudTab t, const uint64_t* __restrict__ ks, const int* bsum, int nb_max, int* seg_start, int shift) {
    const int r = blockIdx.y, n = t.off[r + 1] - t.off[r];
    if ((int)blockIdx.x * CHUNK >= n) return;
    gs_heads_write_body(ks + t.toff[r], n, bsum + (size_t)r * nb_max, seg_start + t.toff[r] + r, shift);
}
__global__ __launch_bounds__(BS) void gs_reduce_b(CloudTab t, const float* __restrict__ P, const float* __restrict__ F, int fdim, const int* __restrict__ cls, int ldim,
                                                  const uint32_t* __restrict__ vs, const int* __restrict__ seg_start, GsParams* prm,
                                                  float* out_p, float* out_f, int* out_c, long long* out_m) {
    const int r = blockIdx.y; const size_t o = (size_t)t.off[r];
    gs_reduce_body(P + 3 * o, F ? F + o * fdim : nullptr, fdim, cls ? cls + o * ldim : nullptr, ldim, vs + t.toff[r], seg_start + t.toff[r] + r, prm + r, nullptr,
                   out_p + 3 * o, out_f ? out_f + o * fdim : nullptr, out_c ? out_c + o * ldim : nullptr, out_m ? out_m + r : nullptr, nullptr, nullptr);
}
__global__ __launch_bounds__(BS) void gs_reduce_labels_b(CloudTab t, const int* __restrict__ cls, int ldim, const uint32_t* __restrict__ vs, const int* __restrict__ seg_start,
                                                         GsParams* prm, int* out_c) {
    const int r = blockIdx.y; const size_t o = (size_t)t.off[r];
    gs_reduce_labels_body(cls + o * ldim, ldim, vs + t.toff[r], seg_start + t.toff[r] + r, prm + r, nullptr, out_c + o * ldim);
}

// ---- prune: the voxel grid of the superpoint partition (partition/ply_c/ply_c.cpp:289-383, AttributeGrid :160-287) -----------
// Same family as grid_subsampling with different conventions: origin = the bounding box minimum itself, bin = floor((p - min) / w),
// float32 position sums and uint32 colour sums in input order, label / object histograms instead of a majority vote, uint8 colours
// by truncation, and rows in the order in which the voxels are first met.
struct PruneParams { float mn[3]; float w; int nbin[3]; int pad; };
__global__ __launch_bounds__(BS) void prune_params(const float* partial, int nparts, float w, PruneParams* pp) {
    __shared__ float s_mm[(BS / 64) * 6];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = threadIdx.x; i < nparts; i += BS) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], partial[6 * i + d]); mx[d] = fmaxf(mx[d], partial[6 * i + 3 + d]); }
    }
    block_minmax3(mn, mx, s_mm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { pp->mn[d] = mn[d]; pp->nbin[d] = (int)(unsigned)ceilf((mx[d] - mn[d]) / w); }      // :329-331 (informational)
        pp->w = w;
    }
}
// key = the bin triple (21 bits each; a point on the upper face gets bin == nbin, as in the reference's map of triples)
__global__ __launch_bounds__(BS) void prune_keys(const float* __restrict__ P, int n, const PruneParams* pp, uint64_t* keys, uint32_t* vals, int* status) {
    const float x0 = pp->mn[0], y0 = pp->mn[1], z0 = pp->mn[2], w = pp->w;
    for (int i = blockIdx.x * BS + threadIdx.x; i < n; i += gridDim.x * BS) {
        const unsigned bx = (unsigned)floorf((P[3 * (size_t)i] - x0) / w), by = (unsigned)floorf((P[3 * (size_t)i + 1] - y0) / w), bz = (unsigned)floorf((P[3 * (size_t)i + 2] - z0) / w);   // :337-339
        if ((bx | by | bz) >> 21) atomicOr(status, 1);
        keys[i] = (uint64_t)bx | ((uint64_t)by << 21) | ((uint64_t)bz << 42);
        vals[i] = (uint32_t)i;
    }
}
// voxel -> (index of its first point, voxel): sorted by the first, this is the reference's row order (add_occurence :172-181)
__global__ __launch_bounds__(BS) void prune_first_keys(const uint32_t* __restrict__ vs, const int* __restrict__ seg_start, const GsParams* prm, uint64_t* fk, uint32_t* fv) {
    const int m = prm->m;
    for (int v = blockIdx.x * BS + threadIdx.x; v < m; v += gridDim.x * BS) { fk[v] = vs[seg_start[v]]; fv[v] = (uint32_t)v; }
}
__global__ __launch_bounds__(BS) void prune_rows(const uint32_t* __restrict__ fv, const GsParams* prm, int* row_of_voxel) {
    const int m = prm->m;
    for (int j = blockIdx.x * BS + threadIdx.x; j < m; j += gridDim.x * BS) row_of_voxel[fv[j]] = j;
}
// one lane per voxel: its points in input order (the sort is stable)
__global__ __launch_bounds__(BS) void prune_reduce(const float* __restrict__ P, const uint8_t* __restrict__ rgb, const uint8_t* __restrict__ lab, int n_labels,
                                                   const uint32_t* __restrict__ obj, int n_objects, const uint32_t* __restrict__ vs, const int* __restrict__ seg_start,
                                                   const GsParams* prm, const int* __restrict__ row_of_voxel, float* out_p, uint8_t* out_rgb, uint32_t* out_lab,
                                                   uint32_t* out_obj, long long* out_m, int* status) {
    const int m = prm->m;
    if (blockIdx.x == 0 && threadIdx.x == 0 && out_m) *out_m = m;
    for (int v = blockIdx.x * BS + threadIdx.x; v < m; v += gridDim.x * BS) {
        const int lo = seg_start[v], hi = seg_start[v + 1], row = row_of_voxel[v];
        float sx = 0.f, sy = 0.f, sz = 0.f; unsigned r = 0, g = 0, b = 0;
        uint32_t* hl = out_lab ? out_lab + (size_t)row * (n_labels + 1) : nullptr;
        uint32_t* ho = out_obj ? out_obj + (size_t)row * (n_objects + 1) : nullptr;
        if (hl) for (int c = 0; c <= n_labels; ++c) hl[c] = 0;
        if (ho) for (int c = 0; c <= n_objects; ++c) ho[c] = 0;
        for (int j = lo; j < hi; ++j) {
            const size_t i = vs[j];
            sx = sx + P[3 * i]; sy = sy + P[3 * i + 1]; sz = sz + P[3 * i + 2];                  // :277-279
            if (rgb) { r += rgb[3 * i]; g += rgb[3 * i + 1]; b += rgb[3 * i + 2]; }
            if (hl) { const int l = lab[i]; if (l <= n_labels) hl[l]++; else atomicOr(status, 2); }       // .at() would throw in the reference
            if (ho) { const unsigned o = obj[i]; if (o <= (unsigned)n_objects) ho[o]++; else atomicOr(status, 2); }
        }
        const float cnt = (float)(hi - lo);
        out_p[3 * (size_t)row] = sx / cnt; out_p[3 * (size_t)row + 1] = sy / cnt; out_p[3 * (size_t)row + 2] = sz / cnt;       // :366-369
        if (out_rgb) { out_rgb[3 * (size_t)row] = (uint8_t)((float)r / cnt); out_rgb[3 * (size_t)row + 1] = (uint8_t)((float)g / cnt); out_rgb[3 * (size_t)row + 2] = (uint8_t)((float)b / cnt); }   // :373-375
    }
}

struct GsState {
    RadixSorter sorter;
    DevBuf keys, vals, partial, params, bsum, seg, in_p, in_f, in_c, out_p, out_f, out_c, out_m, row, rec, tidx, pparams, fk, fv, pstat, andor;
    RadixSorter sorter2;
    size_t last_m = 0, last_fdim = 0, last_ldim = 0;
    int last_clouds = 0;        // clouds of the last device-flavour call (their GsParams sit at the front of `params`)
};
GsState& gs(hipStream_t st = nullptr) { return per_stream<GsState>(st); }

}  // namespace

int subsample_order_reference(const uint64_t* d_ks, const uint32_t* d_vs, const int* d_seg_start, const int* d_m, int n_host,
                              int* d_row_of_voxel, hipStream_t s);

int grid_subsample_device(const float* d_p, size_t n, const float* d_f, size_t fdim, const int32_t* d_c, size_t ldim, float dl,
                          int order, float* d_op, float* d_of, int32_t* d_oc, int64_t* d_om, hipStream_t s) {
    GsState& S = gs(s);
    S.last_clouds = 1;
    const int ni = (int)n;
    const int gmm = std::max(1, std::min((ni + BS - 1) / BS, 1024));
    const int nb = (ni + CHUNK - 1) / CHUNK;
    SSDR_TRY(S.keys.reserve(8 * n + 16)); SSDR_TRY(S.vals.reserve(4 * n + 16));
    SSDR_TRY(S.partial.reserve(24 * 1024)); SSDR_TRY(S.params.reserve(sizeof(GsParams)));
    SSDR_TRY(S.bsum.reserve(4 * (size_t)nb + 16)); SSDR_TRY(S.seg.reserve(4 * (n + 2)));
    GsParams* prm = S.params.as<GsParams>();
    hipLaunchKernelGGL(gs_minmax_partial, dim3(gmm), dim3(BS), 0, s, d_p, ni, S.partial.as<float>());
    hipLaunchKernelGGL(gs_params, dim3(1), dim3(BS), 0, s, S.partial.as<float>(), gmm, dl, prm);
    const int g = std::max(1, std::min((ni + BS - 1) / BS, ctx().num_cu * 16));
    const bool packed = 3 + fdim + ldim <= (size_t)REC_W;      // rows of at most 8 words reduce from packed records
    uint32_t* rec = nullptr;
    if (packed) { SSDR_TRY(S.rec.reserve(4 * (size_t)REC_W * n + 16)); rec = S.rec.as<uint32_t>(); }
    hipLaunchKernelGGL(gs_keys, dim3(g), dim3(BS), 0, s, d_p, ni, prm, S.keys.as<uint64_t>(), S.vals.as<uint32_t>(), d_f, (int)fdim, (const int*)d_c, (int)ldim, rec);
    SSDR_TRY(S.sorter.sort(S.keys.as<uint64_t>(), S.vals.as<uint32_t>(), ni, nullptr, s));
    hipLaunchKernelGGL(gs_heads_count, dim3(nb), dim3(BS), 0, s, S.keys.as<uint64_t>(), ni, S.bsum.as<int>());
    hipLaunchKernelGGL(gs_heads_scan, dim3(1), dim3(1024), 0, s, S.bsum.as<int>(), nb, prm, S.seg.as<int>(), ni);
    hipLaunchKernelGGL(gs_heads_write, dim3(nb), dim3(BS), 0, s, S.keys.as<uint64_t>(), ni, S.bsum.as<int>(), S.seg.as<int>());
    const int* row = nullptr;
    if (order == SSDR_ORDER_REFERENCE) {
        SSDR_TRY(S.row.reserve(4 * n + 16));
        SSDR_TRY(subsample_order_reference(S.keys.as<uint64_t>(), S.vals.as<uint32_t>(), S.seg.as<int>(), &prm->m, ni, S.row.as<int>(), s));
        row = S.row.as<int>();
    }
    if (packed) {
        SSDR_TRY(S.tidx.reserve(4 * (n / GS_T + 4)));
        hipLaunchKernelGGL(gs_tile_index, dim3(g), dim3(BS), 0, s, S.seg.as<int>(), prm, ni, S.tidx.as<int>());
        hipLaunchKernelGGL(gs_reduce_packed, dim3(g), dim3(BS), 0, s, rec, (int)fdim, (int)ldim, ni, S.tidx.as<int>(), S.vals.as<uint32_t>(), S.seg.as<int>(), prm, row, d_op, d_of, d_oc, (long long*)d_om);
    } else {
        hipLaunchKernelGGL(gs_reduce, dim3(g), dim3(BS), 0, s, d_p, d_f, (int)fdim, d_c, (int)ldim, S.vals.as<uint32_t>(), S.seg.as<int>(), prm,
                           row, d_op, d_of, d_oc, (long long*)d_om, (uint64_t*)nullptr, S.keys.as<uint64_t>());
        if (d_c) hipLaunchKernelGGL(gs_reduce_labels, dim3(g), dim3(BS), 0, s, d_c, (int)ldim, S.vals.as<uint32_t>(), S.seg.as<int>(), prm, row, d_oc);
    }
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int prune_device(const float* d_p, size_t n, float w, const uint8_t* d_rgb, const uint8_t* d_lab, int n_labels, const uint32_t* d_obj, int n_objects,
                 float* d_op, uint8_t* d_orgb, uint32_t* d_olab, uint32_t* d_oobj, int64_t* d_om, hipStream_t s) {
    GsState& S = gs(s);
    const int ni = (int)n;
    const int gmm = std::max(1, std::min((ni + BS - 1) / BS, 1024));
    const int nb = (ni + CHUNK - 1) / CHUNK;
    SSDR_TRY(S.keys.reserve(8 * n + 16)); SSDR_TRY(S.vals.reserve(4 * n + 16));
    SSDR_TRY(S.partial.reserve(24 * 1024)); SSDR_TRY(S.params.reserve(sizeof(GsParams))); SSDR_TRY(S.pparams.reserve(sizeof(PruneParams)));
    SSDR_TRY(S.bsum.reserve(4 * (size_t)nb + 16)); SSDR_TRY(S.seg.reserve(4 * (n + 2)));
    SSDR_TRY(S.fk.reserve(8 * n + 16)); SSDR_TRY(S.fv.reserve(4 * n + 16)); SSDR_TRY(S.row.reserve(4 * n + 16)); SSDR_TRY(S.pstat.reserve(16));
    GsParams* prm = S.params.as<GsParams>(); PruneParams* pp = S.pparams.as<PruneParams>();
    SSDR_HIP(hipMemsetAsync(S.pstat.p, 0, 16, s));
    hipLaunchKernelGGL(gs_minmax_partial, dim3(gmm), dim3(BS), 0, s, d_p, ni, S.partial.as<float>());
    hipLaunchKernelGGL(prune_params, dim3(1), dim3(BS), 0, s, S.partial.as<float>(), gmm, w, pp);
    const int g = std::max(1, std::min((ni + BS - 1) / BS, ctx().num_cu * 16));
    hipLaunchKernelGGL(prune_keys, dim3(g), dim3(BS), 0, s, d_p, ni, pp, S.keys.as<uint64_t>(), S.vals.as<uint32_t>(), S.pstat.as<int>());
    SSDR_TRY(S.sorter.sort(S.keys.as<uint64_t>(), S.vals.as<uint32_t>(), ni, nullptr, s, 63));
    hipLaunchKernelGGL(gs_heads_count, dim3(nb), dim3(BS), 0, s, S.keys.as<uint64_t>(), ni, S.bsum.as<int>());
    hipLaunchKernelGGL(gs_heads_scan, dim3(1), dim3(1024), 0, s, S.bsum.as<int>(), nb, prm, S.seg.as<int>(), ni);
    hipLaunchKernelGGL(gs_heads_write, dim3(nb), dim3(BS), 0, s, S.keys.as<uint64_t>(), ni, S.bsum.as<int>(), S.seg.as<int>());
    // rows in first-met order: sort the voxels by the index of their first point
    hipLaunchKernelGGL(prune_first_keys, dim3(g), dim3(BS), 0, s, S.vals.as<uint32_t>(), S.seg.as<int>(), prm, S.fk.as<uint64_t>(), S.fv.as<uint32_t>());
    SSDR_TRY(S.sorter2.sort(S.fk.as<uint64_t>(), S.fv.as<uint32_t>(), ni, &prm->m, s, 32));
    hipLaunchKernelGGL(prune_rows, dim3(g), dim3(BS), 0, s, S.fv.as<uint32_t>(), prm, S.row.as<int>());
    hipLaunchKernelGGL(prune_reduce, dim3(g), dim3(BS), 0, s, d_p, d_rgb, d_lab, n_labels, d_obj, n_objects, S.vals.as<uint32_t>(), S.seg.as<int>(), prm, S.row.as<int>(),
                       d_op, d_orgb, d_olab, d_oobj, (long long*)d_om, S.pstat.as<int>());
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

// frontend.hip: bucket partition + per-bucket LDS reduction (rows of at most 7 words, grids of at most 16384 buckets of 1024 voxels)
bool frontend_fits(size_t fdim, size_t ldim);
int frontend_batch_device(const float* d_p, const float* d_f, size_t fdim, const int32_t* d_c, size_t ldim, const int64_t* room_off, size_t nr, float dl,
                          float* d_op, float* d_of, int32_t* d_oc, int64_t* d_om, GsParams* prm, hipStream_t s);

int grid_subsample_batch_device(const float* d_p, const float* d_f, size_t fdim, const int32_t* d_c, size_t ldim, const int64_t* room_off, size_t nr, float dl,
                                float* d_op, float* d_of, int32_t* d_oc, int64_t* d_om, hipStream_t s, int method) {
    GsState& S = gs(s);
    if (method != SSDR_SUBSAMPLE_SORT && frontend_fits(fdim, ldim)) {
        S.last_clouds = (int)nr;
        SSDR_TRY(S.params.reserve(sizeof(GsParams) * nr));
        return frontend_batch_device(d_p, d_f, fdim, d_c, ldim, room_off, nr, dl, d_op, d_of, d_oc, d_om, S.params.as<GsParams>(), s);
    }
    CloudTab t; t.nr = (int)nr;
    int toff = 0, maxn = 0; std::vector<int> n_host(nr);
    for (size_t r = 0; r < nr; ++r) {
        const long n = (long)(room_off[r + 1] - room_off[r]);
        if (n <= 0 || room_off[r + 1] > 0x3fffffff) { set_error("grid_subsample_batch: every cloud needs >= 1 point"); return n <= 0 ? SSDR_ERR_EMPTY : SSDR_ERR_INVALID; }
        t.off[r] = (int)room_off[r]; t.toff[r] = toff; n_host[r] = (int)n; maxn = std::max(maxn, (int)n);
        toff += ((int)n + RADIX_TILE - 1) / RADIX_TILE * RADIX_TILE;
        if (toff > 0x3fffffff) { set_error("grid_subsample_batch: too many points"); return SSDR_ERR_INVALID; }
    }
    t.off[nr] = (int)room_off[nr]; t.toff[nr] = toff;
    S.last_clouds = (int)nr;
    const int nb_max = (maxn + CHUNK - 1) / CHUNK;
    SSDR_TRY(S.keys.reserve(8 * (size_t)toff + 16)); SSDR_TRY(S.vals.reserve(4 * (size_t)toff + 16));
    SSDR_TRY(S.partial.reserve(24 * (size_t)PB * nr)); SSDR_TRY(S.params.reserve(sizeof(GsParams) * nr)); SSDR_TRY(S.andor.reserve(16 * (size_t)nr));
    SSDR_TRY(S.bsum.reserve(4 * (size_t)nb_max * nr + 16)); SSDR_TRY(S.seg.reserve(4 * ((size_t)toff + nr + 2)));
    GsParams* prm = S.params.as<GsParams>();
    const unsigned R = (unsigned)nr;
    hipLaunchKernelGGL(gs_minmax_partial_b, dim3(PB, R), dim3(BS), 0, s, t, d_p, S.partial.as<float>());
    // sort words (voxel key << idx_bits) | index inside the cloud: 8 bytes per point through the sorter instead of 12
    int idx_bits = 1; while ((1L << idx_bits) < (long)maxn) ++idx_bits;
    hipLaunchKernelGGL(gs_params_b, dim3(R), dim3(BS), 0, s, S.partial.as<float>(), dl, prm, idx_bits);
    const int g = std::max(1, std::min((maxn + BS - 1) / BS, 256));
    const bool packed = 3 + fdim + ldim <= (size_t)REC_W;
    uint32_t* rec = nullptr;
    if (packed) { SSDR_TRY(S.rec.reserve(4 * (size_t)REC_W * (size_t)room_off[nr] + 16)); rec = S.rec.as<uint32_t>(); }
    // voxel keys of a room at this grid size span 17-24 bits: three digit passes.  The keys are written into the sorter's other
    // buffer, so that after an odd number of passes the sorted pairs sit in S.keys / S.vals without a copy (any other count
    // still ends there, through the sorter's final copy).
    SSDR_TRY(S.sorter.reserve((size_t)toff));
    uint64_t* const k_in = S.sorter.alt_keys(); uint32_t* const v_in = S.sorter.alt_vals();
    (void)v_in;
    hipLaunchKernelGGL(gs_keys_b, dim3(g, R), dim3(BS), 0, s, t, d_p, prm, k_in, (uint32_t*)nullptr, d_f, (int)fdim, (const int*)d_c, (int)ldim, rec, idx_bits);
    // the key range follows from the grid dimensions (gs_params): no pass over the keys to find the digits that vary
    static_assert(sizeof(GsParams) % 8 == 0 && offsetof(GsParams, key_or) == offsetof(GsParams, key_and) + 8, "AND / OR pair");
    SSDR_HIP(hipMemcpy2DAsync(S.andor.p, 16, reinterpret_cast<const char*>(prm) + offsetof(GsParams, key_and), sizeof(GsParams), 16, nr, hipMemcpyDeviceToDevice, s));
    SSDR_TRY(S.sorter.sort_segments(S.keys.as<uint64_t>(), nullptr, (int)nr, t.toff, n_host.data(), nullptr, s, 64 - idx_bits, true, S.andor.as<unsigned long long>(), idx_bits));
    hipLaunchKernelGGL(gs_heads_count_b, dim3(nb_max, R), dim3(BS), 0, s, t, S.keys.as<uint64_t>(), S.bsum.as<int>(), nb_max, idx_bits, S.vals.as<uint32_t>());
    hipLaunchKernelGGL(gs_heads_scan_b, dim3(R), dim3(1024), 0, s, t, S.bsum.as<int>(), nb_max, prm, S.seg.as<int>());
    hipLaunchKernelGGL(gs_heads_write_b, dim3(nb_max, R), dim3(BS), 0, s, t, S.keys.as<uint64_t>(), S.bsum.as<int>(), nb_max, S.seg.as<int>(), idx_bits);
    if (packed) {
        SSDR_TRY(S.tidx.reserve(4 * ((size_t)toff / GS_T + 2 * nr + 4)));
        hipLaunchKernelGGL(gs_tile_index_b, dim3(g, R), dim3(BS), 0, s, t, S.seg.as<int>(), prm, S.tidx.as<int>());
        hipLaunchKernelGGL(gs_reduce_packed_b, dim3(g, R), dim3(BS), 0, s, t, rec, (int)fdim, (int)ldim, S.tidx.as<int>(), S.vals.as<uint32_t>(), S.seg.as<int>(), prm, d_op, d_of, d_oc, (long long*)d_om);
    } else {
        hipLaunchKernelGGL(gs_reduce_b, dim3(g, R), dim3(BS), 0, s, t, d_p, d_f, (int)fdim, d_c, (int)ldim, S.vals.as<uint32_t>(), S.seg.as<int>(), prm,
                           d_op, d_of, d_oc, (long long*)d_om);
        if (d_c) hipLaunchKernelGGL(gs_reduce_labels_b, dim3(g, R), dim3(BS), 0, s, t, d_c, (int)ldim, S.vals.as<uint32_t>(), S.seg.as<int>(), prm, d_oc);
    }
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

}  // namespace ssdr

using namespace ssdr;

// process-wide (not per stream, not per thread): read once per ssdr_grid_subsample_batch_dev call, set by ssdr_grid_subsample_set_method
static std::atomic<int> g_batch_method{[] { const char* e = getenv("SSDR_SUBSAMPLE_METHOD"); return e && !strcmp(e, "sort") ? SSDR_SUBSAMPLE_SORT : SSDR_SUBSAMPLE_AUTO; }()};

extern "C" {

int ssdr_grid_subsample_dev(const float* d_points, size_t n, const float* d_features, size_t fdim, const int32_t* d_classes,
                            size_t ldim, float sampleDl, int order, float* d_out_points, float* d_out_features,
                            int32_t* d_out_classes, int64_t* d_out_m, void* stream) {
    if (!d_points || !d_out_points || n == 0 || n > 0x3fffffff) { set_error("grid_subsample: bad points / n"); return n == 0 ? SSDR_ERR_EMPTY : SSDR_ERR_INVALID; }
    if (!(sampleDl > 0.f)) { set_error("grid_subsample: sampleDl must be > 0"); return SSDR_ERR_INVALID; }
    if (d_features && (!fdim || !d_out_features)) { set_error("grid_subsample: features given without fdim / output"); return SSDR_ERR_INVALID; }
    if (d_classes && (!ldim || !d_out_classes)) { set_error("grid_subsample: classes given without ldim / output"); return SSDR_ERR_INVALID; }
    if (order != SSDR_ORDER_REFERENCE && order != SSDR_ORDER_KEY) { set_error("grid_subsample: unknown order %d", order); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    return grid_subsample_device(d_points, n, d_features, d_features ? fdim : 0, d_classes, d_classes ? ldim : 0, sampleDl, order,
                                 d_out_points, d_out_features, d_out_classes, d_out_m, pick_stream(stream));
}

int ssdr_grid_subsample_batch_dev(const float* d_points, const float* d_features, size_t fdim, const int32_t* d_classes, size_t ldim,
                                  const int64_t* cloud_offsets, size_t num_clouds, float sampleDl,
                                  float* d_out_points, float* d_out_features, int32_t* d_out_classes, int64_t* d_out_m, void* stream) {
    if (!d_points || !d_out_points || !cloud_offsets || !d_out_m || num_clouds == 0 || num_clouds > RADIX_MAX_SEG) { set_error("grid_subsample_batch: bad arguments (1..%d clouds)", RADIX_MAX_SEG); return SSDR_ERR_INVALID; }
    if (!(sampleDl > 0.f)) { set_error("grid_subsample_batch: sampleDl must be > 0"); return SSDR_ERR_INVALID; }
    if (d_features && (!fdim || !d_out_features)) { set_error("grid_subsample_batch: features given without fdim / output"); return SSDR_ERR_INVALID; }
    if (d_classes && (!ldim || !d_out_classes)) { set_error("grid_subsample_batch: classes given without ldim / output"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    return grid_subsample_batch_device(d_points, d_features, d_features ? fdim : 0, d_classes, d_classes ? ldim : 0, cloud_offsets, num_clouds, sampleDl,
                                       d_out_points, d_out_features, d_out_classes, d_out_m, pick_stream(stream), g_batch_method.load());
}

/* which implementation ssdr_grid_subsample_batch_dev uses: SSDR_SUBSAMPLE_AUTO (default: the bucket partition of frontend.hip for rows of at most 7 words,
 * the sort-based path otherwise) or SSDR_SUBSAMPLE_SORT (always the sort: any grid, any voxel population).  A cloud the partition path cannot take is
 * reported by ssdr_grid_subsample_status (bits 2, 4); the caller then repeats the call with SSDR_SUBSAMPLE_SORT. */
int ssdr_grid_subsample_set_method(int method) {
    if (method != SSDR_SUBSAMPLE_AUTO && method != SSDR_SUBSAMPLE_SORT) { set_error("grid_subsample_set_method: unknown method %d", method); return SSDR_ERR_INVALID; }
    g_batch_method.store(method);
    return SSDR_OK;
}

int ssdr_grid_subsample(const float* points, size_t n, const float* features, size_t fdim, const int32_t* classes, size_t ldim,
                        float sampleDl, int order, size_t* out_m) {
    if (!points || !out_m) { set_error("grid_subsample: NULL argument"); return SSDR_ERR_INVALID; }
    if (n == 0) { set_error("Error"); return SSDR_ERR_EMPTY; }
    SSDR_TRY(ensure_init());
    GsState& S = gs(); Context& c = ctx(); hipStream_t s = c.stream;
    if (!features) fdim = 0;
    if (!classes) ldim = 0;
    SSDR_TRY(S.in_p.reserve(12 * n)); SSDR_TRY(S.out_p.reserve(12 * n)); SSDR_TRY(S.out_m.reserve(16));
    SSDR_HIP(hipMemcpyAsync(S.in_p.p, points, 12 * n, hipMemcpyHostToDevice, s));
    if (fdim) { SSDR_TRY(S.in_f.reserve(4 * n * fdim)); SSDR_TRY(S.out_f.reserve(4 * n * fdim)); SSDR_HIP(hipMemcpyAsync(S.in_f.p, features, 4 * n * fdim, hipMemcpyHostToDevice, s)); }
    if (ldim) { SSDR_TRY(S.in_c.reserve(4 * n * ldim)); SSDR_TRY(S.out_c.reserve(4 * n * ldim)); SSDR_HIP(hipMemcpyAsync(S.in_c.p, classes, 4 * n * ldim, hipMemcpyHostToDevice, s)); }
    SSDR_HIP(hipEventRecord(c.ev0, s));
    SSDR_TRY(ssdr_grid_subsample_dev(S.in_p.as<float>(), n, fdim ? S.in_f.as<float>() : nullptr, fdim, ldim ? S.in_c.as<int32_t>() : nullptr, ldim,
                                     sampleDl, order, S.out_p.as<float>(), fdim ? S.out_f.as<float>() : nullptr,
                                     ldim ? S.out_c.as<int32_t>() : nullptr, S.out_m.as<int64_t>(), s));
    SSDR_HIP(hipEventRecord(c.ev1, s));
    GsParams h;
    SSDR_HIP(hipMemcpyAsync(&h, S.params.p, sizeof(h), hipMemcpyDeviceToHost, s));
    SSDR_HIP(hipStreamSynchronize(s));
    SSDR_HIP(hipEventElapsedTime(&c.last_ms, c.ev0, c.ev1));
    if (h.status) { set_error("grid_subsample: a voxel holds more than %d distinct labels in one column (unsupported)", LAB_CAP); return SSDR_ERR_UNSUPPORTED; }
    S.last_m = (size_t)h.m; S.last_fdim = fdim; S.last_ldim = ldim;
    *out_m = S.last_m;
    if (S.last_m == 0) { set_error("Error"); return SSDR_ERR_EMPTY; }
    return SSDR_OK;
}

int ssdr_grid_subsample_fetch(float* out_points, float* out_features, int32_t* out_classes) {
    SSDR_TRY(ensure_init());
    GsState& S = gs(); hipStream_t s = ctx().stream;
    if (out_points) SSDR_HIP(hipMemcpyAsync(out_points, S.out_p.p, 12 * S.last_m, hipMemcpyDeviceToHost, s));
    if (out_features && S.last_fdim) SSDR_HIP(hipMemcpyAsync(out_features, S.out_f.p, 4 * S.last_m * S.last_fdim, hipMemcpyDeviceToHost, s));
    if (out_classes && S.last_ldim) SSDR_HIP(hipMemcpyAsync(out_classes, S.out_c.p, 4 * S.last_m * S.last_ldim, hipMemcpyDeviceToHost, s));
    SSDR_HIP(hipStreamSynchronize(s));
    return SSDR_OK;
}


/* libply_c.prune (partition/ply_c/ply_c.cpp:289-383): device flavour.  Outputs are sized for n rows; *d_out_m receives the voxel count. */
int ssdr_prune_dev(const float* d_xyz, size_t n, float voxel_size, const uint8_t* d_rgb, const uint8_t* d_labels, int n_labels, const uint32_t* d_objects, int n_objects,
                   float* d_out_xyz, uint8_t* d_out_rgb, uint32_t* d_out_labels, uint32_t* d_out_objects, int64_t* d_out_m, void* stream) {
    if (!d_xyz || !d_out_xyz || !d_out_m || !(voxel_size > 0.f) || n > 0x3fffffff || n_labels < 0 || n_objects < 0 || (n_labels > 0 && (!d_labels || !d_out_labels)) ||
        (n_objects > 0 && (!d_objects || !d_out_objects))) { set_error("prune: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream);
    if (n == 0) { SSDR_HIP(hipMemsetAsync(d_out_m, 0, 8, s)); return SSDR_OK; }
    return prune_device(d_xyz, n, voxel_size, d_rgb, n_labels > 0 ? d_labels : nullptr, n_labels, n_objects > 0 ? d_objects : nullptr, n_objects,
                        d_out_xyz, d_out_rgb, n_labels > 0 ? d_out_labels : nullptr, n_objects > 0 ? d_out_objects : nullptr, d_out_m, s);
}
/* what the kernels of the last device-flavour grid subsample on `stream` found (waits for it): bit 0 = a voxel with more than LAB_CAP distinct
 * labels in one column (its majority label may be wrong) — the host flavour reports the same through its return value */
int ssdr_grid_subsample_status(void* stream, int32_t* out_status) {
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream);
    GsState& S = gs(s);
    int st = 0;
    SSDR_HIP(hipStreamSynchronize(s));
    if (S.params.p && S.last_clouds > 0) {
        std::vector<GsParams> h((size_t)S.last_clouds);
        SSDR_HIP(hipMemcpy(h.data(), S.params.p, sizeof(GsParams) * h.size(), hipMemcpyDeviceToHost));
        for (auto& g : h) st |= g.status;
    }
    if (out_status) *out_status = st;
    if (st) { set_error("grid_subsample: device status 0x%x (1 = a voxel holds more than %d distinct labels in one column; 2 = a cloud's grid is too large for "
                        "the batch flavour (partition path: more than 16384 buckets of 1024 voxels, or coordinates outside the grid; sort path: no room for the index in "
                        "the sort words); 4 = a voxel of more than 1024 points in the partition path: repeat with ssdr_grid_subsample_set_method(SSDR_SUBSAMPLE_SORT), or "
                        "use ssdr_grid_subsample_dev)", st, LAB_CAP); return SSDR_ERR_UNSUPPORTED; }
    return SSDR_OK;
}

/* status of the last ssdr_prune_dev on `stream` (waits for it): bit 0 = more than 2^21 bins along an axis, bit 1 = a label / object id above its declared maximum */
int ssdr_prune_status(void* stream, int32_t* out_status) {
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream);
    GsState& S = gs(s);
    int st = 0;
    SSDR_HIP(hipStreamSynchronize(s));
    if (S.pstat.p) SSDR_HIP(hipMemcpy(&st, S.pstat.p, 4, hipMemcpyDeviceToHost));
    if (out_status) *out_status = st;
    if (st) { set_error("prune: device status 0x%x (1 = more than 2^21 bins along an axis, 2 = label / object id above its declared maximum)", st); return SSDR_ERR_INVALID; }
    return SSDR_OK;
}

}
