// Grid subsampling of a batch of room-scale clouds without a global sort: one partition pass into spatial buckets, then one
// workgroup per bucket that orders and reduces its voxels in LDS.
//
// Reference: S3/utils/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106.  Same contract as subsample.hip
// (fp32 voxel arithmetic :27-31/:53-56, per-voxel sums in INPUT order :59-70, `sum * (float)(1.0/count)` for positions and
// `sum / (float)count` for features :87-95, majority label with the reference's tie rule :97-101); rows come out by ascending voxel key.
//
// The sort-based formulation (subsample.hip) moved ~5.4 GB through HBM for 0.48 GB of algorithmic bytes (three digit passes over
// 13.4 M sort words plus a random 32-byte gather per point).  Here a point travels twice:
//
//   minmax / params   bounding box -> grid origin and dimensions (the reference's :27-31), bucket geometry: a bucket is a block of
//                     2^sx x 8 x 8 voxels (sx = 3, or 4 when that leaves more than FE_NBMAX buckets)
//   count             every chunk of FE_CH points histograms its points over the cloud's buckets in LDS -> one row of a
//                     [chunk][bucket] table
//   colscan / bscan   column prefix over the chunks and prefix over the buckets -> where each chunk's points of each bucket go; one
//                     work item per non-empty bucket
//   scatter           second read of the points: 32-byte records (xyz, index in the cloud | features, labels) written behind LDS cursors,
//                     a bucket's records contiguous (order inside a bucket is whatever the LDS atomics give: the index travels)
//   reduce            one workgroup per bucket: records -> LDS, LDS histogram over the bucket's <= 1024 voxels, every voxel's members
//                     ranked by their index (input order), sequential sums, label vote; the bucket's rows (32 bytes: row + voxel id) go
//                     back over the bucket's own records, with its occupancy per local row (y, z)
//   rowpre / rowscan  number of occupied voxels in front of every (row, bucket) in key order (key = ix + nx (iy + ny iz): rows of x)
//   move              every bucket's rows to their final position
//
// HBM traffic: 12 N (minmax) + 12 N (count) + 28 N + 32 N (scatter) + 32 N + 32 M (reduce) + 32 M + 28 M (move) bytes for N points and M
// voxels, against 28 (N + M) algorithmic.  Clouds whose grid does not fit (more than FE_NBMAX buckets of 1024 voxels) or that
// hold a voxel of more than FE_CAP points are flagged (ssdr_grid_subsample_status) and belong to the sort-based entry point.
#include "ssdr_internal.hpp"
#include "block_prims.hpp"
#include "voxel_label.hpp"
#include "subsample_types.hpp"

namespace ssdr {
namespace {

constexpr int FE_NBMAX = 16384;      // buckets per cloud: the count / scatter kernels keep one LDS word per bucket (64 KiB)
constexpr int FE_CH = 32768;         // points per chunk of the count / scatter kernels
constexpr int FE_NT = 1024;          // their workgroup size
constexpr int FE_PPT = 12, FE_SNT = 1024;      // points per thread and step of the scatter, its workgroup size: 12288 points per step leave ~12 records = three
                                              // whole lines in a bucket at a time, and one workgroup per CU halves the lines the L2 has open (PPT 4: 0.51 ms, 8: 0.44, 12: 0.36;
                                              // 512 threads x 24 the same, x 32 = the whole chunk in one step spills: 0.37)
#ifndef FE_PAIR_STORES
#define FE_PAIR_STORES 1
#endif
constexpr int FE_LR = 64;            // local rows (y, z) of a bucket: 8 x 8
constexpr int FE_VMAX = 1024;        // voxels per bucket (sx <= 4)
#ifndef FE_RNT_OPT
#define FE_RNT_OPT 128
#endif
constexpr int FE_RNT = FE_RNT_OPT;   // workgroup size of the reduction: 128 threads with eight records each (two waves per barrier; 256 threads: front end 1.154-1.161 against
                                     // 1.106-1.110 ms on one box, tools/gpu_fe_ab2.sh; 512 threads were 13 % slower in round 4; 64 threads leave two waves per SIMD by LDS)
#ifndef FE_WALK_OPT
#define FE_WALK_OPT 4
#endif
constexpr int FE_WALK = FE_WALK_OPT; // members of a voxel whose gathers the ordered walk keeps in flight
constexpr int FE_CAP = 1024;         // records of a bucket (or of a slice of it) the reduction holds in LDS
constexpr int FE_RPT = FE_CAP / FE_RNT;
constexpr int FE_SLICES = 62;        // slices of a bucket with more than FE_CAP records
constexpr int FE_ROWCAP = FE_LR * FE_NBMAX;      // padded rows (y, z) per cloud: at most 64 per bucket column

struct FeGeom {
    float org[3]; float dl;
    int nx, ny, nz, sx;
    int nbx, nby, nbz, nb;
    int ok, n, pad0, pad1;
};
// one non-empty bucket: everything its workgroup needs in one 32-byte load
struct FeItem { unsigned base, n, rb, sx; float org[3]; float dl; };       // base: first record (absolute); rb = cloud << 16 | bucket

struct FeTab { int nr; int chunks_max; int ovf_cap; int n_total; int off[RADIX_MAX_SEG + 1]; };

// voxel coordinates of a point (grid_subsampling.cpp:53-56: floor((p - origin) / dl), the same fp32 operations); false when the point
// falls outside the grid the bounding box gives (a wrapped size_t in the reference: only NaNs and overflowing coordinates get here)
__device__ __forceinline__ bool fe_voxel(const FeGeom& g, float x, float y, float z, int& ix, int& iy, int& iz) {
    const float fx = floorf((x - g.org[0]) / g.dl), fy = floorf((y - g.org[1]) / g.dl), fz = floorf((z - g.org[2]) / g.dl);
    const bool ok = fx >= 0.f && fx < (float)g.nx && fy >= 0.f && fy < (float)g.ny && fz >= 0.f && fz < (float)g.nz;
    ix = (int)fx; iy = (int)fy; iz = (int)fz;
    return ok;
}
__device__ __forceinline__ int fe_bucket(const FeGeom& g, int ix, int iy, int iz) { return (ix >> g.sx) + g.nbx * ((iy >> 3) + g.nby * (iz >> 3)); }
// voxel of a record inside its bucket: x fastest, then the local row (y, z)
__device__ __forceinline__ int fe_vid(const FeItem& it, float x, float y, float z) {
    const int ix = (int)floorf((x - it.org[0]) / it.dl), iy = (int)floorf((y - it.org[1]) / it.dl), iz = (int)floorf((z - it.org[2]) / it.dl);
    return (ix & ((1 << it.sx) - 1)) | (((iy & 7) | ((iz & 7) << 3)) << it.sx);
}

__global__ __launch_bounds__(BS) void fe_minmax_partial(FeTab t, const float* __restrict__ P, float* partial) {
    const int r = blockIdx.y;
    gs_minmax_partial_body(P + 3 * (size_t)t.off[r], t.off[r + 1] - t.off[r], partial + (size_t)r * PB * 6);
}

// (float)(1.0 / (double)k), the factor the reference multiplies a voxel's position sums with (cloud.h:120), k <= FE_CAP: once per state
__global__ void fe_rcp_init(float* rcp) { const int k = blockIdx.x * blockDim.x + threadIdx.x; if (k <= FE_CAP) rcp[k] = k ? (float)(1.0 / (double)k) : 0.f; }

// one workgroup per cloud: grid origin / dimensions exactly as the reference derives them, bucket geometry, per-call counters
__global__ __launch_bounds__(BS) void fe_params(FeTab t, const float* partial, float dl, GsParams* prm, FeGeom* geom, int* counters) {
    __shared__ float s_mm[(BS / 64) * 6];
    const int r = blockIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = threadIdx.x; i < PB; i += BS) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], partial[((size_t)r * PB + i) * 6 + d]); mx[d] = fmaxf(mx[d], partial[((size_t)r * PB + i) * 6 + 3 + d]); }
    }
    block_minmax3(mn, mx, s_mm);
    if (threadIdx.x == 0) {
        if (r == 0) { for (int i = 0; i < 8; ++i) counters[i] = 0; }          // [0] work items, [3] overflow rows
        GsParams p; FeGeom g;
        const float inv = 1 / dl;               // grid_subsampling.cpp:27-31
        float nd[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { g.org[d] = floorf(mn[d] * inv) * dl; p.org[d] = g.org[d]; nd[d] = floorf((mx[d] - g.org[d]) / dl) + 1.f; }
        g.dl = dl; p.dl = dl;
        bool ok = nd[0] >= 1.f && nd[1] >= 1.f && nd[2] >= 1.f && nd[0] < 65536.f && nd[1] < 1.0e6f && nd[2] < 1.0e6f;      // false for NaNs too (x: 16-bit row prefixes)
        // the smallest coordinate must land in voxel 0 (floor(min * inv) * dl can round above min: the reference's size_t index then wraps)
#pragma unroll
        for (int d = 0; d < 3; ++d) ok = ok && floorf((mn[d] - g.org[d]) / dl) >= 0.f;
        g.nx = ok ? (int)nd[0] : 1; g.ny = ok ? (int)nd[1] : 1; g.nz = ok ? (int)nd[2] : 1;
        p.nx = (unsigned long long)g.nx; p.ny = (unsigned long long)g.ny; p.m = 0; p.status = 0; p.key_and = 0ull; p.key_or = ~0ull;
        g.nby = (g.ny + 7) >> 3; g.nbz = (g.nz + 7) >> 3;
        g.sx = 3; g.nbx = (g.nx + 7) >> 3;
        if ((long long)g.nbx * g.nby * g.nbz > FE_NBMAX) { g.sx = 4; g.nbx = (g.nx + 15) >> 4; }
        if ((long long)g.nbx * g.nby * g.nbz > FE_NBMAX) ok = false;
        g.nb = ok ? g.nbx * g.nby * g.nbz : 0;
        g.ok = ok ? 1 : 0; g.n = t.off[r + 1] - t.off[r]; g.pad0 = g.pad1 = 0;
        if (!ok) p.status = 2;
        prm[r] = p; geom[r] = g;
    }
}

// [chunk][bucket] point counts of one cloud; chunk c = points [c FE_CH, (c + 1) FE_CH)
__global__ __launch_bounds__(FE_NT) void fe_count(FeTab t, const float* __restrict__ P, const FeGeom* __restrict__ geom, GsParams* prm, unsigned* cntm) {
    __shared__ unsigned s_h[FE_NBMAX];
    const int r = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const FeGeom g = geom[r];
    if (!g.ok || c * FE_CH >= g.n) return;
    for (int b = tid; b < g.nb; b += FE_NT) s_h[b] = 0u;
    __syncthreads();
    const float* Pr = P + 3 * (size_t)t.off[r];
    const int hi = min(g.n, (c + 1) * FE_CH);
    bool bad = false;
    for (int i = c * FE_CH + tid; i < hi; i += FE_NT) {
        int ix, iy, iz;
        if (fe_voxel(g, Pr[3 * (size_t)i], Pr[3 * (size_t)i + 1], Pr[3 * (size_t)i + 2], ix, iy, iz)) atomicAdd(&s_h[fe_bucket(g, ix, iy, iz)], 1u);
        else bad = true;
    }
    if (bad) atomicOr(&prm[r].status, 2);
    __syncthreads();
    unsigned* row = cntm + ((size_t)r * t.chunks_max + c) * FE_NBMAX;
    for (int b = tid; b < g.nb; b += FE_NT) row[b] = s_h[b];
}

// exclusive scan over the workgroup; returns the exclusive prefix and the total
__device__ __forceinline__ unsigned long long fe_block_scan(unsigned long long v, unsigned long long* s_w, int nt, unsigned long long& total) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long y = __shfl_up(incl, o); if (lane >= o) incl += y; }
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    unsigned long long wbase = 0, tot = 0;
    for (int w = 0; w < nt / 64; ++w) { const unsigned long long cw = s_w[w]; if (w < wid) wbase += cw; tot += cw; }
    __syncthreads();
    total = tot;
    return wbase + incl - v;
}

// one thread per bucket: cntm[c][b] becomes the number of points of bucket b in the chunks before c; the bucket's total
__global__ __launch_bounds__(BS) void fe_colscan(FeTab t, const FeGeom* __restrict__ geom, unsigned* cntm, unsigned* tot) {
    const int r = blockIdx.y, b = blockIdx.x * BS + threadIdx.x;
    const FeGeom g = geom[r];
    if (!g.ok || b >= g.nb) return;
    const int nchunk = (g.n + FE_CH - 1) / FE_CH;
    unsigned* cm = cntm + (size_t)r * t.chunks_max * FE_NBMAX + b;
    unsigned run = 0;
    for (int c0 = 0; c0 < nchunk; c0 += 8) {
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = c0 + u < nchunk ? cm[(size_t)(c0 + u) * FE_NBMAX] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (c0 + u < nchunk) { cm[(size_t)(c0 + u) * FE_NBMAX] = run; run += v[u]; }
    }
    tot[(size_t)r * FE_NBMAX + b] = run;
}

// one workgroup per cloud: boff[b] = where bucket b starts among the cloud's records; one work item per non-empty bucket
__global__ __launch_bounds__(FE_NT) void fe_bscan(FeTab t, const FeGeom* __restrict__ geom, const unsigned* __restrict__ tot, unsigned* boff, FeItem* items, int* counters) {
    __shared__ unsigned long long s_w[FE_NT / 64];
    __shared__ unsigned s_base;
    const int r = blockIdx.x, tid = threadIdx.x;
    const FeGeom g = geom[r];
    unsigned* bo = boff + (size_t)r * (FE_NBMAX + 1);
    if (!g.ok) { if (tid == 0) bo[0] = 0u; return; }
    unsigned long long carry = 0;          // records << 20 | non-empty buckets of the step
    for (int b0 = 0; b0 < g.nb; b0 += FE_NT) {
        const int b = b0 + tid;
        const unsigned n = b < g.nb ? tot[(size_t)r * FE_NBMAX + b] : 0u;
        unsigned long long total;
        const unsigned long long ex = fe_block_scan(((unsigned long long)n << 20) | (n ? 1ull : 0ull), s_w, FE_NT, total);
        const unsigned start = (unsigned)((carry + ex) >> 20);
        if (b < g.nb) bo[b] = start;
        const unsigned nne = (unsigned)(total & 0xfffffull);
        if (tid == 0) s_base = nne ? (unsigned)atomicAdd(&counters[0], (int)nne) : 0u;
        __syncthreads();
        if (n) {
            FeItem it; it.base = (unsigned)t.off[r] + start; it.n = n; it.rb = ((unsigned)r << 16) | (unsigned)b; it.sx = (unsigned)g.sx;
            it.org[0] = g.org[0]; it.org[1] = g.org[1]; it.org[2] = g.org[2]; it.dl = g.dl;
            items[s_base + (unsigned)(ex & 0xfffffull)] = it;
        }
        carry += total & ~0xfffffull;
        __syncthreads();
    }
    if (tid == 0) bo[g.nb] = (unsigned)(carry >> 20);
}

// second read of the points: packed records behind the LDS cursors of this chunk, FE_PPT points per thread and step (their loads issued
// together).  The kernel runs at the rate the memory system takes scattered 32-byte writes: what helps is more records per bucket and step (lines
// complete before the L2 drops them); issuing a step's loads ahead of the previous step's stores changed nothing.  FD / LD >= 0: row layout known at compile time (straight-line loads).
struct FePoint { float x, y, z; uint32_t w[4]; };
// the value of lane ^ 1: a DPP move (quad_perm [1,0,3,2]) instead of a trip through the LDS crossbar, which the cursors' atomics already use
__device__ __forceinline__ uint32_t fe_swap1(uint32_t v) {
#ifndef HIPEMU
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
#else
    return (uint32_t)__shfl_xor((int)v, 1);
#endif
}
template <int FD, int LD>
__device__ __forceinline__ FePoint fe_load_point(const float* __restrict__ Pr, const float* __restrict__ Fr, const int* __restrict__ Cr, int fdim, int ldim, int i) {
    FePoint pt;
    pt.x = Pr[3 * (size_t)i]; pt.y = Pr[3 * (size_t)i + 1]; pt.z = Pr[3 * (size_t)i + 2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t v = 0;
        if (FD >= 0) { if (q < FD) v = __float_as_uint(Fr[(size_t)i * FD + q]); else if (q - FD < LD) v = (uint32_t)Cr[(size_t)i * LD + (q - FD)]; }
        else { if (q < fdim) v = __float_as_uint(Fr[(size_t)i * fdim + q]); else if (q - fdim < ldim) v = (uint32_t)Cr[(size_t)i * ldim + (q - fdim)]; }
        pt.w[q] = v;
    }
    return pt;
}
template <int FD, int LD>
__global__ __launch_bounds__(FE_SNT) void fe_scatter(FeTab t, const float* __restrict__ P, const float* __restrict__ F, int fdim, const int* __restrict__ cls, int ldim,
                                                    const FeGeom* __restrict__ geom, const unsigned* __restrict__ cntm, const unsigned* __restrict__ boff, uint4* rec) {
    __shared__ unsigned s_cur[FE_NBMAX];
    const int r = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const FeGeom g = geom[r];
    if (!g.ok || c * FE_CH >= g.n) return;
    const unsigned* row = cntm + ((size_t)r * t.chunks_max + c) * FE_NBMAX;
    const unsigned* bo = boff + (size_t)r * (FE_NBMAX + 1);
    for (int b = tid; b < g.nb; b += FE_SNT) s_cur[b] = bo[b] + row[b];
    __syncthreads();
    const size_t o = (size_t)t.off[r];
    const float* Pr = P + 3 * o;
    const float* Fr = F ? F + o * fdim : nullptr;
    const int* Cr = cls ? cls + o * ldim : nullptr;
    uint4* R = rec + 2 * o;
    const int lo = c * FE_CH, hi = min(g.n, (c + 1) * FE_CH);
    for (int i0 = lo; i0 < hi; i0 += FE_SNT * FE_PPT) {
        FePoint cur[FE_PPT];
#pragma unroll
        for (int k = 0; k < FE_PPT; ++k) cur[k] = fe_load_point<FD, LD>(Pr, Fr, Cr, fdim, ldim, min(i0 + k * FE_SNT + tid, hi - 1));
#pragma unroll
        for (int k = 0; k < FE_PPT; ++k) {
            const int i = i0 + k * FE_SNT + tid;
            int ix, iy, iz;
            unsigned slot = 0xffffffffu;
            if (i < hi && fe_voxel(g, cur[k].x, cur[k].y, cur[k].z, ix, iy, iz)) slot = atomicAdd(&s_cur[fe_bucket(g, ix, iy, iz)], 1u);
            // (non-temporal stores here: 2.6x slower — the L2 merges the two halves of a record, and little else: a chunk adds ~4 records to a bucket)
#if FE_PAIR_STORES
            // the two 16-byte halves of a record leave from ADJACENT lanes, so a store instruction carries 32 runs of 32 bytes instead of 64 of 16: lanes
            // 2j and 2j + 1 swap a half each (the even lane hands over its second half and takes the odd lane's first)
            const bool odd = tid & 1;
            const uint32_t a0 = __float_as_uint(cur[k].x), a1 = __float_as_uint(cur[k].y), a2 = __float_as_uint(cur[k].z), a3 = (uint32_t)i;
            const uint32_t b0 = cur[k].w[0], b1 = cur[k].w[1], b2 = cur[k].w[2], b3 = cur[k].w[3];
            const uint32_t g0 = fe_swap1(odd ? a0 : b0), g1 = fe_swap1(odd ? a1 : b1), g2 = fe_swap1(odd ? a2 : b2), g3 = fe_swap1(odd ? a3 : b3);
            const unsigned pslot = fe_swap1(slot);
            const unsigned se = odd ? pslot : slot, so = odd ? slot : pslot;          // the even lane's record, the odd lane's record
            if (se != 0xffffffffu) R[2 * (size_t)se + (odd ? 1 : 0)] = odd ? make_uint4(g0, g1, g2, g3) : make_uint4(a0, a1, a2, a3);
            if (so != 0xffffffffu) R[2 * (size_t)so + (odd ? 1 : 0)] = odd ? make_uint4(b0, b1, b2, b3) : make_uint4(g0, g1, g2, g3);
#else
            if (slot != 0xffffffffu) {
                R[2 * (size_t)slot] = make_uint4(__float_as_uint(cur[k].x), __float_as_uint(cur[k].y), __float_as_uint(cur[k].z), (uint32_t)i);
                R[2 * (size_t)slot + 1] = make_uint4(cur[k].w[0], cur[k].w[1], cur[k].w[2], cur[k].w[3]);
            }
#endif
        }
    }
}

// ---- reduction: one workgroup per bucket ------------------------------------------------------------------------------------
struct FeRedArgs {
    FeTab t; const FeItem* items; GsParams* prm; int* counters; const float* rcp;
    uint4* rec; unsigned* nocc; unsigned* trow; unsigned char* lrc; int fdim, ldim;
};

// IMG: the bucket's (or slice's) records are also kept in LDS and the ordered walk reads them there — 51 KB per workgroup, three workgroups per CU.
// !IMG: the walk reads the records where the scatter left them (the workgroup has just streamed them: L2) and the rows go to the overflow region —
// 19 KB per workgroup, the occupancy is the registers'.
template <bool IMG>
struct FeRedLds {
    uint4 rec[IMG ? FE_CAP * 2 : 1];       // IMG: the records of the bucket (or slice), in arrival order
    unsigned midx[FE_CAP];                 // per voxel segment: the members' indices, in arrival order
    unsigned short perm[FE_CAP];           // per voxel segment: the members' positions in `rec` (IMG) / among the bucket's records (!IMG), by ascending index (input order)
    unsigned cnt[FE_VMAX + 1];             // histogram, then start of every voxel's segment
    union {
        unsigned fill[FE_VMAX];            // slices: members of a voxel seen so far (while a slice's records arrive)
        unsigned short slow[2 * FE_VMAX];  // row halves with a label column to settle from the members: row | half << 10 | columns << 11 (while the voxels are walked)
    };
    unsigned short ovid[FE_VMAX];          // the occupied voxels, ascending
    unsigned short orank[FE_VMAX + 2];     // occupied voxels in front of a voxel
    unsigned short slice[FE_SLICES + 2];
    unsigned long long sw[FE_RNT / 64];
    int nq, nslice, bad, nslow; unsigned ovf;
};

// A label column's vote from the members themselves (labels outside [0,13), more than 255 members): rare, kept out of line so that its
// tables do not weigh on the reduction's registers
__device__ __attribute__((noinline)) int fe_label_exact(const uint32_t* W, const unsigned short* perm, int word, int s, int e, int* status) {
    auto getl = [&](int j) { return (int)W[(size_t)perm[j] * 8 + word]; };
    int best = voxel_label_fast_t(getl, s, e);
    if (best < 0) best = voxel_label_t(getl, s, e, status);
    return best;
}

// Rows o_begin .. o_end - 1 of the bucket.  TWO lanes per occupied voxel walk its members in input order (L.perm), 16 bytes per member and
// lane: the even lane owns words 0..3 of the row (x, y, z and the voxel id in place of the record's index), the odd lane words 4..7 (features, labels).  Sums are taken
// sequentially in that order (what makes them the reference's, grid_subsampling.cpp:59-70); a label word is voted on with packed byte
// counters (labels 0..15; pk0: 0..7, pk1: 8..15) and the step at which a label is first seen kept the same way (fp0 / fp1): the
// reference's vote takes the first maximum in the iteration order of its unordered_map<int,int>, which for labels in [0,13) is "first
// seen last" (voxel_label.hpp).  Labels outside [0,13) or more than 255 members: noted in L.slow, settled by a second pass.
// s0 = start of the slice's first segment.  FD / LD >= 0: row layout known at compile time.
template <int FD, int LD, bool IMG>
__device__ __forceinline__ void fe_reduce_voxels(FeRedLds<IMG>& L, const FeRedArgs& a, int* status, int o_begin, int o_end, unsigned s0, uint4* rows, const uint4* __restrict__ Rg) {
    const int fdim = FD >= 0 ? FD : a.fdim, ldim = LD >= 0 ? LD : a.ldim;
    const int half = (int)threadIdx.x & 1;
    for (int o = o_begin + ((int)threadIdx.x >> 1); o < o_end; o += FE_RNT / 2) {
        const int v = L.ovid[o];
        const int s = (int)(L.cnt[v] - s0), e = (int)(L.cnt[v + 1] - s0), count = e - s;
        const float rc = a.rcp[count];                       // (float)(1.0 / (double)count): cloud.h:120 via grid_subsampling.cpp:87 (the load travels during the walk)
        float f[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned long long pk0[4] = {0ull, 0ull, 0ull, 0ull}, pk1[4] = {0ull, 0ull, 0ull, 0ull}, fp0[4] = {0ull, 0ull, 0ull, 0ull}, fp1[4] = {0ull, 0ull, 0ull, 0ull};
        bool big = false;
        for (int j = s; j < e; j += FE_WALK) {               // FE_WALK members' positions, then their halves, in flight together
            int p[FE_WALK]; uint4 w[FE_WALK];
#pragma unroll
            for (int h = 0; h < FE_WALK; ++h) p[h] = L.perm[min(j + h, e - 1)];
#pragma unroll
            for (int h = 0; h < FE_WALK; ++h) w[h] = IMG ? L.rec[2 * p[h] + half] : Rg[2 * (size_t)p[h] + half];
#pragma unroll
            for (int h = 0; h < FE_WALK; ++h) {
                if (j + h < e) {
                    const uint32_t ww[4] = {w[h].x, w[h].y, w[h].z, w[h].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // record word half * 4 + q: words 0..2 positions, word 3 the index (no sum); words 4.. the fdim features, then the ldim labels
                        const bool is_sum = half == 0 ? q < 3 : q < fdim, is_lab = half == 1 && q >= fdim && q < fdim + ldim;
                        if (is_sum) f[q] += __uint_as_float(ww[q]);
                        else if (is_lab) {
                            const unsigned Lb = ww[q], sh = (Lb & 7u) * 8u;
                            big |= Lb >= 13u;
                            const bool lo = Lb < 8u, hi = Lb >= 8u && Lb < 16u;
                            const unsigned long long cur = lo ? pk0[q] : pk1[q];
                            const unsigned long long first = ((cur >> sh) & 0xffull) ? 0ull : ((unsigned long long)((j + h - s) & 0xff) << sh);
                            const unsigned long long inc = 1ull << sh;
                            pk0[q] += lo ? inc : 0ull; pk1[q] += hi ? inc : 0ull;
                            fp0[q] |= lo ? first : 0ull; fp1[q] |= hi ? first : 0ull;
                        }
                    }
                }
            }
        }
        uint32_t out[4] = {0u, 0u, 0u, 0u};
        unsigned slow = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (half == 0) out[q] = q < 3 ? __float_as_uint(f[q] * rc) : (uint32_t)v;          // the row: (x, y, z, voxel id | features, labels)
            else if (q < fdim) out[q] = __float_as_uint(f[q] / (float)count);          // :90-94
            else if (q < fdim + ldim) {
                int best = 0, bestc = -1, bestf = -1;          // largest count; among equals the label first seen last
#pragma unroll
                for (int l = 0; l < 13; ++l) {
                    const int ck = (int)(((l < 8 ? pk0[q] : pk1[q]) >> ((l & 7) * 8)) & 0xffull);
                    const int fk = (int)(((l < 8 ? fp0[q] : fp1[q]) >> ((l & 7) * 8)) & 0xffull);
                    if (ck > bestc || (ck == bestc && fk > bestf)) { bestc = ck; bestf = fk; best = l; }
                }
                if (big || count > 255) slow |= 1u << q;          // labels outside [0,13) or byte counters too small: the exact routines
                out[q] = (uint32_t)best;
            }
        }
        if (slow) L.slow[atomicAdd(&L.nslow, 1)] = (unsigned short)((o - o_begin) | (half << 10) | (slow << 11));
        rows[2 * (size_t)o + half] = make_uint4(out[0], out[1], out[2], out[3]);
    }
    __syncthreads();
    if (L.nslow) {
        const uint32_t* W = IMG ? reinterpret_cast<const uint32_t*>(L.rec) : reinterpret_cast<const uint32_t*>(Rg);
        for (int t = (int)threadIdx.x; t < L.nslow; t += FE_RNT) {
            const int o = o_begin + (L.slow[t] & 1023), hf = (L.slow[t] >> 10) & 1;
            const unsigned mask = L.slow[t] >> 11;
            const int v = L.ovid[o];
            const int s = (int)(L.cnt[v] - s0), e = (int)(L.cnt[v + 1] - s0);
            for (int q = 0; q < 4; ++q) if ((mask >> q) & 1u)
                reinterpret_cast<uint32_t*>(rows)[(size_t)o * 8 + hf * 4 + q] = (uint32_t)fe_label_exact(W, L.perm, hf * 4 + q, s, e, status);
        }
    }
}

// segment starts, the occupied voxels and their ranks from the histogram in L.cnt (V = 512 or 1024 voxels)
template <bool IMG>
__device__ __forceinline__ void fe_segments(FeRedLds<IMG>& L, int V) {
    constexpr int EM = FE_VMAX / FE_RNT;
    const int tid = threadIdx.x, E = V / FE_RNT;           // 2 or 4 voxels per thread (256 threads)
    unsigned c4[EM]; unsigned tot = 0, occ = 0;
#pragma unroll
    for (int k = 0; k < EM; ++k) { c4[k] = k < E ? L.cnt[tid * E + k] : 0u; tot += c4[k]; occ += c4[k] ? 1u : 0u; if (c4[k] > (unsigned)FE_CAP) L.bad = 1; }
    unsigned long long total;
    const unsigned long long ex = fe_block_scan(((unsigned long long)tot << 20) | occ, L.sw, FE_RNT, total);
    unsigned st = (unsigned)(ex >> 20), orr = (unsigned)(ex & 0xfffffull);
#pragma unroll
    for (int k = 0; k < EM; ++k) if (k < E) {
        const int v = tid * E + k;
        L.cnt[v] = st; L.orank[v] = (unsigned short)orr;
        if (c4[k]) { L.ovid[orr] = (unsigned short)v; ++orr; }
        st += c4[k];
    }
    if (tid == FE_RNT - 1) { L.cnt[V] = st; L.orank[V] = (unsigned short)orr; }
    __syncthreads();
}

#ifdef SSDR_FE_STAMPS
#define FE_STAMP(k) do { if (tid == 0) { const long long t_ = clock64(); t_acc[k] += (unsigned long long)(t_ - t_last); t_last = t_; } } while (0)
#else
#define FE_STAMP(k) do { } while (0)
#endif
// One workgroup per bucket (work items dealt round robin).  A bucket of at most FE_CAP records is read once: its records go to LDS as they
// arrive, with the voxel histogram's returning atomics handing every record its arrival slot inside its voxel.  A larger bucket is cut into
// slices of consecutive voxels (at most FE_CAP records each) after a histogram pass, and read again once per slice.
#ifndef FE_RED_WAVES
#define FE_RED_WAVES 4          // waves per SIMD asked of the no-image variant (A/B builds: -DFE_RED_WAVES=5 / 6 spill 24 / 36 dwords)
#endif
template <int FD, int LD, bool IMG>
__global__ __launch_bounds__(FE_RNT) SSDR_WAVES_PER_EU(IMG ? 2 : FE_RED_WAVES) void fe_reduce(FeRedArgs a) {
    __shared__ FeRedLds<IMG> L;
    const int tid = threadIdx.x;
#ifdef SSDR_FE_STAMPS
    long long t_last = clock64(); unsigned long long t_acc[5] = {0, 0, 0, 0, 0};
#endif
    const int nitems = a.counters[0];
    for (int i = blockIdx.x; i < nitems; i += (int)gridDim.x) {
        const FeItem it = a.items[i];
        const int r = (int)(it.rb >> 16), b = (int)(it.rb & 0xffffu);
        GsParams* prm = a.prm + r;
        const int n = (int)it.n, V = FE_LR << it.sx;
        const bool fast = n <= FE_CAP;
        const uint4* R = a.rec + 2 * (size_t)it.base;
        for (int v = tid; v <= V; v += FE_RNT) L.cnt[v] = 0;
        if (tid == 0) { L.bad = 0; L.nq = 0; L.nslow = 0; }
        __syncthreads();
        // histogram over the bucket's voxels; FE_RPT records per thread, their loads issued together.  The fast path keeps every record's
        // voxel and arrival slot inside the voxel (what the returning atomic hands out) in registers until the segments are known
        int vv[FE_RPT]; unsigned uu[FE_RPT], ix[FE_RPT];
#pragma unroll
        for (int k = 0; k < FE_RPT; ++k) { vv[k] = 0; uu[k] = 0u; ix[k] = 0u; }
        if (fast) {
            // (x, y, z, index) is the FIRST half of a record: without the LDS image this pass reads 16 of its 32 bytes
            uint4 r0[FE_RPT], r1[IMG ? FE_RPT : 1];
#pragma unroll
            for (int k = 0; k < FE_RPT; ++k) { const size_t p = (size_t)min(tid + k * FE_RNT, n - 1); r0[k] = R[2 * p]; if (IMG) r1[k] = R[2 * p + 1]; }
#pragma unroll
            for (int k = 0; k < FE_RPT; ++k) {
                const int p = tid + k * FE_RNT;
                vv[k] = fe_vid(it, __uint_as_float(r0[k].x), __uint_as_float(r0[k].y), __uint_as_float(r0[k].z));
                ix[k] = r0[k].w;
                if (p < n) { uu[k] = atomicAdd(&L.cnt[vv[k]], 1u); if (IMG) { L.rec[2 * p] = r0[k]; L.rec[2 * p + 1] = r1[IMG ? k : 0]; } }
            }
        } else {
            for (int p0 = 0; p0 < n; p0 += FE_CAP) {
                uint4 r0[FE_RPT];
#pragma unroll
                for (int k = 0; k < FE_RPT; ++k) r0[k] = R[2 * (size_t)min(p0 + tid + k * FE_RNT, n - 1)];
#pragma unroll
                for (int k = 0; k < FE_RPT; ++k)
                    if (p0 + tid + k * FE_RNT < n) atomicAdd(&L.cnt[fe_vid(it, __uint_as_float(r0[k].x), __uint_as_float(r0[k].y), __uint_as_float(r0[k].z))], 1u);
            }
        }
        __syncthreads();
        FE_STAMP(0);
        fe_segments(L, V);
        FE_STAMP(1);
        const int nocc = L.orank[V];
        uint4* rows = a.rec + 2 * (size_t)it.base;           // the bucket's rows go over its own records (all of them are in LDS by now) ...
        unsigned sg[FE_RPT], eg[FE_RPT];
#pragma unroll
        for (int k = 0; k < FE_RPT; ++k) { sg[k] = 0u; eg[k] = 0u; }
        if (fast) {          // every voxel's member indices, in arrival order
#pragma unroll
            for (int k = 0; k < FE_RPT; ++k) {
                sg[k] = L.cnt[vv[k]]; eg[k] = L.cnt[vv[k] + 1];
                if (tid + k * FE_RNT < n) L.midx[sg[k] + uu[k]] = ix[k];
            }
        }
        if (tid == 0) {
            if (fast) { L.nslice = 1; L.slice[0] = 0; L.slice[1] = (unsigned short)V; }
            if (!fast || !IMG) {
                // ... unless the bucket is cut into slices that are read again (or the walk reads the records in place): its rows then go to the
                // overflow region behind the records
                const unsigned at = (unsigned)atomicAdd(&a.counters[3], nocc);
                L.ovf = at + (unsigned)nocc <= (unsigned)a.t.ovf_cap ? at : 0xffffffffu;
                if (L.ovf == 0xffffffffu) L.bad = 1;
            }
            if (!fast) {
                int ns = 0, v = 0; L.slice[0] = 0;
                while (v < V && ns < FE_SLICES) {
                    const unsigned s0 = L.cnt[v]; int w = v;
                    while (w < V && L.cnt[w + 1] - s0 <= (unsigned)FE_CAP) ++w;
                    if (w == v) { L.bad = 1; break; }
                    v = w; L.slice[++ns] = (unsigned short)v;
                }
                if (v < V) L.bad = 1;
                L.nslice = ns;
            }
        }
        __syncthreads();
        if (fast) {          // every member's rank by index inside its voxel = its place in the order the reference adds the members in
#pragma unroll
            for (int k = 0; k < FE_RPT; ++k) {
                if (tid + k * FE_RNT < n) {
                    unsigned rank = 0;
                    for (unsigned j0 = sg[k]; j0 < eg[k]; j0 += 8) {          // eight list entries in flight
                        unsigned m8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) m8[u] = L.midx[min(j0 + u, eg[k] - 1)];
#pragma unroll
                        for (int u = 0; u < 8; ++u) rank += (j0 + u < eg[k] && m8[u] < ix[k]) ? 1u : 0u;
                    }
                    L.perm[sg[k] + rank] = (unsigned short)(tid + k * FE_RNT);
                }
            }
            __syncthreads();
        }
        if ((!fast || !IMG) && !L.bad) rows = a.rec + 2 * ((size_t)a.t.n_total + L.ovf);
        const size_t tb = (size_t)r * FE_NBMAX + b;
        if (L.bad) {           // a voxel of more than FE_CAP points (or more slices / overflow rows than there is room for): not this entry point's case
            if (tid == 0) { atomicOr(&prm->status, 4); a.nocc[tb] = 0u; a.trow[tb] = 0u; }
            for (int l = tid; l < FE_LR; l += FE_RNT) a.lrc[tb * FE_LR + l] = 0;
        } else {
            for (int si = 0; si < L.nslice; ++si) {
                const int v0 = L.slice[si], v1 = L.slice[si + 1];
                const unsigned s0 = L.cnt[v0];
                const int ns = (int)(L.cnt[v1] - s0);
                if (ns > 0 && !fast) {          // the slice's records, read again; then L.rec / L.perm as the fast path leaves them
                    __syncthreads();
                    if (tid == 0) L.nq = 0;
                    for (int v = v0 + tid; v < v1; v += FE_RNT) L.fill[v] = 0u;
                    __syncthreads();
                    for (int p0 = 0; p0 < n; p0 += FE_CAP) {
                        uint4 q0[FE_RPT], q1[IMG ? FE_RPT : 1];
#pragma unroll
                        for (int k = 0; k < FE_RPT; ++k) { const size_t p = (size_t)min(p0 + tid + k * FE_RNT, n - 1); q0[k] = R[2 * p]; if (IMG) q1[k] = R[2 * p + 1]; }
#pragma unroll
                        for (int k = 0; k < FE_RPT; ++k) {
                            const int v = fe_vid(it, __uint_as_float(q0[k].x), __uint_as_float(q0[k].y), __uint_as_float(q0[k].z));
                            if (p0 + tid + k * FE_RNT < n && v >= v0 && v < v1) {
                                int q;
                                if (IMG) { q = atomicAdd(&L.nq, 1); L.rec[2 * q] = q0[k]; L.rec[2 * q + 1] = q1[IMG ? k : 0]; }
                                else q = p0 + tid + k * FE_RNT;          // the record's place among the bucket's own (<= FE_SLICES * FE_CAP < 65536)
                                // (index, position) pairs of a voxel, in arrival order: midx holds the indices, perm the positions until the ranks are known
                                const unsigned at = L.cnt[v] - s0 + atomicAdd(&L.fill[v], 1u);
                                L.midx[at] = q0[k].w; L.perm[at] = (unsigned short)q;
                            }
                        }
                    }
                    __syncthreads();
                    // ranks inside every voxel: one thread per voxel orders its (short) list by insertion
                    for (int v = v0 + tid; v < v1; v += FE_RNT) {
                        const int sb = (int)(L.cnt[v] - s0), eb = (int)(L.cnt[v + 1] - s0);
                        for (int x = sb + 1; x < eb; ++x) {
                            const unsigned ki = L.midx[x]; const unsigned short pi = L.perm[x];
                            int y = x - 1;
                            while (y >= sb && L.midx[y] > ki) { L.midx[y + 1] = L.midx[y]; L.perm[y + 1] = L.perm[y]; --y; }
                            L.midx[y + 1] = ki; L.perm[y + 1] = pi;
                        }
                    }
                    __syncthreads();
                    if (tid == 0) L.nslow = 0;      // (L.slow shares its storage with L.fill)
                    __syncthreads();
                }
                FE_STAMP(2);
                if (ns > 0) fe_reduce_voxels<FD, LD, IMG>(L, a, &prm->status, L.orank[v0], L.orank[v1], s0, rows, R);
                FE_STAMP(3);
            }
            // what the move needs: number of rows, where they are, occupied voxels per local row (y, z)
            if (tid == 0) { a.nocc[tb] = (unsigned)nocc; a.trow[tb] = (fast && IMG) ? it.base : 0x80000000u | L.ovf; }
            for (int l = tid; l < FE_LR; l += FE_RNT) a.lrc[tb * FE_LR + l] = (unsigned char)(L.orank[(l + 1) << it.sx] - L.orank[l << it.sx]);
        }
        __syncthreads();
        FE_STAMP(4);
    }
#ifdef SSDR_FE_STAMPS
    if (tid == 0) for (int k = 0; k < 5; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(a.counters) + 8 + k, t_acc[k]);
#endif
}

// occupied voxels of every padded row (iy, iz) of a cloud, and in front of every bucket inside its row (keys grow along x first)
__global__ __launch_bounds__(BS) void fe_rowpre(const FeGeom* __restrict__ geom, const unsigned* __restrict__ boff, const unsigned char* __restrict__ lrc,
                                                unsigned short* pre, unsigned* rowcnt) {
    const int r = blockIdx.y;
    const FeGeom g = geom[r];
    if (!g.ok) return;
    const int ry = g.nby * 8, prows = ry * g.nbz * 8;
    const unsigned* bo = boff + (size_t)r * (FE_NBMAX + 1);
    for (int pr = blockIdx.x * BS + threadIdx.x; pr < prows; pr += gridDim.x * BS) {
        const int iy = pr % ry, iz = pr / ry;
        const int l = (iy & 7) | ((iz & 7) << 3);
        const int b0 = g.nbx * ((iy >> 3) + g.nby * (iz >> 3));
        unsigned run = 0;
        for (int bx = 0; bx < g.nbx; ++bx) {
            const int b = b0 + bx;
            if (bo[b + 1] != bo[b]) {
                const size_t at = ((size_t)r * FE_NBMAX + b) * FE_LR + l;
                pre[at] = (unsigned short)run;
                run += lrc[at];
            }
        }
        rowcnt[(size_t)r * FE_ROWCAP + pr] = run;
    }
}

// one workgroup per cloud: rows in front of every padded row; the cloud's voxel count
__global__ __launch_bounds__(FE_NT) void fe_rowscan(const FeGeom* __restrict__ geom, unsigned* rowcnt, GsParams* prm, long long* out_m) {
    __shared__ unsigned long long s_w[FE_NT / 64];
    const int r = blockIdx.x, tid = threadIdx.x;
    const FeGeom g = geom[r];
    if (!g.ok) { if (tid == 0) { prm[r].m = 0; if (out_m) out_m[r] = 0; } return; }
    const int prows = g.nby * 8 * g.nbz * 8;
    unsigned* rc = rowcnt + (size_t)r * FE_ROWCAP;
    unsigned long long carry = 0;
    for (int p0 = 0; p0 < prows; p0 += FE_NT) {
        const int p = p0 + tid;
        const unsigned v = p < prows ? rc[p] : 0u;
        unsigned long long total;
        const unsigned long long ex = fe_block_scan((unsigned long long)v, s_w, FE_NT, total);
        if (p < prows) rc[p] = (unsigned)(carry + ex);
        carry += total;
    }
    if (tid == 0) { prm[r].m = (int)carry; if (out_m) out_m[r] = (long long)carry; }
}

// one wave per bucket: its rows to their final places
struct FeMoveArgs {
    FeTab t; const FeGeom* geom; const FeItem* items; const int* counters; const uint4* rec; const unsigned* nocc; const unsigned* trow;
    const unsigned char* lrc; const unsigned short* pre; const unsigned* rowbase; int fdim, ldim; float* out_p; float* out_f; int* out_c;
};
// FD / LD >= 0: row layout known at compile time (the hot path's 3 features + 1 label: one 12-byte store each for the position and the features instead of
// seven guarded 4-byte stores)
template <int FD, int LD>
__global__ __launch_bounds__(BS) void fe_move(FeMoveArgs a) {
    const int fdim = FD >= 0 ? FD : a.fdim, ldim = LD >= 0 ? LD : a.ldim;
    const int lane = threadIdx.x & 63;
    const int nw = a.counters[0], nwaves = (int)gridDim.x * (BS / 64);
    for (int tk = (int)blockIdx.x * (BS / 64) + (int)(threadIdx.x >> 6); tk < nw; tk += nwaves) {
        const unsigned we = a.items[tk].rb;
        const int r = (int)(we >> 16), b = (int)(we & 0xffffu);
        const FeGeom g = a.geom[r];
        const size_t tb = (size_t)r * FE_NBMAX + b;
        const int nocc = (int)a.nocc[tb];
        const unsigned tr = a.trow[tb];
        const uint4* rows = a.rec + 2 * ((tr & 0x80000000u) ? (size_t)a.t.n_total + (tr & 0x7fffffffu) : (size_t)tr);
        const int by = (b / g.nbx) % g.nby, bz = b / (g.nbx * g.nby);
        // occupied voxels of the bucket in front of every local row
        const unsigned mine = a.lrc[tb * FE_LR + lane];
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(incl, o); if (lane >= o) incl += y; }
        const unsigned lpre = incl - mine;
        const unsigned short* pre = a.pre + tb * FE_LR;
        const unsigned* rb = a.rowbase + (size_t)r * FE_ROWCAP;
        const size_t o = (size_t)a.t.off[r];
        // a lane PAIR takes a row: the even lane its first half (x, y, z, voxel) and the position, the odd lane the second (features, labels) — a load
        // instruction then reads 32 whole rows as one contiguous kilobyte instead of 64 half rows at a stride of 32 bytes
        const int half = lane & 1, pl = lane >> 1;
        for (int j0 = 0; j0 < nocc; j0 += 64) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = j0 + 32 * h + pl;
                const bool on = j < nocc;
                uint4 rr = make_uint4(0, 0, 0, 0);
                if (on) rr = rows[2 * (size_t)j + half];
                const unsigned other = fe_swap1(rr.w);                      // (unconditional: a DPP move reads its partner lane only while that lane is active)
                const unsigned vw = half ? other : rr.w;                    // the voxel word lives in the first half
                const int l = on ? (int)(vw >> g.sx) : 0;
                const unsigned lp = __shfl(lpre, l);
                if (on) {
                    const int iy = by * 8 + (l & 7), iz = bz * 8 + (l >> 3);
                    const size_t fin = o + rb[iy + g.nby * 8 * iz] + pre[l] + ((unsigned)j - lp);
                    if (!half) { a.out_p[3 * fin] = __uint_as_float(rr.x); a.out_p[3 * fin + 1] = __uint_as_float(rr.y); a.out_p[3 * fin + 2] = __uint_as_float(rr.z); }
                    else {
                        const uint32_t w[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (k < fdim) a.out_f[fin * fdim + k] = __uint_as_float(w[k]);
                            else if (k - fdim < ldim) a.out_c[fin * ldim + (k - fdim)] = (int)w[k];
                        }
                    }
                }
            }
        }
    }
}

struct FeState { DevBuf partial, geom, cntm, tot, boff, items, counters, rec, nocc, trow, lrc, pre, rowcnt, rcp; bool rcp_ready = false; };

}  // namespace

bool frontend_fits(size_t fdim, size_t ldim) { return 3 + fdim + ldim <= 7; }

// All clouds of a batch in the same launches; prm: the caller's per-cloud GsParams (status / voxel count as the sort-based path leaves them).
int frontend_batch_device(const float* d_p, const float* d_f, size_t fdim, const int32_t* d_c, size_t ldim, const int64_t* room_off, size_t nr, float dl,
                          float* d_op, float* d_of, int32_t* d_oc, int64_t* d_om, GsParams* prm, hipStream_t s) {
    FeState& S = per_stream<FeState>(s);
    FeTab t; t.nr = (int)nr;
    int maxn = 0;
    for (size_t r = 0; r < nr; ++r) {
        const long n = (long)(room_off[r + 1] - room_off[r]);
        if (n <= 0 || room_off[r + 1] > 0x3fffffff) { set_error("grid_subsample_batch: every cloud needs >= 1 point"); return n <= 0 ? SSDR_ERR_EMPTY : SSDR_ERR_INVALID; }
        t.off[r] = (int)room_off[r]; maxn = std::max(maxn, (int)n);
    }
    t.off[nr] = (int)room_off[nr];
    t.n_total = (int)room_off[nr];
    t.chunks_max = (maxn + FE_CH - 1) / FE_CH;
    // The reduction reads a bucket's records in place (no LDS image: 19 KB instead of 51 KB per workgroup, four workgroups per CU by registers) and every
    // bucket's rows go to the overflow region, which then holds up to one row per point.  Measured on one box (tools/gpu_fe_ab.sh, round 5): fe_reduce 0.51-0.56
    // -> 0.45-0.46 ms, front end 1.23-1.26 -> 1.16-1.18 ms, headline 178.2-178.7 -> 180.9-181.8 Mpoints/s; 5 / 6 waves per SIMD (spilling 24 / 36 dwords)
    // 0.45-0.51 ms.  SSDR_FE_IMAGE=1 keeps the LDS-image kernel (A/B runs).
    static const bool fe_image = [] { const char* e = getenv("SSDR_FE_IMAGE"); return e && e[0] == '1'; }();
    static const int fe_wgs = [] { const char* e = getenv("SSDR_FE_WGS"); return e ? atoi(e) : 0; }();
    static const int fe_pad = [] { const char* e = getenv("SSDR_FE_PADLDS"); return e ? atoi(e) : 0; }();      // (development: unused dynamic LDS caps the workgroups per CU)
    t.ovf_cap = fe_image ? std::max(65536, t.n_total / 8) : t.n_total;
    const unsigned R = (unsigned)nr;
    SSDR_TRY(S.partial.reserve(24 * (size_t)PB * nr)); SSDR_TRY(S.geom.reserve(sizeof(FeGeom) * nr)); SSDR_TRY(S.counters.reserve(256));
    SSDR_TRY(S.cntm.reserve(4 * (size_t)FE_NBMAX * t.chunks_max * nr)); SSDR_TRY(S.tot.reserve(4 * (size_t)FE_NBMAX * nr)); SSDR_TRY(S.boff.reserve(4 * (size_t)(FE_NBMAX + 1) * nr));
    SSDR_TRY(S.items.reserve(sizeof(FeItem) * (size_t)FE_NBMAX * nr)); SSDR_TRY(S.nocc.reserve(4 * (size_t)FE_NBMAX * nr)); SSDR_TRY(S.trow.reserve(4 * (size_t)FE_NBMAX * nr));
    SSDR_TRY(S.lrc.reserve((size_t)FE_NBMAX * FE_LR * nr)); SSDR_TRY(S.pre.reserve(2 * (size_t)FE_NBMAX * FE_LR * nr));
    SSDR_TRY(S.rowcnt.reserve(4 * (size_t)FE_ROWCAP * nr));
    SSDR_TRY(S.rec.reserve(32 * ((size_t)t.n_total + t.ovf_cap) + 64));
    if (!S.rcp_ready) {
        SSDR_TRY(S.rcp.reserve(4 * (FE_CAP + 1)));
        hipLaunchKernelGGL(fe_rcp_init, dim3((FE_CAP + 256) / 256), dim3(256), 0, s, S.rcp.as<float>());
        S.rcp_ready = true;
    }
    FeGeom* geom = S.geom.as<FeGeom>(); int* counters = S.counters.as<int>(); FeItem* items = S.items.as<FeItem>();
    // profiler sites (ssdr_prof_enable; bench.py's roofline leg): `work` = the algorithmic bytes a site is charged with — the one read of the inputs
    // (28 B per point) goes to the scatter, the one write of the rows (28 B per voxel: the count is the device's, bench.py adds it) to the reduction
    {
    ProfScope prof("fe_bbox_count_scan", s, 0.0);
    hipLaunchKernelGGL(fe_minmax_partial, dim3(PB, R), dim3(BS), 0, s, t, d_p, S.partial.as<float>());
    hipLaunchKernelGGL(fe_params, dim3(R), dim3(BS), 0, s, t, S.partial.as<float>(), dl, prm, geom, counters);
    hipLaunchKernelGGL(fe_count, dim3(t.chunks_max, R), dim3(FE_NT), 0, s, t, d_p, geom, prm, S.cntm.as<unsigned>());
    hipLaunchKernelGGL(fe_colscan, dim3(FE_NBMAX / BS, R), dim3(BS), 0, s, t, geom, S.cntm.as<unsigned>(), S.tot.as<unsigned>());
    hipLaunchKernelGGL(fe_bscan, dim3(R), dim3(FE_NT), 0, s, t, geom, S.tot.as<unsigned>(), S.boff.as<unsigned>(), items, counters);
    }
    {
    ProfScope prof("fe_scatter", s, (double)(12 + 4 * (fdim + ldim)) * (double)t.n_total);
    if (fdim == 3 && ldim == 1)
        hipLaunchKernelGGL((fe_scatter<3, 1>), dim3(t.chunks_max, R), dim3(FE_SNT), 0, s, t, d_p, d_f, (int)fdim, (const int*)d_c, (int)ldim, geom, S.cntm.as<unsigned>(), S.boff.as<unsigned>(),
                           S.rec.as<uint4>());
    else
        hipLaunchKernelGGL((fe_scatter<-1, -1>), dim3(t.chunks_max, R), dim3(FE_SNT), 0, s, t, d_p, d_f, (int)fdim, (const int*)d_c, (int)ldim, geom, S.cntm.as<unsigned>(), S.boff.as<unsigned>(),
                           S.rec.as<uint4>());
    }
    FeRedArgs ra; ra.t = t; ra.items = items; ra.prm = prm; ra.counters = counters; ra.rcp = S.rcp.as<float>(); ra.rec = S.rec.as<uint4>();
    ra.nocc = S.nocc.as<unsigned>(); ra.trow = S.trow.as<unsigned>(); ra.lrc = S.lrc.as<unsigned char>(); ra.fdim = (int)fdim; ra.ldim = (int)ldim;
    {
    ProfScope prof("fe_reduce", s, 0.0);
    if (!fe_image) {
        // The items (buckets) differ in size and are dealt to the workgroups round robin: with twice the resident workgroups (16 per CU: rounds 4-5) the
        // kernel ended with its unluckiest workgroup; with ~ one workgroup per item the hardware's dispatcher does the balancing.  Same box, workgroups
        // per CU (SSDR_FE_WGS): 16: 0.43 ms, 32: 0.39, 64: 0.35, **128: 0.338**, 256: 0.337, 512: 0.343, 4096: slower than 16 (front end 1.08 -> 1.00 ms,
        // 198 -> 201 Mpoints/s).  Capped by the points (a bucket holds ~10^3 of them; workgroups beyond the item count exit at once).
        const long cap = std::max<long>((long)ctx().num_cu * 16, (long)t.n_total / 64);
        const int g = (int)std::min<long>((long)ctx().num_cu * (fe_wgs > 0 ? fe_wgs : 128), fe_wgs > 0 ? (1L << 30) : cap);
        if (fdim == 3 && ldim == 1) hipLaunchKernelGGL((fe_reduce<3, 1, false>), dim3(g), dim3(FE_RNT), (size_t)fe_pad, s, ra);
        else hipLaunchKernelGGL((fe_reduce<-1, -1, false>), dim3(g), dim3(FE_RNT), (size_t)fe_pad, s, ra);
    } else if (fdim == 3 && ldim == 1) hipLaunchKernelGGL((fe_reduce<3, 1, true>), dim3(ctx().num_cu * (fe_wgs > 0 ? fe_wgs : 3)), dim3(FE_RNT), 0, s, ra);        // the hot path's rows
    else hipLaunchKernelGGL((fe_reduce<-1, -1, true>), dim3(ctx().num_cu * (fe_wgs > 0 ? fe_wgs : 3)), dim3(FE_RNT), 0, s, ra);
    }
    ProfScope prof_move("fe_rows_move", s, 0.0);
    hipLaunchKernelGGL(fe_rowpre, dim3(256, R), dim3(BS), 0, s, geom, S.boff.as<unsigned>(), S.lrc.as<unsigned char>(), S.pre.as<unsigned short>(), S.rowcnt.as<unsigned>());
    hipLaunchKernelGGL(fe_rowscan, dim3(R), dim3(FE_NT), 0, s, geom, S.rowcnt.as<unsigned>(), prm, (long long*)d_om);
    FeMoveArgs ma; ma.t = t; ma.geom = geom; ma.items = items; ma.counters = counters; ma.rec = S.rec.as<uint4>(); ma.nocc = S.nocc.as<unsigned>();
    ma.trow = S.trow.as<unsigned>(); ma.lrc = S.lrc.as<unsigned char>(); ma.pre = S.pre.as<unsigned short>(); ma.rowbase = S.rowcnt.as<unsigned>(); ma.fdim = (int)fdim; ma.ldim = (int)ldim;
    ma.out_p = d_op; ma.out_f = d_of; ma.out_c = d_oc;
    // a wave per item, round robin: 8 workgroups per CU give every wave two items of different sizes, 32 one (0.100 -> 0.094 ms; SSDR_FE_MOVE_WGS)
    static const int mv_wgs = [] { const char* e = getenv("SSDR_FE_MOVE_WGS"); return e ? atoi(e) : 32; }();
    if (fdim == 3 && ldim == 1) hipLaunchKernelGGL((fe_move<3, 1>), dim3(ctx().num_cu * mv_wgs), dim3(BS), 0, s, ma);
    else hipLaunchKernelGGL((fe_move<-1, -1>), dim3(ctx().num_cu * mv_wgs), dim3(BS), 0, s, ma);
    SSDR_HIP(hipGetLastError());
#ifdef SSDR_FE_STAMPS
    {
        SSDR_HIP(hipStreamSynchronize(s));
        unsigned long long h[8]; int hc[8];
        SSDR_HIP(hipMemcpy(hc, counters, sizeof(hc), hipMemcpyDeviceToHost));
        SSDR_HIP(hipMemcpy(h, reinterpret_cast<unsigned long long*>(counters) + 8, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "fe_reduce: %d items, %d overflow rows; clock ticks per phase (sum over workgroups): load+hist %llu, segments %llu, rank %llu, voxels %llu, tail %llu\n",
                hc[0], hc[3], h[0], h[1], h[2], h[3], h[4]);
        SSDR_HIP(hipMemset(reinterpret_cast<unsigned long long*>(counters) + 8, 0, sizeof(h)));
    }
#endif
    return SSDR_OK;
}

}  // namespace ssdr
