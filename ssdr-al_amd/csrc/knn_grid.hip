// Uniform-grid K-nearest-neighbour search with an exact tie-aware hand-over to the kd-tree walk (gfx950).
//
// The reference op (S3/utils/nearest_neighbors/knn_.cxx:22-135 over nanoflann.hpp) returns, per query, the K support
// indices by ascending fp32 distance ((dx*dx + dy*dy) + dz*dz, no FMA); among equal distances the order is the order in
// which nanoflann's tree walk met them (KNNResultSet::addPoint :63-92, searchLevel :1271-1329).  Whenever the K + 1 smallest
// distances of a query are pairwise distinct that answer does not depend on the tree at all, so:
//
//   fast path  points are binned into a dense uniform grid per support set (cell size measured per set: the radius that
//              holds ~20 points around sampled points), cell-sorted 16-byte records (x, y, z, index); one lane per query, in
//              cell order, scans the 3 x 3 rows of cells around its own cell (x is the fastest cell dimension: each row is one
//              contiguous range), keeps the K + 1 best in registers, and accepts when the (K+1)-th distance lies inside the
//              radius the scanned block guarantees; otherwise the block grows (5^3, 7^3).
//   hand-over  a row with two equal distances among its K + 1 best, with the K-th and (K+1)-th closer than 2^-19 relative
//              (a tree bound rounded the other way could have pruned one of them), with fewer than K + 1 support points, or
//              not settled inside the 7^3 block goes to a work list and is answered by kd_walk on the exact nanoflann
//              tree (kdtree.hip); only support sets that own such rows get their tree built (device-side flags).
//
// Same arithmetic as kdtree.hip: this TU is compiled with -ffp-contract=off.
#include "ssdr_internal.hpp"
#include "block_prims.hpp"
#include "knn_regset.hpp"
#include <cfloat>
#include <cstring>

namespace ssdr {

namespace {

constexpr int GS_SAMPLES = 8;            // sampled points per set for the cell-size measurement
constexpr int GS_BINS = 192;             // log-scale histogram of squared distances, four bins per octave
constexpr int GS_BIN0 = (87 << 2);       // first bin: d^2 = 2^-40
constexpr int GS_RMAX = 3;               // largest block: (2*3+1)^3 cells
constexpr int SCAN_BLK = 4096;           // cells per scan block
// A row whose K-th and (K+1)-th distances are closer than this goes to the tree walk.  nanoflann prunes a far branch when its bound
// m = fl(fl(m + cut) - dists[idx]) exceeds worstDist (searchLevel :1316-1319).  The per-dimension terms are rounded exactly like a point's
// ((a - b)^2, monotone), so with exact sums the bound never exceeds a point's distance; but m is an incrementally rounded sum: every far
// transition of the path adds at most 3 ulps of its value, the point's own two additions 2 more.  With at most MAX_LEVELS = 40 transitions
// (deeper trees are refused) the bound can exceed the computed distance of a point below it by < (3 * 40 + 6) * 2^-24 < 2^-17 relative:
// only if the reference's K-th and the dropped point are that close can the reference's answer differ from the K smallest distances.
constexpr float NEAR_TIE = 1.0000076293945312f;      // 1 + 2^-17
constexpr int RETRY_FROM_R1 = 1 << 30;   // tag on a retry entry's job id: the 3^3 block of that job's grid has not been searched yet

__device__ __forceinline__ int cell_of(float v, float o, float inv_c, int n) {
    const float f = (v - o) * inv_c;
    int c = f > 0.f ? (int)fminf(f, 1.0e9f) : 0;
    return c < n ? c : n - 1;
}

// bounding box, cell size, grid dimensions; zeroes the set's cell counters
constexpr int PREP_NT = 1024;      // one workgroup per set, and a set is tens of thousands of points: sixteen waves to cover the load latency
__global__ __launch_bounds__(PREP_NT) void grid_prepare_kernel(GridDesc* desc, int* cell, int target_pts) {
    __shared__ float s_mm[(PREP_NT / 64) * 6];
    __shared__ float s_smp[GS_SAMPLES][3];
    __shared__ int s_hist[GS_BINS];
    const int t = blockIdx.x, tid = threadIdx.x;
    GridDesc d = desc[t];
    const float* P = d.pts; const int n = d.n;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = tid; i < n; i += PREP_NT) {
        const float x = P[3 * (size_t)i], y = P[3 * (size_t)i + 1], z = P[3 * (size_t)i + 2];
        mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x); mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y); mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
    }
    block_minmax3<PREP_NT>(mn, mx, s_mm);
    for (int i = tid; i < GS_BINS; i += PREP_NT) s_hist[i] = 0;
    if (tid < GS_SAMPLES && n > 0) {
        const int i = (int)(((long long)tid * n) / GS_SAMPLES + n / (2 * GS_SAMPLES));
        s_smp[tid][0] = P[3 * (size_t)i]; s_smp[tid][1] = P[3 * (size_t)i + 1]; s_smp[tid][2] = P[3 * (size_t)i + 2];
    }
    __syncthreads();
    // distances of every stride-th point to the samples -> histogram over log2(d^2) (LDS atomics)
    const int stride = n > 8192 ? n / 8192 : 1;
    for (int i = tid * stride; i < n; i += PREP_NT * stride) {
        const float x = P[3 * (size_t)i], y = P[3 * (size_t)i + 1], z = P[3 * (size_t)i + 2];
#pragma unroll
        for (int j = 0; j < GS_SAMPLES; ++j) {
            const float dx = x - s_smp[j][0], dy = y - s_smp[j][1], dz = z - s_smp[j][2];
            const float d2 = dx * dx + dy * dy + dz * dz;
            int b = (int)(__float_as_uint(d2) >> 21) - GS_BIN0;
            b = b < 0 ? 0 : (b >= GS_BINS ? GS_BINS - 1 : b);
            atomicAdd(&s_hist[b], 1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        const float emax = fmaxf(ex, fmaxf(ey, ez));
        float c = emax > 0.f ? emax : 1.f;
        if (n > 4 * target_pts) {
            // smallest radius holding `target_pts` points around a sample, on average
            const long long want = ((long long)GS_SAMPLES * target_pts + stride - 1) / stride;
            long long cum = 0; int b = 0;
            for (; b < GS_BINS; ++b) { cum += s_hist[b]; if (cum >= want) break; }
            if (b < GS_BINS) {
                const float r2 = __uint_as_float((unsigned)(b + 1 + GS_BIN0) << 21);      // upper edge of the bin
                const float r = sqrtf(r2) * 1.1f;
                if (r > 0.f && r < c) c = r;
            }
        }
        // dense table: at most d.cell_cap cells
        for (int it = 0; it < 64; ++it) {
            const double cells = ((double)floorf(ex / c) + 1.0) * ((double)floorf(ey / c) + 1.0) * ((double)floorf(ez / c) + 1.0);
            if (cells <= (double)d.cell_cap) break;
            c *= 1.26f;
        }
        d.c = c; d.inv_c = 1.0f / c;
        d.nx = (int)floorf(ex / c) + 1; d.ny = (int)floorf(ey / c) + 1; d.nz = (int)floorf(ez / c) + 1;
        if ((long long)d.nx * d.ny * d.nz > d.cell_cap) { d.nx = d.ny = d.nz = 1; d.c = emax * 2.f + 1.f; d.inv_c = 1.0f / d.c; }      // unreachable safety net
        d.ncell = d.nx * d.ny * d.nz;
        for (int k = 0; k < 3; ++k) { d.lo[k] = mn[k]; d.hi[k] = mx[k]; }
        if (n == 0) { d.nx = d.ny = d.nz = 1; d.ncell = 1; d.lo[0] = d.lo[1] = d.lo[2] = 0.f; d.c = 1.f; d.inv_c = 1.f; }
        {
            float mag = 0.f;
            for (int k = 0; k < 3; ++k) mag = fmaxf(mag, fmaxf(fabsf(d.lo[k]), fabsf(d.hi[k])));
            d.slack = 8.f * FLT_EPSILON * (2.f * mag + (float)(max(d.nx, max(d.ny, d.nz)) + 1) * d.c);
            d.pad_ = 0;
        }
        desc[t] = d;
        s_mm[0] = __int_as_float(d.ncell);
    }
    __syncthreads();
    const int ncell = __float_as_int(s_mm[0]);
    for (int i = tid; i <= ncell; i += PREP_NT) cell[d.cell_off + i] = 0;
}

// per point: its rank inside its cell (returning atomic on the cell counter)
__global__ __launch_bounds__(256) void grid_count_kernel(const GridDesc* __restrict__ desc, int* cell, int* __restrict__ rank) {
    const GridDesc d = desc[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n) return;
    const float x = d.pts[3 * (size_t)i], y = d.pts[3 * (size_t)i + 1], z = d.pts[3 * (size_t)i + 2];
    const int cid = (cell_of(z, d.lo[2], d.inv_c, d.nz) * d.ny + cell_of(y, d.lo[1], d.inv_c, d.ny)) * d.nx + cell_of(x, d.lo[0], d.inv_c, d.nx);
    rank[d.pt_off + i] = atomicAdd(&cell[d.cell_off + cid], 1);
}

// exclusive scan of the cell counters, two launches: block sums, then offsets + local scan (in place; cell[ncell] = n)
__global__ __launch_bounds__(256) void grid_scan_sums_kernel(const GridDesc* __restrict__ desc, const int* __restrict__ cell, int* __restrict__ bsum, int max_blk) {
    __shared__ int s_part[4][2];
    const GridDesc d = desc[blockIdx.y];
    const int blk = blockIdx.x;
    if (blk * SCAN_BLK >= d.ncell) return;
    const int lo = blk * SCAN_BLK, hi = min(lo + SCAN_BLK, d.ncell);
    int s = 0, dummy = 0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) s += cell[d.cell_off + i];
    block_sum2(s, dummy, &s_part[0][0]);
    if (threadIdx.x == 0) bsum[(size_t)blockIdx.y * max_blk + blk] = s;
}

__global__ __launch_bounds__(256) void grid_scan_apply_kernel(const GridDesc* __restrict__ desc, int* cell, const int* __restrict__ bsum, int max_blk) {
    __shared__ int s_wave[4];
    const GridDesc d = desc[blockIdx.y];
    const int blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (blk * SCAN_BLK >= d.ncell) return;
    int base = 0;
    for (int b = 0; b < blk; ++b) base += bsum[(size_t)blockIdx.y * max_blk + b];
    // thread tid owns cells lo + tid*16 .. +16
    const int lo = blk * SCAN_BLK + tid * (SCAN_BLK / 256);
    int v[SCAN_BLK / 256]; int s = 0;
#pragma unroll
    for (int u = 0; u < SCAN_BLK / 256; ++u) { const int i = lo + u; v[u] = i < d.ncell ? cell[d.cell_off + i] : 0; s += v[u]; }
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
    if (lane == 63) s_wave[wid] = incl;
    __syncthreads();
    int run = base + incl - s;
    for (int w2 = 0; w2 < wid; ++w2) run += s_wave[w2];
#pragma unroll
    for (int u = 0; u < SCAN_BLK / 256; ++u) { const int i = lo + u; if (i < d.ncell) cell[d.cell_off + i] = run; run += v[u]; }
    if (blk == (d.ncell - 1) / SCAN_BLK && tid == 0) cell[d.cell_off + d.ncell] = d.n;
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const GridDesc* __restrict__ desc, const int* __restrict__ cell, const int* __restrict__ rank,
                                                           float4* __restrict__ sorted) {
    const GridDesc d = desc[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n) return;
    const float x = d.pts[3 * (size_t)i], y = d.pts[3 * (size_t)i + 1], z = d.pts[3 * (size_t)i + 2];
    const int cid = (cell_of(z, d.lo[2], d.inv_c, d.nz) * d.ny + cell_of(y, d.lo[1], d.inv_c, d.ny)) * d.nx + cell_of(x, d.lo[0], d.inv_c, d.nx);
    sorted[d.pt_off + cell[d.cell_off + cid] + rank[d.pt_off + i]] = make_float4(x, y, z, __int_as_float(i));
}

struct GridSearchArgs {
    const GridDesc* desc; const int* cell; const float4* sorted; const GridJob* jobs; int job0;
    int* retry; int* retry_count;       // rows whose 3^3 block did not settle: answered by the second pass (compacted: a lane that needs a
                                        // larger block would otherwise hold its whole wave in the loop)
    int* work; int* work_count; int work_cap; int* need; int* status;
    float4* ball_q; int* ball_tree; int* ball_count; float ball_scale;      // what the tree build is cut to (KdBalls)
};

// true when the K + 1 best found inside the block [x0..x1] x [y0..y1] x [z0..z1] are final: every point outside it is farther than the (K+1)-th
template <int K>
__device__ __forceinline__ bool grid_settled(const GridDesc& d, float qx, float qy, float qz, int x0, int x1, int y0, int y1, int z0, int z1, const RegSet<K + 1>& rs) {
    // every point outside the scanned block is at least g away (faces on the grid boundary have nothing beyond them)
    float g = FLT_MAX;
    if (x0 > 0) g = fminf(g, qx - (d.lo[0] + (float)x0 * d.c));
    if (x1 < d.nx - 1) g = fminf(g, (d.lo[0] + (float)(x1 + 1) * d.c) - qx);
    if (y0 > 0) g = fminf(g, qy - (d.lo[1] + (float)y0 * d.c));
    if (y1 < d.ny - 1) g = fminf(g, (d.lo[1] + (float)(y1 + 1) * d.c) - qy);
    if (z0 > 0) g = fminf(g, qz - (d.lo[2] + (float)z0 * d.c));
    if (z1 < d.nz - 1) g = fminf(g, (d.lo[2] + (float)(z1 + 1) * d.c) - qz);
    if (g == FLT_MAX) return true;                          // the block is the whole grid
    if (!(g > 0.f)) return false;
    const float gs = g * 0.99998f - d.slack;                // cell boundaries are rounded fp32 products: stay inside, relatively and absolutely
    return gs > 0.f && rs.worst() <= gs * gs;
}

// distance from q to the slab of cell index i along one axis, shortened by the margin (cell boundaries are rounded fp32 products and
// a point's cell comes from a rounded quotient: the margin is far above both)
__device__ __forceinline__ float slab_gap(float q, float lo, float c, int i, int ci, float mg) {
    const float gap = i == ci ? 0.f : (i > ci ? (lo + (float)i * c) - q : q - (lo + (float)(i + 1) * c));
    return fmaxf(gap - mg, 0.f);
}

// rows whose answer depends on the tree (ties, a near-tie at the K-th boundary, too few support points, no settled block) go to the
// work list; the others are written
template <int K, typename OutT>
__device__ __forceinline__ void grid_finish(const GridSearchArgs& a, const GridJob& job, int jid, int q, bool settled, const RegSet<K + 1>& rs) {
    bool hard = !settled || rs.d[K] == FLT_MAX;
    const bool unsettled = hard;
#pragma unroll
    for (int j = 0; j < K; ++j) hard |= rs.d[j] == rs.d[j + 1];
    hard |= rs.d[K] <= rs.d[K - 1] * NEAR_TIE;
    if (hard) {
        if (unsettled) atomicAdd(a.status + 1, 1);                // diagnostics: rows that left the grid for lack of a settled block
        const int w = atomicAdd(a.work_count, 1);
        if (w < a.work_cap) { a.work[2 * (size_t)w] = jid; a.work[2 * (size_t)w + 1] = q; a.need[job.sup] = 1; }
        else atomicOr(a.status, 8);
        // the walk of this row stays (mostly) inside the ball of its (K+1)-th distance: the tree is built where the balls are
        const int bi = atomicAdd(a.ball_count, 1);
        if (bi < GRID_BALL_CAP) {
            const float* qp = job.qpts + 3 * (size_t)q;
            a.ball_q[bi] = make_float4(qp[0], qp[1], qp[2], unsettled ? FLT_MAX : rs.d[K] * a.ball_scale);
            a.ball_tree[bi] = job.sup;
        }
        return;
    }
    OutT* o = reinterpret_cast<OutT*>(job.out) + (size_t)q * K;
#pragma unroll
    for (int j = 0; j < K; ++j) o[j] = (OutT)rs.id[j];
}

// The (2R+1)^3 block around a query by MARKING: a lane whose candidate passes `dist < worst` makes its whole wave execute the
// ~70-instruction sorted insertion, and with 64 queries of the same cells in a wave that is nearly every candidate.  So the scan of
// the (2R+1)^2 rows of cells only marks the candidates inside the radius the block guarantees (distance from the query to the
// nearest block face that has cells beyond it) in one 64-bit mask per row — registers only, four loads in flight — and the
// sorted insertion then runs over the marked candidates alone.  With n1 > 0 the two nearest marked candidates with index < n1 are
// tracked as well (interp_idx of a pyramid = nearest of the prefix, tf_map's sub-sampling).
// Returns 0: rs holds the final K + 1 best; 1: fewer than K + 1 candidates inside the radius (a larger block is needed);
// 2: the masks cannot hold this block (a row of more than 64 candidates) or the radius is not positive (query outside the support box).
// where the records of the cell-sorted array come from: global memory, or the workgroup's LDS copy of the rows of cells its queries touch
struct SrcGlobal { const float4* S; __device__ __forceinline__ float4 operator()(int, int, int i) const { return S[i]; } };
struct SrcLds {
    const float4* S; const float4* lds; const int* delta; int z0, y0, nys; bool staged;
    __device__ __forceinline__ float4 operator()(int z, int y, int i) const { return staged ? lds[i + delta[(z - z0) * nys + (y - y0)]] : S[i]; }
};
template <int K, int R, class Src>
__device__ __forceinline__ int grid_mark_select(const GridDesc& d, const int* __restrict__ cell, const Src S, float qx, float qy, float qz,
                                                RegSet<K + 1>& rs, int n1, float& b0, float& b1, int& i0, float& tau) {
    constexpr int W = 2 * R + 1;
    const int cx = cell_of(qx, d.lo[0], d.inv_c, d.nx), cy = cell_of(qy, d.lo[1], d.inv_c, d.ny), cz = cell_of(qz, d.lo[2], d.inv_c, d.nz);
    const int x0 = max(cx - R, 0), x1 = min(cx + R, d.nx - 1);
    // every point outside the block is at least g away (faces on the grid boundary have nothing beyond them)
    float g = FLT_MAX;
    if (x0 > 0) g = fminf(g, qx - (d.lo[0] + (float)x0 * d.c));
    if (x1 < d.nx - 1) g = fminf(g, (d.lo[0] + (float)(x1 + 1) * d.c) - qx);
    if (cy - R > 0) g = fminf(g, qy - (d.lo[1] + (float)(cy - R) * d.c));
    if (cy + R < d.ny - 1) g = fminf(g, (d.lo[1] + (float)(cy + R + 1) * d.c) - qy);
    if (cz - R > 0) g = fminf(g, qz - (d.lo[2] + (float)(cz - R) * d.c));
    if (cz + R < d.nz - 1) g = fminf(g, (d.lo[2] + (float)(cz + R + 1) * d.c) - qz);
    if (!(g > 0.f)) return 2;
    tau = FLT_MAX;
    if (g < FLT_MAX) { const float gs = g * 0.99998f - d.slack; if (!(gs > 0.f)) return 2; tau = gs * gs; }      // cell boundaries are rounded fp32 products: stay inside, relatively and absolutely
    unsigned long long m[W * W]; int rs0[W * W];
    int cnt = 0; bool long_row = false;
#pragma unroll
    for (int r = 0; r < W * W; ++r) {
        const int z = cz + r / W - R, y = cy + r % W - R;
        m[r] = 0ull; rs0[r] = 0;
        if (z < 0 || z >= d.nz || y < 0 || y >= d.ny) continue;
        const int row = (z * d.ny + y) * d.nx;      // x is the fastest cell dimension: the cells x0..x1 of a row are one contiguous range
        const int s = cell[row + x0], e = cell[row + x1 + 1];
        rs0[r] = s;
        if (e - s > 64) { long_row = true; continue; }
        for (int i = s; i < e; i += 4) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = S(z, y, min(i + u, e - 1));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float dx = qx - p[u].x, dy = qy - p[u].y, dz = qz - p[u].z;
                float dist = dx * dx; dist = dist + dy * dy; dist = dist + dz * dz;
                if (i + u < e && dist < tau) { m[r] |= 1ull << (i + u - s); ++cnt; }
            }
        }
    }
    if (long_row) return 2;
    if (cnt < K + 1) {
        if (n1 > 0) {      // the prefix job may still be answerable from the few candidates inside the radius
#pragma unroll
            for (int r = 0; r < W * W; ++r) {
                unsigned long long mm = m[r];
                while (mm) {
                    const int bpos = __ffsll((unsigned long long)mm) - 1;
                    mm &= mm - 1ull;
                    const float4 p = S(cz + r / W - R, cy + r % W - R, rs0[r] + bpos);
                    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
                    float dist = dx * dx; dist = dist + dy * dy; dist = dist + dz * dz;
                    if (__float_as_int(p.w) < n1 && dist < b1) { if (dist < b0) { b1 = b0; b0 = dist; i0 = __float_as_int(p.w); } else b1 = dist; }
                }
            }
        }
        return 1;
    }
    rs.init();
#pragma unroll
    for (int r = 0; r < W * W; ++r) {
        unsigned long long mm = m[r];
        while (mm) {
            const int bpos = __ffsll((unsigned long long)mm) - 1;
            mm &= mm - 1ull;
            const float4 p = S(cz + r / W - R, cy + r % W - R, rs0[r] + bpos);
            const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
            float dist = dx * dx; dist = dist + dy * dy; dist = dist + dz * dz;
            const int id = __float_as_int(p.w);
            if (id < n1 && dist < b1) { if (dist < b0) { b1 = b0; b0 = dist; i0 = id; } else b1 = dist; }
            if (dist < rs.worst()) rs.add(dist, id);
        }
    }
    return 0;
}

// first pass: grid.y = job (one support set x one query set x one output block), grid.x = blocks of 256 queries; the 3^3 block.
// job.job1 >= 0 (pyramid): the K = 1 job over the first n1 support points is answered in the same scan when its nearest prefix point
// lies inside the guaranteed radius and clear of the runner-up; otherwise that job's own (coarser) grid answers the row in its retry pass.
// what a query does in the first pass once it knows its coordinates (shared by the two forms of the kernel)
template <int K, typename OutT, class Src>
__device__ __forceinline__ void grid_search_query(const GridSearchArgs& a, const GridJob& job, int jid, const GridDesc& d, const Src src, int q, float qx, float qy, float qz) {
    RegSet<K + 1> rs;
    float b0 = FLT_MAX, b1 = FLT_MAX, tau = 0.f; int i0 = 0;
    const bool fuse1 = K == 16 && job.job1 >= 0;
    const int st = grid_mark_select<K, 1>(d, a.cell + d.cell_off, src, qx, qy, qz, rs, fuse1 ? job.n1 : 0, b0, b1, i0, tau);
    if (fuse1) {
        // final iff the masks held the block, the nearest prefix point lies inside the guaranteed radius and the runner-up (seen, or anything beyond the
        // radius) is clear of it by more than the near-tie margin
        if (st != 2 && b0 < tau && b0 * NEAR_TIE < fminf(b1, tau)) {
            const GridJob j1 = a.jobs[job.job1];
            reinterpret_cast<OutT*>(j1.out)[q] = (OutT)i0;
        } else {
            const int w = atomicAdd(a.retry_count + 1, 1);
            int* r1 = a.retry + 2 * (size_t)a.work_cap;          // retry list of K = 1 follows the one of K = 16
            r1[2 * (size_t)w] = job.job1 | RETRY_FROM_R1; r1[2 * (size_t)w + 1] = q;      // its own grid has not been tried yet
        }
    }
    if (st != 0) {
        const int w = atomicAdd(a.retry_count, 1);             // < work_cap by construction (one entry per query at most)
        a.retry[2 * (size_t)w] = st == 2 ? (jid | RETRY_FROM_R1) : jid; a.retry[2 * (size_t)w + 1] = q;      // st == 2: the 3^3 block can still settle by streaming
        return;
    }
    grid_finish<K, OutT>(a, job, jid, q, true, rs);
}

template <int K, typename OutT>
__global__ __launch_bounds__(256) SSDR_WAVES_PER_EU(5) void grid_search_kernel(GridSearchArgs a) {      // 94 registers instead of 106: five waves per SIMD, no spills (six spill: slower)
    int bx, by; xcd_tile_map(bx, by);          // a job's support set (one tile's records and cell table) into one XCD's L2
    const int jid = a.job0 + by;
    const GridJob job = a.jobs[jid];
    const int qi = bx * 256 + (int)threadIdx.x;
    if (qi >= job.nq) return;
    const GridDesc d = a.desc[job.sup];
    int q = qi; float qx, qy, qz;
    if (job.ord >= 0) {       // the queries are the points of set `ord`: take them in its cell order (lanes of a wave scan the same cells)
        const float4 r = a.sorted[a.desc[job.ord].pt_off + qi];
        q = __float_as_int(r.w); qx = r.x; qy = r.y; qz = r.z;
    } else {
        qx = job.qpts[3 * (size_t)q]; qy = job.qpts[3 * (size_t)q + 1]; qz = job.qpts[3 * (size_t)q + 2];
    }
    grid_search_query<K, OutT>(a, job, jid, d, SrcGlobal{a.sorted + d.pt_off}, q, qx, qy, qz);
}

// The north_star's form of the first pass, built to be measured beside the one above: the workgroup's 256 cell-sorted queries touch a few rows of
// cells (x is the fastest cell dimension: ~7 rows at ~0.8 points per cell), so the rows z-1..z+1 x y-1..y+1 around them — whole rows, ~30 records each —
// are copied into LDS once (a wave per row) and every lane's scan of its 3 x 3 rows reads the records from there instead of through the vector-memory
// path (16 bytes per candidate per lane).  A workgroup whose queries straddle the end of a z slab (its rows span the whole y range) or whose rows hold
// more than the buffer keeps the global reads.  Same arithmetic, same results.
constexpr int GL_CAP = 2048, GL_ROWS = 96;
template <int K, typename OutT>
__global__ __launch_bounds__(256) void grid_search_lds_kernel(GridSearchArgs a) {
    __shared__ float4 s_rec[GL_CAP];
    __shared__ int s_delta[GL_ROWS], s_start[GL_ROWS], s_len[GL_ROWS], s_mm[4], s_tot;
    int bx, by; xcd_tile_map(bx, by);
    const int jid = a.job0 + by;
    const GridJob job = a.jobs[jid];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int qi = bx * 256 + tid;
    const bool live = qi < job.nq;
    const GridDesc d = a.desc[job.sup];
    int q = qi; float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) {
        if (job.ord >= 0) { const float4 r = a.sorted[a.desc[job.ord].pt_off + qi]; q = __float_as_int(r.w); qx = r.x; qy = r.y; qz = r.z; }
        else { qx = job.qpts[3 * (size_t)q]; qy = job.qpts[3 * (size_t)q + 1]; qz = job.qpts[3 * (size_t)q + 2]; }
    }
    if (tid == 0) { s_mm[0] = 0x7fffffff; s_mm[1] = -1; s_mm[2] = 0x7fffffff; s_mm[3] = -1; }
    __syncthreads();
    if (live) {
        const int cy = cell_of(qy, d.lo[1], d.inv_c, d.ny), cz = cell_of(qz, d.lo[2], d.inv_c, d.nz);
        atomicMin(&s_mm[0], cz); atomicMax(&s_mm[1], cz); atomicMin(&s_mm[2], cy); atomicMax(&s_mm[3], cy);
    }
    __syncthreads();
    if (s_mm[1] < 0) return;                          // a workgroup past the end of this job's queries (the grid is sized by the largest job)
    const int z0 = max(s_mm[0] - 1, 0), z1 = min(s_mm[1] + 1, d.nz - 1), y0 = max(s_mm[2] - 1, 0), y1 = min(s_mm[3] + 1, d.ny - 1);
    const int nys = y1 - y0 + 1, nrows = (z1 - z0 + 1) * nys;
    const int* cell = a.cell + d.cell_off;
    const float4* S = a.sorted + d.pt_off;
    bool staged = nrows > 0 && nrows <= GL_ROWS;
    if (staged) {
        if (tid < nrows) { const int row = ((z0 + tid / nys) * d.ny + (y0 + tid % nys)) * d.nx; const int gs = cell[row]; s_start[tid] = gs; s_len[tid] = cell[row + d.nx] - gs; }
        __syncthreads();
        if (tid == 0) { int run = 0; for (int r = 0; r < nrows; ++r) { s_delta[r] = run - s_start[r]; run += s_len[r]; } s_tot = run; }
        __syncthreads();
        staged = s_tot <= GL_CAP;
        if (staged)
            for (int r = wid; r < nrows; r += 4)              // a wave per row of cells
                for (int i = lane; i < s_len[r]; i += 64) s_rec[s_delta[r] + s_start[r] + i] = S[s_start[r] + i];
        __syncthreads();
    }
    if (!live) return;
    grid_search_query<K, OutT>(a, job, jid, d, SrcLds{S, s_rec, s_delta, z0, y0, nys, staged}, q, qx, qy, qz);
}

// second pass over the rows the first one left (compacted: a lane that needs a larger block no longer holds its wave): 3^3
// (for the rows the masks could not hold and for K = 1 rows on their own grid), then the 5^3 and 7^3 shells cut to the ball of the current
// (K+1)-th distance.  The rows are few (2-4 % of the queries: a few hundred waves at a lane per row) and each waits on a chain of dependent
// loads (cell table, then the records of one row of cells after the other), so a row is spread over RL lanes: lane l of a row's group takes
// the rows of cells l, l + RL, ... of the block or shell, four records in flight, and appends the candidates closer than the row's current
// (K+1)-th distance to the group's list in LDS; the group's first lane alone runs the sorted insertion over that list (drained when it
// may overflow and at the end of a block / shell).  Which of several equal candidates enters depends on the order of the appends, but a
// row with equal distances among its K + 1 best is handed to the tree anyway (grid_finish) and a tie behind them does not change the
// answer, so the results are those of the lane-per-row scan.
constexpr int RL = 8, RCB = 64;          // lanes per row; candidates a group's list holds (a scan step appends at most 4 per lane)
template <int K, typename OutT>
__global__ __launch_bounds__(64) void grid_retry_kernel(GridSearchArgs a) {
    constexpr int RPW = 64 / RL;
    __shared__ float c_d[RPW][RCB];
    __shared__ int c_i[RPW][RCB];
    __shared__ int c_n[RPW];
    __shared__ float s_gw[RPW];
    __shared__ int s_set[RPW];
    const int lane = threadIdx.x, sub = lane % RL, grp = lane / RL;
    const int n = min(*a.retry_count, a.work_cap);
    for (int e0 = blockIdx.x * RPW; e0 < n; e0 += gridDim.x * RPW) {          // uniform over the wave
        const bool live = e0 + grp < n;
        const int e = min(e0 + grp, n - 1);
        const int tag = a.retry[2 * (size_t)e], q = a.retry[2 * (size_t)e + 1];
        const int jid = tag & ~RETRY_FROM_R1;
        const GridJob job = a.jobs[jid];
        const GridDesc d = a.desc[job.sup];
        const int* __restrict__ cell = a.cell + d.cell_off;
        const float4* __restrict__ S = a.sorted + d.pt_off;
        const float qx = job.qpts[3 * (size_t)q], qy = job.qpts[3 * (size_t)q + 1], qz = job.qpts[3 * (size_t)q + 2];
        const int cx = cell_of(qx, d.lo[0], d.inv_c, d.nx), cy = cell_of(qy, d.lo[1], d.inv_c, d.ny), cz = cell_of(qz, d.lo[2], d.inv_c, d.nz);
        RegSet<K + 1> G;                        // the row's set: kept by the first lane of the group
        G.init();
        float gw = FLT_MAX; bool settled = false;
        const float mg = d.c * 1.0e-3f + 1.0e-6f * fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fabsf(qz));
        if (sub == 0) c_n[grp] = 0;
        auto drain = [&]() {                    // wave-uniform: the first lanes insert their group's list, everyone learns the new (K+1)-th distance
            wave_sync();
            if (sub == 0) {
                const int m = c_n[grp];
                for (int j = 0; j < m; ++j) { const float dist = c_d[grp][j]; if (dist < G.worst()) G.add(dist, c_i[grp][j]); }
                c_n[grp] = 0; s_gw[grp] = G.worst();
            }
            wave_sync();
            gw = s_gw[grp];
        };
        for (int R = 1; R <= GS_RMAX; ++R) {
            const int x0 = max(cx - R, 0), x1 = min(cx + R, d.nx - 1), y0 = max(cy - R, 0), y1 = min(cy + R, d.ny - 1), z0 = max(cz - R, 0), z1 = min(cz + R, d.nz - 1);
            // only the cells of the block (R = 1) or of the new shell that the ball of the current (K+1)-th distance reaches (fewer than K + 1 found so
            // far: everything)
            const int W = 2 * R + 1;
            const float r = sqrtf(gw) * 1.00001f + mg;
            const float r2 = r * r;
            const int zs = max(z0, cell_of(qz - r, d.lo[2], d.inv_c, d.nz)), ze = min(z1, cell_of(qz + r, d.lo[2], d.inv_c, d.nz));
            const int ix0 = cx - (R - 1), ix1 = cx + (R - 1);
            for (int t0 = 0; t0 < W * W; t0 += RL) {                           // uniform: the lanes' rows of cells advance together
                const int t = t0 + sub;
                int cs[2] = {1, 1}, ce[2] = {0, 0}, row = 0;
                if (!settled && t < W * W) {
                    const int z = cz - R + t / W, y = cy - R + t % W;
                    const float gz = slab_gap(qz, d.lo[2], d.c, z, cz, mg);
                    const float remz = r2 - gz * gz;
                    if (z >= zs && z <= ze && remz > 0.f) {
                        const float hy = sqrtf(remz);
                        const int ys = max(y0, cell_of(qy - hy, d.lo[1], d.inv_c, d.ny)), ye = min(y1, cell_of(qy + hy, d.lo[1], d.inv_c, d.ny));
                        const float gy = slab_gap(qy, d.lo[1], d.c, y, cy, mg);
                        const float rem = remz - gy * gy;
                        if (y >= ys && y <= ye && rem > 0.f) {
                            const float h = sqrtf(rem);
                            const int xs = max(x0, cell_of(qx - h, d.lo[0], d.inv_c, d.nx)), xe = min(x1, cell_of(qx + h, d.lo[0], d.inv_c, d.nx));
                            row = (z * d.ny + y) * d.nx;
                            const bool inner = R > 1 && abs(z - cz) < R && abs(y - cy) < R;      // this row of cells crosses the block already scanned
                            // up to two pieces of the row: left and right of the scanned block (or the whole reach)
                            cs[0] = xs; ce[0] = inner ? min(xe + 1, ix0) : xe + 1;
                            if (inner) { cs[1] = max(xs, ix1 + 1); ce[1] = xe + 1; }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    int i = 0, e1 = 0;
                    if (cs[u] < ce[u]) { i = cell[row + cs[u]]; e1 = cell[row + ce[u]]; }
                    while (__ballot(i < e1)) {
                        wave_sync();
                        if (__ballot(c_n[grp] > RCB - 4 * RL)) drain();
                        if (i < e1) {
                            float4 p[4];
#pragma unroll
                            for (int v = 0; v < 4; ++v) p[v] = S[min(i + v, e1 - 1)];
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const float dx = qx - p[v].x, dy = qy - p[v].y, dz = qz - p[v].z;
                                float dist = dx * dx; dist = dist + dy * dy; dist = dist + dz * dz;
                                if (i + v < e1 && dist < gw) { const int pos = atomicAdd(&c_n[grp], 1); c_d[grp][pos] = dist; c_i[grp][pos] = __float_as_int(p[v].w); }
                            }
                            i += 4;
                        }
                    }
                }
            }
            drain();
            if (sub == 0 && !settled) s_set[grp] = grid_settled<K>(d, qx, qy, qz, x0, x1, y0, y1, z0, z1, G) ? 1 : 0;
            wave_sync();
            if (!settled) settled = s_set[grp] != 0;
            wave_sync();                        // s_set is rewritten by the next shell
            if (__ballot(!settled) == 0ull) break;
        }
        if (sub == 0 && live) grid_finish<K, OutT>(a, job, jid, q, settled, G);
    }
}

}  // namespace

int grid_build(GridForest& g, const std::vector<GridDesc>& sets_in, int target_pts, hipStream_t s) {
    std::vector<GridDesc> sets = sets_in;
    long pt = 0, cl = 0; int maxn = 0; long maxcap = 0;
    for (auto& d : sets) {
        d.pt_off = (int)pt; d.cell_off = (int)cl; d.cell_cap = 8 * d.n + 64;
        pt += d.n; cl += (long)d.cell_cap + 1; maxn = std::max(maxn, d.n); maxcap = std::max(maxcap, (long)d.cell_cap);
        if (pt > 0x3fffffffL || cl > 0x7ffffff0L) { set_error("grid_build: too many points"); return SSDR_ERR_INVALID; }
    }
    g.nsets = (int)sets.size(); g.total_pts = (int)pt; g.max_n = maxn;
    g.max_blk = (int)((maxcap + SCAN_BLK - 1) / SCAN_BLK);
    if (g.nsets == 0) return SSDR_OK;
    SSDR_TRY(g.desc.reserve(sizeof(GridDesc) * sets.size()));
    SSDR_TRY(g.cell.reserve(4 * (size_t)cl + 16)); SSDR_TRY(g.rank.reserve(4 * (size_t)std::max(pt, 1L))); SSDR_TRY(g.sorted.reserve(16 * (size_t)std::max(pt, 1L)));
    SSDR_TRY(g.bsum.reserve(4 * (size_t)g.nsets * g.max_blk));
    SSDR_TRY(g.need.reserve(4 * (2 * (size_t)g.nsets + 16)));
    SSDR_TRY(g.balls.reserve((size_t)GRID_BALL_CAP * 20));
    // descriptors travel through a ring of pinned staging buffers (ssdr_internal.hpp: StagingRing)
    {
        GridDesc* st = nullptr;
        const int slot = g.staging.acquire(sets.size(), &st);
        if (slot < 0) { set_error("grid_build: pinned staging buffer"); return SSDR_ERR_HIP; }
        memcpy(st, sets.data(), sizeof(GridDesc) * sets.size());
        SSDR_HIP(hipMemcpyAsync(g.desc.p, st, sizeof(GridDesc) * sets.size(), hipMemcpyHostToDevice, s));
        if (g.staging.release(slot, s)) { set_error("grid_build: staging event"); return SSDR_ERR_HIP; }
    }
    SSDR_HIP(hipMemsetAsync(g.need.p, 0, 4 * (2 * (size_t)g.nsets + 16), s));      // need[nsets], counters()[16], need2[nsets]
    GridDesc* dd = g.desc.as<GridDesc>();
    const dim3 gp((unsigned)((maxn + 255) / 256), (unsigned)g.nsets), gb((unsigned)g.max_blk, (unsigned)g.nsets);
    ProfScope prof("knn_grid_build", s, 28.0 * (double)pt);
    hipLaunchKernelGGL(grid_prepare_kernel, dim3(g.nsets), dim3(PREP_NT), 0, s, dd, g.cell.as<int>(), target_pts);
    hipLaunchKernelGGL(grid_count_kernel, gp, dim3(256), 0, s, dd, g.cell.as<int>(), g.rank.as<int>());
    hipLaunchKernelGGL(grid_scan_sums_kernel, gb, dim3(256), 0, s, dd, g.cell.as<int>(), g.bsum.as<int>(), g.max_blk);
    hipLaunchKernelGGL(grid_scan_apply_kernel, gb, dim3(256), 0, s, dd, g.cell.as<int>(), g.bsum.as<int>(), g.max_blk);
    hipLaunchKernelGGL(grid_scatter_kernel, gp, dim3(256), 0, s, dd, g.cell.as<int>(), g.rank.as<int>(), g.sorted.as<float4>());
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

// jobs [job0, job0 + njobs) of the table uploaded by grid_set_jobs, all with the same K; rows that need the tree go to the K's work list
int grid_search(const GridForest& g, int job0, int njobs, int max_nq, int K, bool out_i64, hipStream_t s, double prof_bytes) {
    if (njobs <= 0) return SSDR_OK;
    if (K != 16 && K != 1) { set_error("grid search: K=%d has no instantiation (1, 16)", K); return SSDR_ERR_UNSUPPORTED; }
    const int wl = K == 16 ? 0 : 1;
    int* ctr = g.counters();            // [0], [1]: work counts of the two lists, [2]: status, [3]: unsettled rows, [4], [5]: retry counts
    GridSearchArgs a{g.desc.as<GridDesc>(), g.cell.as<int>(), g.sorted.as<float4>(), g.jobs.as<GridJob>(), job0,
                     g.work_list(2 + wl), ctr + 4 + wl, g.work_list(wl), ctr + wl, g.work_cap, g.need.as<int>(), ctr + 2,
                     g.ball_q(), g.ball_tree(), ctr + 6, g.ball_scale};      // K = 16: the retry list / counter of K = 1 lie right behind its own
    const dim3 grid((unsigned)((max_nq + 255) / 256), (unsigned)njobs);
    const dim3 rgrid((unsigned)std::max(1, std::min(g.work_cap / 64 + 1, ctx().num_cu * 16)));
    const bool first = max_nq > 0;      // max_nq == 0: the jobs were answered inside another scan, only their left-over rows remain
    // profiler sites: one per kernel template (the first pass carries the stage's algorithmic bytes; the retry answers what it left over)
    static const bool lds_form = [] { const char* e = getenv("SSDR_KNN_LDS"); return e && e[0] == '1'; }();      // the LDS-staged first pass (measured beside the default, profiles/)
    if (first) {
        ProfScope prof(K == 16 ? "grid_search_kernel<16>" : "grid_search_kernel<1>", s, prof_bytes);
        if (K == 16) {
            if (out_i64) { if (lds_form) hipLaunchKernelGGL((grid_search_lds_kernel<16, int64_t>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((grid_search_kernel<16, int64_t>), grid, dim3(256), 0, s, a); }
            else { if (lds_form) hipLaunchKernelGGL((grid_search_lds_kernel<16, int32_t>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((grid_search_kernel<16, int32_t>), grid, dim3(256), 0, s, a); }
        } else {
            if (out_i64) hipLaunchKernelGGL((grid_search_kernel<1, int64_t>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((grid_search_kernel<1, int32_t>), grid, dim3(256), 0, s, a);
        }
    }
    {
        ProfScope prof(K == 16 ? "grid_retry_kernel<16>" : "grid_retry_kernel<1>", s, first ? 0.0 : prof_bytes);
        if (K == 16) { if (out_i64) hipLaunchKernelGGL((grid_retry_kernel<16, int64_t>), rgrid, dim3(64), 0, s, a); else hipLaunchKernelGGL((grid_retry_kernel<16, int32_t>), rgrid, dim3(64), 0, s, a); }
        else { if (out_i64) hipLaunchKernelGGL((grid_retry_kernel<1, int64_t>), rgrid, dim3(64), 0, s, a); else hipLaunchKernelGGL((grid_retry_kernel<1, int32_t>), rgrid, dim3(64), 0, s, a); }
    }
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int grid_set_jobs(GridForest& g, const std::vector<GridJob>& jobs, hipStream_t s) {
    if (jobs.empty()) return SSDR_OK;
    long tq = 0; for (auto& j : jobs) tq += j.nq;
    g.work_cap = (int)std::min(std::max(tq, 1024L), 0x3fffffffL);      // per list: every query of every job can go to the tree
    SSDR_TRY(g.work.reserve(4 * 8 * (size_t)g.work_cap));      // hand-over lists (K = 16, K = 1) and retry lists
    SSDR_TRY(g.jobs.reserve(sizeof(GridJob) * jobs.size()));
    {
        GridJob* st = nullptr;
        const int slot = g.jstaging.acquire(jobs.size(), &st);
        if (slot < 0) { set_error("grid_set_jobs: pinned staging buffer"); return SSDR_ERR_HIP; }
        memcpy(st, jobs.data(), sizeof(GridJob) * jobs.size());
        SSDR_HIP(hipMemcpyAsync(g.jobs.p, st, sizeof(GridJob) * jobs.size(), hipMemcpyHostToDevice, s));
        if (g.jstaging.release(slot, s)) { set_error("grid_set_jobs: staging event"); return SSDR_ERR_HIP; }
    }
    return SSDR_OK;
}

}  // namespace ssdr
