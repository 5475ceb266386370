// Stable LSD radix sort of (u64 key, u32 value) pairs for gfx950, 8 bits per pass, over one or several independent
// SEGMENTS of one buffer in the same launches (the 16 rooms of a batch sort together: launch count, not bytes, bounds
// sorts of 0.1-1 M pairs).
//
// Used by the voxel-grid subsampling (sort points by voxel key, ties keep input order), the tile generator (sort by
// distance), the region ranking and the emulation of the reference's output order.  HBM-bound integer work: per
// executed pass each pair is read twice and written once (histogram, scatter).  Passes whose digit is identical in
// every key of every segment are skipped on the device without a host round-trip: a pre-pass accumulates AND / OR of
// the keys per segment and every kernel of pass p derives "executed?" and its ping-pong parity from those words.
//
// Layout: segment s owns the slots [off[s], off[s+1]) of keys / vals (off[] multiples of the 2048-pair tile, so a tile
// never straddles segments); its live pairs are the first cnt[s] slots, cnt[s] = min(d_cnt[s], n_host[s]) when a
// device count is given.  Stability inside a tile: wave w owns 512 consecutive pairs and ranks them with wave-private
// digit counters (8 ballots give every lane the mask of lanes holding the same digit), two barriers per tile.
#include "ssdr_internal.hpp"

namespace ssdr {
namespace {

constexpr int RS_BS = 256, RS_ITEMS = 8, RS_TILE = RS_BS * RS_ITEMS;
static_assert(RS_TILE == RADIX_TILE, "RADIX_TILE in ssdr_internal.hpp");

struct SegTab { int nseg; int off[RADIX_MAX_SEG + 1]; int n_host[RADIX_MAX_SEG]; };

struct SortPtrs {
    uint64_t* k[2]; uint32_t* v[2];
    unsigned long long* andor;   // [nseg][2] = AND, OR of the segment's keys
    unsigned long long* andor_part;   // [nseg][RS_ANDOR_G][2] partials
    unsigned* hist;              // [256][nblocks_max]
    unsigned* tot;               // [nseg][256] digit totals of the current pass
    const int* d_cnt;            // optional device counts per segment
    int home;                    // which of k[] / v[] is the caller's buffer (the sorted pairs end up there); the input is always in k[0] / v[0]
    int nblocks_max;
    int shift;                   // the sort key is the part of the 64-bit word from this bit up (pass p: bits [shift + 8p, shift + 8p + 8))
    int kv;                      // 0: keys only — no value array is read or written (a payload may sit below `shift` inside the word)
    SegTab seg;
};
__device__ __forceinline__ unsigned digit_of(const SortPtrs& s, unsigned long long key, int p) { return (unsigned)((key >> (s.shift + 8 * p)) & 0xffull); }

__device__ __forceinline__ int seg_count(const SortPtrs& s, int sg) { return s.d_cnt ? min(s.d_cnt[sg], s.seg.n_host[sg]) : s.seg.n_host[sg]; }
__device__ __forceinline__ int seg_of_tile(const SortPtrs& s, int b) {
    int sg = 0;
    while (sg + 1 < s.seg.nseg && b * RS_TILE >= s.seg.off[sg + 1]) ++sg;
    return sg;
}
__device__ __forceinline__ unsigned long long union_diff(const SortPtrs& s) {
    unsigned long long diff = 0ull;
    for (int sg = 0; sg < s.seg.nseg; ++sg) {
        const unsigned long long a = s.andor[2 * sg], o = s.andor[2 * sg + 1];
        if (!(a == ~0ull && o == 0ull)) diff |= a ^ o;        // an empty segment contributes nothing
    }
    return diff;
}
__device__ __forceinline__ bool pass_runs(const SortPtrs& s, int p, int& parity) {
    const unsigned long long diff = union_diff(s) >> s.shift;
    int par = 0;
    for (int q = 0; q < p; ++q) par ^= ((diff >> (8 * q)) & 0xffull) != 0;
    parity = par;
    return ((diff >> (8 * p)) & 0xffull) != 0;
}

// grid (RS_ANDOR_G, segment): partial AND / OR of a slice of the segment's keys -> part[segment][block][2].  (Thousands
// of workgroups folding into one word per segment with atomics serialise at ~10 ns each; a second tiny kernel folds.)
constexpr int RS_ANDOR_G = 256;
__global__ __launch_bounds__(RS_BS) void rs_andor(SortPtrs s) {
    __shared__ unsigned long long sa[RS_BS / 64], so[RS_BS / 64];
    const int sg = blockIdx.y, n = seg_count(s, sg);
    const uint64_t* K = s.k[0] + s.seg.off[sg];
    unsigned long long a = ~0ull, o = 0ull;
    const int stride = gridDim.x * RS_BS;
    int i = blockIdx.x * RS_BS + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const unsigned long long k0 = K[i], k1 = K[i + stride], k2 = K[i + 2 * stride], k3 = K[i + 3 * stride];
        a &= k0 & k1 & k2 & k3; o |= k0 | k1 | k2 | k3;
    }
    for (; i < n; i += stride) { unsigned long long k = K[i]; a &= k; o |= k; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned lo = __shfl_xor((unsigned)a, off), hi = __shfl_xor((unsigned)(a >> 32), off);
        a &= ((unsigned long long)hi << 32) | lo;
        lo = __shfl_xor((unsigned)o, off); hi = __shfl_xor((unsigned)(o >> 32), off);
        o |= ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; so[threadIdx.x >> 6] = o; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < RS_BS / 64; ++w) { a &= sa[w]; o |= so[w]; }
        s.andor_part[2 * ((size_t)sg * RS_ANDOR_G + blockIdx.x)] = a;
        s.andor_part[2 * ((size_t)sg * RS_ANDOR_G + blockIdx.x) + 1] = o;
    }
}
// one wavefront per segment folds the partials (an empty segment keeps AND = ~0, OR = 0)
__global__ __launch_bounds__(64) void rs_andor_fold(SortPtrs s, int nparts) {
    const int sg = blockIdx.x, lane = threadIdx.x;
    unsigned long long a = ~0ull, o = 0ull;
    for (int j = lane; j < nparts; j += 64) { a &= s.andor_part[2 * ((size_t)sg * RS_ANDOR_G + j)]; o |= s.andor_part[2 * ((size_t)sg * RS_ANDOR_G + j) + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned lo = __shfl_xor((unsigned)a, off), hi = __shfl_xor((unsigned)(a >> 32), off);
        a &= ((unsigned long long)hi << 32) | lo;
        lo = __shfl_xor((unsigned)o, off); hi = __shfl_xor((unsigned)(o >> 32), off);
        o |= ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 0) { s.andor[2 * sg] = a; s.andor[2 * sg + 1] = o; }
}

__global__ __launch_bounds__(RS_BS) void rs_hist(SortPtrs s, int p) {
    __shared__ unsigned h[256];
    int par; if (!pass_runs(s, p, par)) return;
    const int nb = s.seg.off[s.seg.nseg] / RS_TILE;
    const uint64_t* K = s.k[par];
    for (int b = blockIdx.x; b < nb; b += gridDim.x) {
        const int sg = seg_of_tile(s, b), end = s.seg.off[sg] + seg_count(s, sg);
        if (b * RS_TILE >= end) continue;                 // tile past the segment's live pairs: rs_scan never reads its row
        h[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            int i = b * RS_TILE + j * RS_BS + threadIdx.x;
            if (i < end) atomicAdd(&h[digit_of(s, K[i], p)], 1u);
        }
        __syncthreads();
        s.hist[(size_t)threadIdx.x * s.nblocks_max + b] = h[threadIdx.x];
        __syncthreads();
    }
}

// block (d, segment): exclusive scan of hist[d][tiles of the segment] over the tile axis, digit total to tot[segment][d]
__global__ __launch_bounds__(RS_BS) void rs_scan(SortPtrs s, int p) {
    __shared__ unsigned wsum[RS_BS / 64];
    __shared__ unsigned carry_s;
    int par; if (!pass_runs(s, p, par)) return;
    const int sg = blockIdx.y;
    const int b0 = s.seg.off[sg] / RS_TILE, nb = (seg_count(s, sg) + RS_TILE - 1) / RS_TILE;     // live tiles only
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    unsigned* row = s.hist + (size_t)blockIdx.x * s.nblocks_max + b0;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += RS_BS) {
        const int e = base + tid;
        const unsigned x = e < nb ? row[e] : 0u;
        unsigned incl = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { unsigned y = __shfl_up(incl, off); if (lane >= off) incl += y; }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        unsigned wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < RS_BS / 64; ++w) { unsigned c = wsum[w]; if (w < wid) wbase += c; tot += c; }
        const unsigned carry = carry_s;
        if (e < nb) row[e] = carry + wbase + incl - x;
        __syncthreads();
        if (tid == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (tid == 0) s.tot[sg * 256 + blockIdx.x] = carry_s;
}

// The tile is first ordered by digit in LDS (stable: waves in order, rounds in order, lanes in order), then written
// out position by position, so consecutive threads store consecutive pairs of one digit's run: whole 64-128 B pieces
// of a cache line per run instead of 64 isolated 8-byte stores per wave.
template <bool KV>
__global__ __launch_bounds__(RS_BS) void rs_scatter(SortPtrs s, int p) {
    constexpr int NW = RS_BS / 64;
    __shared__ unsigned s_cnt[NW][256];      // per wave: running count of each digit, then the wave's start inside the tile
    __shared__ unsigned s_run[256];          // global position of the digit's run of this tile minus its start inside the tile
    __shared__ unsigned s_wt[2][NW];
    __shared__ uint64_t s_key[RS_TILE];
    __shared__ uint32_t s_val[KV ? RS_TILE : 1];
    int par; if (!pass_runs(s, p, par)) return;
    const int nb = s.seg.off[s.seg.nseg] / RS_TILE;
    const uint64_t* K = s.k[par]; const uint32_t* V = s.v[par];
    uint64_t* KO = s.k[par ^ 1]; uint32_t* VO = s.v[par ^ 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int w = 0; w < NW; ++w) s_cnt[w][tid] = 0;
    __syncthreads();
    for (int b = blockIdx.x; b < nb; b += gridDim.x) {
        const int sg = seg_of_tile(s, b), end = s.seg.off[sg] + seg_count(s, sg);
        if (b * RS_TILE >= end) continue;
        const int ntile = min(RS_TILE, end - b * RS_TILE);
        uint64_t key[RS_ITEMS]; uint32_t val[RS_ITEMS]; unsigned rk[RS_ITEMS];
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const int i = b * RS_TILE + wid * (RS_TILE / NW) + j * 64 + lane;
            key[j] = 0; val[j] = 0;
            if (i < end) { key[j] = K[i]; if (KV) val[j] = V[i]; }
        }
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const int i = b * RS_TILE + wid * (RS_TILE / NW) + j * 64 + lane;
            const bool valid = i < end;
            const unsigned d = valid ? digit_of(s, key[j], p) : 0u;
            unsigned long long m = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool one = (d >> bit) & 1;
                const unsigned long long bal = __ballot(valid && one);
                m &= one ? bal : ~bal;
            }
            const unsigned before = valid ? s_cnt[wid][d] : 0u;         // this digit's count in the wave's earlier rounds
            rk[j] = valid ? before + (unsigned)__popcll(m & lt) : 0xffffffffu;
            (void)__ballot(1);                                            // every lane has read `before` ...
            if (valid && (m & lt) == 0) s_cnt[wid][d] = before + (unsigned)__popcll(m);   // ... then the group leader adds
        }
        __syncthreads();
        {   // thread tid owns digit tid: start of the digit inside the tile and in the segment (two exclusive scans)
            unsigned cw[NW], tcount = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { cw[w] = s_cnt[w][tid]; tcount += cw[w]; }
            const unsigned gtot = s.tot[sg * 256 + tid];
            unsigned inc_t = tcount, inc_g = gtot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                unsigned y = __shfl_up(inc_t, off), z = __shfl_up(inc_g, off);
                if (lane >= off) { inc_t += y; inc_g += z; }
            }
            if (lane == 63) { s_wt[0][wid] = inc_t; s_wt[1][wid] = inc_g; }
            __syncthreads();
            unsigned base_t = 0, base_g = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) if (w < wid) { base_t += s_wt[0][w]; base_g += s_wt[1][w]; }
            const unsigned lstart = base_t + inc_t - tcount;
            s_run[tid] = (unsigned)s.seg.off[sg] + base_g + inc_g - gtot + s.hist[(size_t)tid * s.nblocks_max + b] - lstart;
            unsigned run = lstart;
#pragma unroll
            for (int w = 0; w < NW; ++w) { s_cnt[w][tid] = run; run += cw[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            if (rk[j] != 0xffffffffu) {
                const unsigned d = digit_of(s, key[j], p);
                const unsigned l = s_cnt[wid][d] + rk[j];
                s_key[l] = key[j]; if (KV) s_val[l] = val[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; ++w) s_cnt[w][tid] = 0;            // for the next tile (read again only after the next barrier)
#pragma unroll
        for (int j = 0; j < RS_ITEMS; ++j) {
            const int l = j * RS_BS + tid;
            if (l < ntile) {
                const uint64_t k = s_key[l];
                const unsigned pos = s_run[digit_of(s, k, p)] + (unsigned)l;
                KO[pos] = k; if (KV) VO[pos] = s_val[l];
            }
        }
        __syncthreads();
    }
}

// Passes 4..7 (key bits 32..63) in ONE single-workgroup kernel.  Voxel keys, distance keys and composite ranks of
// realistic inputs never have varying bits up there, so this costs one (empty) launch instead of twelve; if a key
// set does need them the sort stays correct, only slower for those passes.
__global__ __launch_bounds__(1024) void rs_high_passes(SortPtrs s) {
    __shared__ unsigned h[256];
    __shared__ unsigned s_cnt[16][256];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int p = 4; p < 8; ++p) {
        int par; if (!pass_runs(s, p, par)) continue;      // uniform
        for (int sg = 0; sg < s.seg.nseg; ++sg) {
            const int n = seg_count(s, sg), so = s.seg.off[sg];
            const uint64_t* K = s.k[par] + so; const uint32_t* V = s.v[par] + so;
            uint64_t* KO = s.k[par ^ 1] + so; uint32_t* VO = s.v[par ^ 1] + so;
            if (tid < 256) h[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += 1024) atomicAdd(&h[digit_of(s, K[i], p)], 1u);
            __syncthreads();
            if (tid == 0) { unsigned run = 0; for (int d = 0; d < 256; ++d) { unsigned c = h[d]; h[d] = run; run += c; } }
            __syncthreads();
            for (int base = 0; base < n; base += 1024) {        // stable: tiles in order, waves in order, lanes in order
                const int i = base + tid;
                const bool valid = i < n;
                uint64_t key = 0; uint32_t val = 0; unsigned d = 0;
                if (valid) { key = K[i]; if (s.kv) val = V[i]; d = digit_of(s, key, p); }
                for (int w = 0; w < 16; ++w) if (tid < 256) s_cnt[w][tid] = 0;
                __syncthreads();
                unsigned long long m = __ballot(valid);
#pragma unroll
                for (int bit = 0; bit < 8; ++bit) {
                    const bool one = (d >> bit) & 1;
                    const unsigned long long bal = __ballot(valid && one);
                    m &= one ? bal : ~bal;
                }
                const int rank = __popcll(m & lt);
                if (valid && rank == 0) s_cnt[wid][d] = (unsigned)__popcll(m);
                __syncthreads();
                if (valid) {
                    unsigned pos = h[d] + rank;
                    for (int w = 0; w < wid; ++w) pos += s_cnt[w][d];
                    KO[pos] = key; if (s.kv) VO[pos] = val;
                }
                __syncthreads();
                if (tid < 256) { unsigned t = 0; for (int w = 0; w < 16; ++w) t += s_cnt[w][tid]; h[tid] += t; }
                __syncthreads();
            }
            __syncthreads();
        }
    }
}

// after the last pass the result may sit in the other buffer: bring it home to the caller's (grid.y = segment)
__global__ __launch_bounds__(RS_BS) void rs_finish(SortPtrs s) {
    const unsigned long long diff = union_diff(s) >> s.shift;
    int par = 0;
    for (int q = 0; q < 8; ++q) par ^= ((diff >> (8 * q)) & 0xffull) != 0;
    if (par == s.home) return;                            // already in the caller's buffer
    const int sg = blockIdx.y, n = seg_count(s, sg), so = s.seg.off[sg];
    for (int i = blockIdx.x * RS_BS + threadIdx.x; i < n; i += gridDim.x * RS_BS) { s.k[s.home][so + i] = s.k[par][so + i]; if (s.kv) s.v[s.home][so + i] = s.v[par][so + i]; }
}

}  // namespace

int RadixSorter::reserve(size_t slots) {
    nblocks_max = (int)((slots + RS_TILE - 1) / RS_TILE) + 1;
    SSDR_TRY(k1.reserve(8 * slots + 16)); SSDR_TRY(v1.reserve(4 * slots + 16));
    SSDR_TRY(hist.reserve(4 * 256 * (size_t)nblocks_max + 4 * 256 * (size_t)RADIX_MAX_SEG)); SSDR_TRY(andor.reserve(16 * RADIX_MAX_SEG * (size_t)(RS_ANDOR_G + 1)));
    return SSDR_OK;
}

int RadixSorter::sort_segments(uint64_t* keys, uint32_t* vals, int nseg, const int* off, const int* n_host, const int* d_cnt, hipStream_t st, int key_bits, bool input_in_alt,
                                const unsigned long long* d_andor, int shift) {
    if (nseg <= 0) return SSDR_OK;
    if (nseg > RADIX_MAX_SEG) { set_error("radix sort: more than %d segments", RADIX_MAX_SEG); return SSDR_ERR_INVALID; }
    const int slots = off[nseg];
    if (slots <= 0) return SSDR_OK;
    SSDR_TRY(reserve((size_t)slots));
    SortPtrs s; s.k[0] = keys; s.k[1] = k1.as<uint64_t>(); s.v[0] = vals; s.v[1] = v1.as<uint32_t>(); s.home = 0;
    if (input_in_alt) { std::swap(s.k[0], s.k[1]); std::swap(s.v[0], s.v[1]); s.home = 1; }     // the producer wrote into alt_keys() / alt_vals()
    s.andor = andor.as<unsigned long long>(); s.andor_part = s.andor + 2 * RADIX_MAX_SEG; s.hist = hist.as<unsigned>(); s.tot = hist.as<unsigned>() + 256 * (size_t)nblocks_max;
    s.d_cnt = d_cnt; s.nblocks_max = nblocks_max; s.seg.nseg = nseg;
    s.shift = shift; s.kv = vals != nullptr;
    if (shift < 0 || shift + key_bits > 64) { set_error("radix sort: key bits [%d, %d) do not fit 64", shift, shift + key_bits); return SSDR_ERR_INVALID; }
    int maxn = 0;
    for (int i = 0; i < nseg; ++i) {
        if (off[i] % RS_TILE || off[i + 1] < off[i] + n_host[i]) { set_error("radix sort: segment offsets must be tile-aligned and hold the segment"); return SSDR_ERR_INVALID; }
        s.seg.off[i] = off[i]; s.seg.n_host[i] = n_host[i]; maxn = std::max(maxn, n_host[i]);
    }
    s.seg.off[nseg] = off[nseg];
    const int nb = (slots + RS_TILE - 1) / RS_TILE;
    const int g = std::max(1, std::min(nb, ctx().num_cu * 8));
    const int gseg = std::max(1, std::min((maxn + RS_BS * 8 - 1) / (RS_BS * 8), ctx().num_cu * 4));
    const int gao = std::max(1, std::min(std::min(gseg, RS_ANDOR_G), std::max(1, ctx().num_cu * 16 / nseg)));
    if (d_andor) SSDR_HIP(hipMemcpyAsync(s.andor, d_andor, 16 * (size_t)nseg, hipMemcpyDeviceToDevice, st));
    else {
        hipLaunchKernelGGL(rs_andor, dim3(gao, nseg), dim3(RS_BS), 0, st, s);
        hipLaunchKernelGGL(rs_andor_fold, dim3(nseg), dim3(64), 0, st, s, gao);
    }
    // wide passes for the low 32 bits; the high ones in one single-workgroup kernel (voxel and distance keys never vary up there) — unless the set is large:
    // the ranking of an AL round's 123 k regions (float bits above the index) spent 1.14 ms in that kernel, 4 x 22 us as wide passes
    const int all = (std::max(key_bits, 1) + 7) / 8;
    const bool wide_high = this->wide_high && key_bits > 32 && maxn >= 16384;
    const int npass = wide_high ? all : std::min(4, all);      // (maxn >= 16384: below, the one kernel is as fast as twelve launches)
    for (int p = 0; p < npass; ++p) {
        hipLaunchKernelGGL(rs_hist, dim3(g), dim3(RS_BS), 0, st, s, p);
        hipLaunchKernelGGL(rs_scan, dim3(256, nseg), dim3(RS_BS), 0, st, s, p);
        if (s.kv) hipLaunchKernelGGL(rs_scatter<true>, dim3(g), dim3(RS_BS), 0, st, s, p);
        else hipLaunchKernelGGL(rs_scatter<false>, dim3(g), dim3(RS_BS), 0, st, s, p);
    }
    if (key_bits > 32 && !wide_high) hipLaunchKernelGGL(rs_high_passes, dim3(1), dim3(1024), 0, st, s);   // ... one kernel for the rest
    hipLaunchKernelGGL(rs_finish, dim3(std::min(gseg, std::max(1, 2048 / nseg)), nseg), dim3(RS_BS), 0, st, s);     // usually nothing to copy: keep the (empty) launch small
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int RadixSorter::sort(uint64_t* keys, uint32_t* vals, int n_host, const int* d_n, hipStream_t st, int key_bits, bool input_in_alt) {
    if (n_host <= 0) return SSDR_OK;
    // one segment; the tail of its last tile is never touched, so a buffer of exactly n_host slots is enough
    const int off[2] = {0, (n_host + RS_TILE - 1) / RS_TILE * RS_TILE};
    return sort_segments(keys, vals, 1, off, &n_host, d_n, st, key_bits, input_in_alt);
}

}  // namespace ssdr
