// RandLA-Net matrix products on the bf16 matrix cores of gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// Two arithmetic modes (randla.hpp): PREC_BF16X3 carries every fp32 operand as two bf16 pieces (hi, lo) and evaluates
// hi*hi + lo*hi + hi*lo — products exact in fp32, operands good to 2^-16 — and PREC_BF16 rounds the operands to bf16 once
// (BASELINE configuration 3).  The exact-f32 kernels of randla_kernels.hip stay the reference arithmetic (PREC_F32).
//
//   dense_bf16_kernel   per-point 1x1 convs (helper_tf_util.py:111-166 with BN folded; RandLANet.py:506-512, :159-172):
//                       activations are split while they are staged into LDS, weights arrive pre-split and transposed.
//   lfa_bf16_kernel     the K-expanded half of building_block (RandLANet.py:514-527, :572-585) for d >= 64: relative
//                       position encoding and LocSE conv as in lfa_att_kernel (K = 10, exact f32 MFMA), the LFAmlp2 conv and
//                       the position half of the attention product on the bf16 cores; the neighbour half arrives as gathered
//                       rows of G = f W[0:h] (one dense launch per point instead of per neighbour row), softmax over the 16
//                       neighbours and the weighted sum in the accumulator layout.
#include "ssdr_internal.hpp"
#include "randla.hpp"
#include "randla_dev.hpp"
#include <cmath>

namespace ssdr {

// 16-byte LDS / global accesses of packed bf16 fragments
__device__ __forceinline__ u32x4 ld128(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st128(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

template <int TERMS>
__device__ __forceinline__ f32x4 mma_split(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x4 c) {
    c = mfma_bf16(a[0], b[0], c);
    if (TERMS == 2) { c = mfma_bf16(a[1], b[0], c); c = mfma_bf16(a[0], b[1], c); }
    return c;
}

// ---- dense ------------------------------------------------------------------------------------------------------
// TM x 64 output tile per workgroup of 4 waves, K in chunks of KC.  TM = 128: wave w owns rows [32w, 32w+32) x 64 columns
// (2 x 4 MFMA tiles); TM = 32 (layers with few rows: 4x the workgroups): wave w owns columns [16w, 16w+16) of both row tiles.
// LDS rows hold KC bf16 (+8 of padding: 16-byte aligned rows whose 16-byte slots rotate through the banks); the next chunk
// is fetched into registers while the current one is multiplied.
// (register budget: one more wave per SIMD than the allocator's own choice — 4 instead of 3 for the 128-row tiles, 5 instead of 4 for the 32-row tiles, no
// spills: 372 -> 344 and 301 -> 277 us per step; the same squeeze on the LFA kernels spills and costs 15-30 %)
template <int TM, int KC, int TERMS, bool VEC>
__global__ __launch_bounds__(256) SSDR_WAVES_PER_EU(TM == 128 ? 4 : 5) void dense_bf16_kernel(DenseArgs a) {
    constexpr int KS = KC + 8;                         // LDS row stride in bf16 elements
    constexpr int TPR = KC / 8;                        // threads per staged row (8 k each)
    constexpr int RPP = 256 / TPR;                     // rows staged per pass
    constexpr int APASS = TM / RPP, BPASS = 64 / RPP;
    static_assert(TM % RPP == 0 && 64 % RPP == 0, "staging passes");
    constexpr int CT = TM == 128 ? 4 : 1;              // column tiles per wave (2 row tiles each)
    __shared__ __attribute__((aligned(16))) uint16_t As[TERMS][TM * KS];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[TERMS][64 * KS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bxr = (int)blockIdx.x, byc = (int)blockIdx.y;
    if (a.idx2 && a.m_per_batch % TM == 0 && a.M % a.m_per_batch == 0 && ((a.M / a.m_per_batch) & 7) == 0) {
        // decoder layers gather rows of the coarser level's table of the row's batch element: a batch element's row blocks (all their column
        // blocks) to one XCD (block_prims.hpp, xcd_tile_map: blocks b and b + 8 share an XCD), so the table is fetched into one L2 instead of eight
        const int gy = (int)gridDim.y, rpb = a.m_per_batch / TM;
        const int lin = byc * (int)gridDim.x + bxr, xcd = lin & 7, j = lin >> 3, rb = j / gy;
        byc = j % gy; bxr = (xcd + 8 * (rb / rpb)) * rpb + rb % rpb;
    }
    const int row0 = bxr * TM, col0 = byc * 64;
    const int K = a.k1 + a.k2;
    const int sr = tid / TPR, sk = (tid % TPR) * 8;
    const float* x1r[APASS]; const float* x2r[APASS];
#pragma unroll
    for (int h = 0; h < APASS; ++h) {
        const int grow = row0 + sr + RPP * h;
        x1r[h] = nullptr; x2r[h] = nullptr;
        if (grow < a.M) {
            x1r[h] = a.x1 + (size_t)grow * a.k1;
            if (a.k2) {
                size_t r2 = (size_t)grow;
                if (a.idx2) r2 = (size_t)(grow / a.m_per_batch) * a.x2_rows_per_batch + (size_t)a.idx2[grow];
                x2r[h] = a.x2 + r2 * a.k2;
            }
        }
    }
    float ra[APASS][8]; u32x4 rb[BPASS][TERMS];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int h = 0; h < APASS; ++h) {
            if (VEC) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int gk = kc + sk + 4 * q;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (x1r[h] && gk < K) v = (gk < a.k1) ? *reinterpret_cast<const float4*>(x1r[h] + gk) : *reinterpret_cast<const float4*>(x2r[h] + (gk - a.k1));
                    ra[h][4 * q] = v.x; ra[h][4 * q + 1] = v.y; ra[h][4 * q + 2] = v.z; ra[h][4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int gk = kc + sk + j;
                    float v = 0.f;
                    if (x1r[h]) { if (gk < a.k1) v = x1r[h][gk]; else if (gk < K) v = x2r[h][gk - a.k1]; }
                    ra[h][j] = v;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < BPASS; ++h) {
            const int gc = col0 + sr + RPP * h;
#pragma unroll
            for (int t = 0; t < TERMS; ++t) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (gc < a.N && kc + sk < a.kp) v = ld128((t ? a.wt_lo : a.wt_hi) + (size_t)gc * a.kp + kc + sk);     // rows are padded to kp (a multiple of 64)
                rb[h][t] = v;
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int h = 0; h < APASS; ++h) {
            u32x4 hi, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned hq, lq;
                split_bf16(ra[h][2 * q], ra[h][2 * q + 1], hq, lq);
                hi[q] = hq; lo[q] = lq;
            }
            st128(&As[0][(sr + RPP * h) * KS + sk], hi);
            if (TERMS == 2) st128(&As[TERMS - 1][(sr + RPP * h) * KS + sk], lo);
        }
#pragma unroll
        for (int h = 0; h < BPASS; ++h)
#pragma unroll
            for (int t = 0; t < TERMS; ++t) st128(&Bs[t][(sr + RPP * h) * KS + sk], rb[h][t]);
    };
    f32x4 acc[2][CT];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rbase = TM == 128 ? w * 32 : 0, cbase = TM == 128 ? 0 : w * 16;
    const int fr = lane & 15, fk = 8 * (lane >> 4);
    fetch(0);
    for (int kc = 0; kc < K; kc += KC) {
        stash();
        __syncthreads();
        if (kc + KC < K) fetch(kc + KC);          // in flight while the MFMAs below run
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) {
            u32x4 af[2][2], bf[CT][2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int t = 0; t < TERMS; ++t) af[r][t] = ld128(&As[t][(rbase + r * 16 + fr) * KS + ks * 32 + fk]);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int t = 0; t < TERMS; ++t) bf[c][t] = ld128(&Bs[t][(cbase + c * 16 + fr) * KS + ks * 32 + fk]);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[r][c] = mma_split<TERMS>(af[r], bf[c], acc[r][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = col0 + cbase + c * 16 + (lane & 15);
        if (col >= a.N) continue;
        const float bias = a.b ? a.b[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = row0 + rbase + r * 16 + (lane >> 4) * 4 + q;
                if (row < a.M) { float v = acc[r][c][q] + bias; a.y[(size_t)row * (a.ldy ? a.ldy : a.N) + col] = a.act ? lrelu(v) : v; }
            }
    }
}


// ---- att_pooling's mlp chained into lrelu(mlp2 + shortcut) (RandLANet.py:583-585, :510-512) at the wide-row levels (d <= 64) ------------------
// y1 = lrelu(x1 Wa + ba) (d -> d), y = lrelu([y1 | x2] Wb + bb) (d + k2 -> 2 d) for a 128-row tile per workgroup: both layers stream their rows from HBM
// (655 k / 164 k rows at levels 0 / 1: 42 MB in and out for the first, 63 in and 84 out for the second), and the d-channel intermediate between them
// now stays in LDS (58 -> 33 us at level 0, 65.6 -> 48.6 at level 1): the A image [x1 | x2 | 0] (bf16 pieces, rows of K2P + 8) is staged ONCE, the first product overwrites its x1 columns with y1 (a wave
// reads and writes its own 32 rows only), and the second runs over the whole K of that image, one 64-column block of Wb at a time.  One launch and
// 84 MB of traffic less per level.
struct ChainArgs {
    const float* x1; const float* x2; int k2;
    const uint16_t* wa_hi; const uint16_t* wa_lo; int kpa; const float* ba;
    const uint16_t* wb_hi; const uint16_t* wb_lo; int kpb; const float* bb;
    float* y; int M;
};
template <int D, int K2P, int TM, int TERMS>
__global__ __launch_bounds__(256) void dense_chain_kernel(ChainArgs a) {
    constexpr int KS = K2P + 8;                        // LDS row stride in bf16 elements
    constexpr int TPR = K2P <= 32 ? 4 : 16;            // threads per staged row (8 k each), a power of two
    constexpr int RPP = 256 / TPR;                     // rows per staging pass
    // TM = 128: wave w owns rows [32 w, 32 w + 32) and every column; TM = 64 (more, smaller workgroups: 53 KB of LDS instead of 79 at d = 64): waves 2 x 2,
    // wave (wr, wc) owns rows [32 wr, +32) and half of the column tiles
    constexpr int WC = TM == 128 ? 1 : 2;              // waves across the columns
    constexpr int CT1 = (D / 16 + WC - 1) / WC;        // 16-column tiles per wave of the first product
    constexpr int CT2 = 4 / WC;                        // ... of a 64-column block of the second
    static_assert(D % 16 == 0 && D <= 64 && K2P % 32 == 0 && TPR * 8 >= K2P && (TM == 128 || TM == 64) && TM % RPP == 0, "shapes");
    __shared__ __attribute__((aligned(16))) uint16_t As[TERMS][TM * KS];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[TERMS][64 * KS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = TM == 128 ? w : (w >> 1), wc = TM == 128 ? 0 : (w & 1);
    const int row0 = (int)blockIdx.x * TM;
    const int sr = tid / TPR, sk = (tid % TPR) * 8;
    const int K = D + a.k2;
    // ---- A image: [x1 | x2 | 0], split into bf16 pieces
    if (sk < K2P) {
        float4 va[TM / RPP][2];          // every load of the tile in flight before the first conversion
#pragma unroll
        for (int h = 0; h < TM / RPP; ++h) {
            const int grow = row0 + sr + RPP * h;
            va[h][0] = make_float4(0.f, 0.f, 0.f, 0.f); va[h][1] = va[h][0];
            if (grow < a.M) {
                const float* src = sk < D ? a.x1 + (size_t)grow * D + sk : (sk < K ? a.x2 + (size_t)grow * a.k2 + (sk - D) : nullptr);
                if (src) { va[h][0] = *reinterpret_cast<const float4*>(src); va[h][1] = *reinterpret_cast<const float4*>(src + 4); }
            }
        }
#pragma unroll
        for (int h = 0; h < TM / RPP; ++h) {
            const int r = sr + RPP * h;
            const float4 v0 = va[h][0], v1 = va[h][1];
            u32x4 hi, lo; unsigned hq, lq;
            split_bf16(v0.x, v0.y, hq, lq); hi[0] = hq; lo[0] = lq;
            split_bf16(v0.z, v0.w, hq, lq); hi[1] = hq; lo[1] = lq;
            split_bf16(v1.x, v1.y, hq, lq); hi[2] = hq; lo[2] = lq;
            split_bf16(v1.z, v1.w, hq, lq); hi[3] = hq; lo[3] = lq;
            st128(&As[0][r * KS + sk], hi);
            if (TERMS == 2) st128(&As[TERMS - 1][r * KS + sk], lo);
        }
    }
    // a 64-row block of transposed weights (rows = output columns), zero beyond the layer's columns and its K
    u32x4 wreg[64 / RPP][TERMS];          // the next weight block travels in registers while the current one is multiplied
    auto load_w = [&](const uint16_t* whi, const uint16_t* wlo, int kp, int col0, int ncols, int kvalid) {
#pragma unroll
        for (int h = 0; h < 64 / RPP; ++h) {
            const int r = sr + RPP * h;
#pragma unroll
            for (int t = 0; t < TERMS; ++t) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (sk < K2P && col0 + r < ncols && sk < kvalid) v = ld128((t ? wlo : whi) + (size_t)(col0 + r) * kp + sk);
                wreg[h][t] = v;
            }
        }
    };
    auto store_w = [&]() {
        if (sk < K2P) {
#pragma unroll
            for (int h = 0; h < 64 / RPP; ++h)
#pragma unroll
                for (int t = 0; t < TERMS; ++t) st128(&Bs[t][(sr + RPP * h) * KS + sk], wreg[h][t]);
        }
    };
    load_w(a.wa_hi, a.wa_lo, a.kpa, 0, D, D);
    store_w();
    __syncthreads();
    load_w(a.wb_hi, a.wb_lo, a.kpb, 0, 2 * D, K);          // in flight across the first product
    const int fr = lane & 15, fk = 8 * (lane >> 4), rbase = wr * 32;
    // ---- y1 = lrelu(x1 Wa + ba) over the wave's 32 rows (its column tiles), written over the x1 columns of the image
    {
        f32x4 acc[2][CT1];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < CT1; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool has1 = wc * CT1 * 16 < D;          // (d = 16 on 2 x 2 waves: the second column of waves has no tile of the first product)
        if (has1) {
#pragma unroll
            for (int ks = 0; ks < (D + 31) / 32; ++ks) {
                u32x4 af[2][2], bf[CT1][2];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) af[r][t] = ld128(&As[t][(rbase + r * 16 + fr) * KS + ks * 32 + fk]);
#pragma unroll
                for (int c = 0; c < CT1; ++c)
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) bf[c][t] = ld128(&Bs[t][((wc * CT1 + c) * 16 + fr) * KS + ks * 32 + fk]);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < CT1; ++c) acc[r][c] = mma_split<TERMS>(af[r], bf[c], acc[r][c]);
            }
        }
        if (WC > 1) __syncthreads();          // the other wave of these rows has read x1 before y1 goes over it (one wave per row block otherwise: its own accesses are in order)
        if (has1) {
#pragma unroll
            for (int c = 0; c < CT1; ++c) {
                const int col = (wc * CT1 + c) * 16 + (lane & 15);
                const float bias = a.ba ? a.ba[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = rbase + r * 16 + (lane >> 4) * 4 + q;
                        const float v = lrelu(acc[r][c][q] + bias);
                        unsigned hq, lq; split_bf16(v, 0.f, hq, lq);
                        As[0][row * KS + col] = (uint16_t)(hq & 0xffffu);
                        if (TERMS == 2) As[TERMS - 1][row * KS + col] = (uint16_t)(lq & 0xffffu);
                    }
            }
        }
    }
    // ---- y = lrelu([y1 | x2] Wb + bb), 64 columns at a time
    for (int ct = 0; ct < (2 * D + 63) / 64; ++ct) {
        __syncthreads();             // every wave is done with the weight block in Bs (and, the first time, has written its y1 tiles)
        store_w();
        __syncthreads();
        if (ct + 1 < (2 * D + 63) / 64) load_w(a.wb_hi, a.wb_lo, a.kpb, (ct + 1) * 64, 2 * D, K);
        f32x4 acc[2][CT2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < CT2; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < K2P / 32; ++ks) {
            u32x4 af[2][2], bf[CT2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int t = 0; t < TERMS; ++t) af[r][t] = ld128(&As[t][(rbase + r * 16 + fr) * KS + ks * 32 + fk]);
#pragma unroll
            for (int c = 0; c < CT2; ++c)
#pragma unroll
                for (int t = 0; t < TERMS; ++t) bf[c][t] = ld128(&Bs[t][((wc * CT2 + c) * 16 + fr) * KS + ks * 32 + fk]);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < CT2; ++c) acc[r][c] = mma_split<TERMS>(af[r], bf[c], acc[r][c]);
        }
#pragma unroll
        for (int c = 0; c < CT2; ++c) {
            const int col = ct * 64 + (wc * CT2 + c) * 16 + (lane & 15);
            if (col >= 2 * D) continue;
            const float bias = a.bb ? a.bb[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = row0 + rbase + r * 16 + (lane >> 4) * 4 + q;
                    if (row < a.M) a.y[(size_t)row * (2 * D) + col] = lrelu(acc[r][c][q] + bias);
                }
        }
    }
}

// ---- fused local-feature-aggregation attention half, d >= 64 -------------------------------------------------------------
// Work split: NW waves = NCG column groups x NPG point groups.  A wave owns NCH column tiles of the neighbour-feature half and
// the NCH matching tiles of the position half, for PW of the workgroup's PTS points.
//
// Both MFMA operand fragments of a 16-wide tile have the same shape (lane = (index, k group)), so one pair of registers gives
// the product in either orientation: mfma(A = x, B = w) has the neighbour row in the accumulator registers and the channel on the
// lane (what softmax / weighted sum need), mfma(A = w, B = x) has four consecutive channels of one neighbour row per lane (what
// the next product's A operand needs: one 8-byte LDS store per piece instead of eight 2-byte scatters).  The position
// encoding chain is therefore computed in both orientations: the transposed one feeds LDS, the other stays in registers
// until the weighted sum.
template <int D> struct LfaBf16Cfg {
    static constexpr int H = D / 2;
    static constexpr int NW = D >= 256 ? 8 : 4;                       // waves per workgroup
    static constexpr int NT = NW * 64;
    static constexpr int NCG = D == 64 ? 2 : NW;                      // column groups
    static constexpr int NPG = NW / NCG;                              // point groups
    static constexpr int PTS = D == 64 ? 8 : 4;                            // points per workgroup (a wave keeps 4 x PW x 4 NCH accumulator / feature registers)
    static constexpr int PW = PTS / NPG;                              // points per wave
    static constexpr int ROWS = PTS * 16;
    static constexpr int LDX = H + 8;                                 // bf16 elements per LDS row of a piece (16-byte aligned, rotating bank slots)
    static constexpr int NCT1 = H / 16;                               // column tiles per half
    static constexpr int NCH = NCT1 / NCG;                            // ... per wave and half
    static_assert(NCT1 % NCG == 0 && PTS % NPG == 0, "tile split");
    static constexpr size_t lds_bytes(int terms) { return (size_t)ROWS * ((size_t)terms * LDX * 2 + 10 * 4 + 4); }
};

#ifndef HIPEMU
// raw buffer loads: 32-bit per-lane byte offset + immediate, one VALU op of address arithmetic per gathered row
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); }
template <int IMM> __device__ __forceinline__ float buf_load(rsrc_t r, unsigned voff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff + IMM, 0, 0)); }
#else
struct rsrc_t { const char* p; };
static inline rsrc_t make_rsrc(const void* p, unsigned) { return rsrc_t{reinterpret_cast<const char*>(p)}; }
template <int IMM> static inline float buf_load(rsrc_t r, unsigned voff) { return *reinterpret_cast<const float*>(r.p + voff + IMM); }
#endif

template <int D, bool SECOND, int TERMS>
__global__ __launch_bounds__(LfaBf16Cfg<D>::NT) SSDR_WAVES_PER_EU((D == 256 && !SECOND) ? 4 : 1) void lfa_bf16_kernel(LfaArgs a) {      // (d = 256, first half: 127 registers instead of 166, 4 waves per SIMD: 90 -> 74 us)
    using C = LfaBf16Cfg<D>;
    constexpr int H = C::H, PTS = C::PTS, ROWS = C::ROWS, LDX = C::LDX, NT = C::NT, NCG = C::NCG, PW = C::PW, NCH = C::NCH, NCT1 = C::NCT1;
    SSDR_DYN_SHARED(float, smem);
    uint16_t* X[2];                                // f_xyz as bf16 pieces [ROWS][LDX]
    X[0] = reinterpret_cast<uint16_t*>(smem);
    X[1] = X[0] + (TERMS == 2 ? (size_t)ROWS * LDX : 0);
    float* REL = reinterpret_cast<float*>(X[0] + (size_t)TERMS * ROWS * LDX);      // [ROWS][10]
    int* NBR = reinterpret_cast<int*>(REL + (size_t)ROWS * 10);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lc = lane & 15, lg = lane >> 4;
    const int cg = w % NCG, p0 = (w / NCG) * PW;   // this wave's column group and first point
    int bx_, b; xcd_tile_map(bx_, b);
    const int pt0 = bx_ * PTS;
    const float* xyz = a.xyz + (size_t)b * a.xyz_batch_stride;
    const int* neigh = a.neigh + (size_t)b * a.n * 16;

    // LocSE weights of this wave's column tiles: one fragment register per k step serves both orientations
    float w1[NCH][3], b1c[NCH]; float4 b1r[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int ct = cg * NCH + j;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int k = ks * 4 + lg;
            const float v = a.w_l1[(k < 10 ? k : 9) * H + ct * 16 + lc];
            w1[j][ks] = k < 10 ? v : 0.f;
        }
        b1c[j] = a.b_l1[ct * 16 + lc];                                           // channel on the lane
        b1r[j] = *reinterpret_cast<const float4*>(a.b_l1 + ct * 16 + 4 * lg);    // channels in the registers
    }

    // relative_pos_encoding (:529-535): [ |d|, d(3), p(3), p_nbr(3) ]
    for (int row = tid; row < ROWS; row += NT) {
        const int n = pt0 + (row >> 4);
        float r[10]; int j = 0;
#pragma unroll
        for (int q = 0; q < 10; ++q) r[q] = 0.f;
        if (n < a.n) {
            j = neigh[(size_t)n * 16 + (row & 15)];
            const float px = xyz[3 * (size_t)n], py = xyz[3 * (size_t)n + 1], pz = xyz[3 * (size_t)n + 2];
            const float qx = xyz[3 * (size_t)j], qy = xyz[3 * (size_t)j + 1], qz = xyz[3 * (size_t)j + 2];
            const float dx = px - qx, dy = py - qy, dz = pz - qz;
            r[0] = sqrtf(dx * dx + dy * dy + dz * dz);
            r[1] = dx; r[2] = dy; r[3] = dz; r[4] = px; r[5] = py; r[6] = pz; r[7] = qx; r[8] = qy; r[9] = qz;
        }
#pragma unroll
        for (int q = 0; q < 10; ++q) REL[row * 10 + q] = r[q];
        NBR[row] = j;
    }
    __syncthreads();

    // four consecutive channels (ct*16 + 4 lg ..) of neighbour row lc of point p as bf16 pieces: one 8-byte store per piece
    auto store_row4 = [&](int p, int ct, const float (&v)[4]) {
        u32x2 hi, lo; unsigned h, l;
        split_bf16(v[0], v[1], h, l); hi[0] = h; lo[0] = l;
        split_bf16(v[2], v[3], h, l); hi[1] = h; lo[1] = l;
        const int e = (p * 16 + lc) * LDX + ct * 16 + 4 * lg;
        *reinterpret_cast<u32x2*>(&X[0][e]) = hi;
        if (TERMS == 2) *reinterpret_cast<u32x2*>(&X[1][e]) = lo;
    };

    // f_xyz of this wave's (point, position-column tile) pairs in the accumulator layout, kept for the weighted sum
    float fx[PW][NCH][4];

    // LocSE conv 10 -> H (LFAmlp1, :518): K = 10 stays on the exact f32 MFMA (3 steps of 4), bias in the accumulator
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        float rf[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { const int k = ks * 4 + lg; rf[ks] = (k < 10) ? REL[((p0 + p) * 16 + lc) * 10 + k] : 0.f; }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            f32x4 at = f32x4{b1r[j].x, b1r[j].y, b1r[j].z, b1r[j].w};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) at = mfma16(w1[j][ks], rf[ks], at);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = lrelu(at[r]);
            store_row4(p0 + p, cg * NCH + j, v);
            if (!SECOND) {
                f32x4 ac = f32x4{b1c[j], b1c[j], b1c[j], b1c[j]};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) ac = mfma16(rf[ks], w1[j][ks], ac);
#pragma unroll
                for (int r = 0; r < 4; ++r) fx[p][j][r] = lrelu(ac[r]);
            }
        }
    }
    __syncthreads();

    if (SECOND) {
        // f_xyz <- lrelu(f_xyz W2 + b2) (LFAmlp2, :523), both orientations from the same fragments; in place in LDS: every
        // product is finished before the first store
        f32x4 accT[PW][NCH], acc2[PW][NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int ct = cg * NCH + j;
            const float4 br = *reinterpret_cast<const float4*>(a.b_l2 + ct * 16 + 4 * lg);
            const float bc = a.b_l2[ct * 16 + lc];
#pragma unroll
            for (int p = 0; p < PW; ++p) { accT[p][j] = f32x4{br.x, br.y, br.z, br.w}; acc2[p][j] = f32x4{bc, bc, bc, bc}; }
        }
        for (int kb = 0; kb < H / 32; ++kb) {
            u32x4 xf[PW][2], wf[NCH][2];
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int t = 0; t < TERMS; ++t) wf[j][t] = ld128((t ? a.l2_lo : a.l2_hi) + (size_t)((cg * NCH + j) * 16 + lc) * a.kp2 + kb * 32 + 8 * lg);
#pragma unroll
            for (int p = 0; p < PW; ++p)
#pragma unroll
                for (int t = 0; t < TERMS; ++t) xf[p][t] = ld128(&X[t][((p0 + p) * 16 + lc) * LDX + kb * 32 + 8 * lg]);
#pragma unroll
            for (int p = 0; p < PW; ++p)
#pragma unroll
                for (int j = 0; j < NCH; ++j) { accT[p][j] = mma_split<TERMS>(wf[j], xf[p], accT[p][j]); acc2[p][j] = mma_split<TERMS>(xf[p], wf[j], acc2[p][j]); }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PW; ++p)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = lrelu(accT[p][j][r]); fx[p][j][r] = lrelu(acc2[p][j][r]); }
                store_row4(p0 + p, cg * NCH + j, v);
            }
        __syncthreads();
    }

    // attention scores S = [f_nbr | f_xyz] Wfc (:578), pre-scaled by log2(e) through the weight pieces.  The neighbour half is
    // the gathered row of G = f Wfc[0:H] (dense launch, one row per point); the position half is multiplied here: A = f_xyz
    // pieces from LDS, B = rows H..D of Wfc (transposed pieces).  Column tile c < NCH: neighbour-feature columns, else position columns.
    const rsrc_t rG = make_rsrc(a.g + (size_t)b * a.n * D, (unsigned)a.n * D * 4u);
    const rsrc_t rF = make_rsrc(a.fin + (size_t)b * a.n * H, (unsigned)a.n * H * 4u);
    f32x4 acc[PW][2 * NCH];
    float fn[PW][NCH][4];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        int nb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) nb[r] = NBR[(p0 + p) * 16 + lg * 4 + r];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned og = (unsigned)nb[r] * (D * 4u) + lc * 4u + cg * NCH * 64u, of = (unsigned)nb[r] * (H * 4u) + lc * 4u + cg * NCH * 64u;
            acc[p][0][r] = buf_load<0>(rG, og); acc[p][NCH][r] = buf_load<H * 4>(rG, og); fn[p][0][r] = buf_load<0>(rF, of);
            if constexpr (NCH == 2) { acc[p][1][r] = buf_load<64>(rG, og); acc[p][NCH + 1][r] = buf_load<H * 4 + 64>(rG, og); fn[p][1][r] = buf_load<64>(rF, of); }
        }
    }
    static_assert(NCH <= 2, "gather immediates are written out for one or two tiles per half");
    for (int kb = 0; kb < H / 32; ++kb) {
        u32x4 af[PW][2], bf[2 * NCH][2];
#pragma unroll
        for (int c = 0; c < 2 * NCH; ++c) {
            const int col = (c < NCH ? cg * NCH + c : NCT1 + cg * NCH + (c - NCH)) * 16 + lc;
#pragma unroll
            for (int t = 0; t < TERMS; ++t) bf[c][t] = ld128((t ? a.fc_lo : a.fc_hi) + (size_t)col * D + H + kb * 32 + 8 * lg);
        }
#pragma unroll
        for (int p = 0; p < PW; ++p)
#pragma unroll
            for (int t = 0; t < TERMS; ++t) af[p][t] = ld128(&X[t][((p0 + p) * 16 + lc) * LDX + kb * 32 + 8 * lg]);
#pragma unroll
        for (int p = 0; p < PW; ++p)
#pragma unroll
            for (int c = 0; c < 2 * NCH; ++c) acc[p][c] = mma_split<TERMS>(af[p], bf[c], acc[p][c]);
    }

    // softmax over the 16 neighbours (:579) and weighted sum (:580-581): out = sum_k f_k 2^(s_k - m) / sum_k 2^(s_k - m)
    float* out = a.out + (size_t)b * a.n * D;
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int n = pt0 + p0 + p;
#pragma unroll
        for (int c = 0; c < 2 * NCH; ++c) {
            const int col = (c < NCH ? cg * NCH + c : NCT1 + cg * NCH + (c - NCH)) * 16 + lc;
            float m = fmaxf(fmaxf(acc[p][c][0], acc[p][c][1]), fmaxf(acc[p][c][2], acc[p][c][3]));
            m = rows_max(m);
            float s = 0.f, v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = fast_exp2(acc[p][c][r] - m);
                const float f = c < NCH ? fn[p][c][r] : fx[p][c - NCH][r];
                s += e; v = fmaf(f, e, v);
            }
            s = rows_sum(s); v = rows_sum(v);
            v = v * fast_rcp(s);
            if (lg == 0 && n < a.n) out[(size_t)n * D + col] = v;
        }
    }
}

// ---- fc1 (32 -> 64) + fc2 (64 -> 32 = last_second_features) + fc (32 -> C) + softmax (RandLANet.py:174-178, :84) ----------------
// One wave per 16 points, no LDS, no barriers: the point rows arrive in the MFMA operand layout straight from global memory,
// and each layer is computed TRANSPOSED (A = weight fragment, B = activation fragment), which leaves four consecutive output
// channels of one point in a lane — exactly the next layer's operand shape once the k slots of that layer are numbered to match
// (slot (g, j) of a 32-wide k block = channel 4g + j of the block's first 16-channel tile for j < 4, of its second tile for j >= 4;
// the weight fragments are fetched in the same numbering: two 8-byte loads per lane instead of one 16-byte load).  fc2 is also
// computed in the other orientation for the coalesced fp32 store of last_second_features.  The class axis of the logits ends up in
// the accumulator registers x 4 lane rows: softmax as in the LFA kernels.
//
// PRE: the last decoder layer (conv2d_transpose over concat[skip, nearest_interpolation(feature)], 32 + 32 -> 32, RandLANet.py:165-172) runs
// in front of fc1 in the same way, so its output never travels through HBM (84 MB written and read back per step otherwise).
struct TailPre {          // x = lrelu([skip | gathered] Wd + bd)
    const float* skip; const float* up; const int* idx; int m_per_batch, up_rows_per_batch;
    const uint16_t *wh, *wl; int kp; const float* b;
};
template <int TERMS, int C, bool PRE>
__global__ __launch_bounds__(256) void tail_bf16_kernel(TailPre pre, const float* __restrict__ x, const uint16_t* __restrict__ w1h, const uint16_t* __restrict__ w1l, int kp1,
                                                        const float* __restrict__ b1, const uint16_t* __restrict__ w2h, const uint16_t* __restrict__ w2l, int kp2,
                                                        const float* __restrict__ b2, const uint16_t* __restrict__ w3h, const uint16_t* __restrict__ w3l, int kp3,
                                                        const float* __restrict__ b3, int M, float* __restrict__ feat32, float* __restrict__ probs) {
    const int lane = threadIdx.x & 63, lc = lane & 15, lg = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    auto ld64x2 = [](const uint16_t* p0, const uint16_t* p1) {       // two 4-element pieces -> one 8-element fragment
        const u32x2 a = *reinterpret_cast<const u32x2*>(p0), b = *reinterpret_cast<const u32x2*>(p1);
        return u32x4{a[0], a[1], b[0], b[1]};
    };
    // weight fragments (both pieces) in the operand layout, ONE copy per workgroup in LDS (round 4: 104 registers per lane held them for every tile of the
    // wave — two waves per SIMD; fetched per use from LDS the waves are three or four): slot = fragment * 2 + piece, 64 lanes x 16 bytes each
    constexpr int F_WD = 0, F_W1 = 4, F_W2 = 8, F_W3 = 12, NFRAG = 13;
    __shared__ u32x4 s_w[NFRAG * 2][64];
    float4 BD[2];
    {
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int t = 0; t < TERMS; ++t) {
            const uint16_t* p1 = t ? w1l : w1h; const uint16_t* p2 = t ? w2l : w2h; const uint16_t* p3 = t ? w3l : w3h;
            if (PRE && wv == 0) {
                const uint16_t* pd = t ? pre.wl : pre.wh;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) s_w[(F_WD + 2 * kb + ct) * 2 + t][lane] = ld128(pd + (size_t)(16 * ct + lc) * pre.kp + 32 * kb + 8 * lg);      // k block 0: skip channels, 1: gathered channels
            }
            if (wv == 1) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)      // fc1: k = input channel; natural order when the input comes from memory, the transposed numbering behind the decoder layer
                    s_w[(F_W1 + ct) * 2 + t][lane] = PRE ? ld64x2(p1 + (size_t)(16 * ct + lc) * kp1 + 4 * lg, p1 + (size_t)(16 * ct + lc) * kp1 + 16 + 4 * lg) : ld128(p1 + (size_t)(16 * ct + lc) * kp1 + 8 * lg);
            }
            if (wv == 2) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) s_w[(F_W2 + 2 * kb + ct) * 2 + t][lane] = ld64x2(p2 + (size_t)(16 * ct + lc) * kp2 + 32 * kb + 4 * lg, p2 + (size_t)(16 * ct + lc) * kp2 + 32 * kb + 16 + 4 * lg);
            }
            if (wv == 3) s_w[F_W3 * 2 + t][lane] = lc < C ? ld64x2(p3 + (size_t)lc * kp3 + 4 * lg, p3 + (size_t)lc * kp3 + 16 + 4 * lg) : u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
    }
    auto wfr = [&](int f, u32x4 (&w)[2]) { w[0] = s_w[f * 2][lane]; w[1] = s_w[f * 2 + (TERMS - 1)][lane]; };
    if (PRE) { BD[0] = *reinterpret_cast<const float4*>(pre.b + 4 * lg); BD[1] = *reinterpret_cast<const float4*>(pre.b + 16 + 4 * lg); }
    float4 B1[4], B2t[2], B3; float B2c[2];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) B1[ct] = *reinterpret_cast<const float4*>(b1 + 16 * ct + 4 * lg);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) { B2t[ct] = *reinterpret_cast<const float4*>(b2 + 16 * ct + 4 * lg); B2c[ct] = b2[16 * ct + lc]; }
    {
        float t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = 4 * lg + r < C ? b3[4 * lg + r] : 0.f;
        B3 = make_float4(t[0], t[1], t[2], t[3]);
    }
    auto frag = [](const float (&v)[8], u32x4 (&f)[2]) {            // eight fp32 values -> hi / lo fragments
#pragma unroll
        for (int q = 0; q < 4; ++q) { unsigned h, l; split_bf16(v[2 * q], v[2 * q + 1], h, l); f[0][q] = h; f[1][q] = l; }
    };
    const int ntiles = (M + 15) / 16;
    // PRE gathers rows of the coarser level's table of the row's batch element: give a batch element's tiles to one XCD (block_prims.hpp,
    // xcd_tile_map: blocks b and b + 8 share an XCD) when the shapes allow, so that table is fetched into one L2 instead of eight
    int tpb = 0, nbe = 0;
    if (PRE && pre.m_per_batch % 16 == 0 && M % pre.m_per_batch == 0 && (gridDim.x & 7) == 0) { tpb = pre.m_per_batch / 16; nbe = M / pre.m_per_batch; if (nbe & 7) tpb = 0; }
    const int lw = ((int)blockIdx.x >> 3) * 4 + ((int)threadIdx.x >> 6), lnw = nwaves >> 3, lnt = tpb ? (nbe >> 3) * tpb : 0;
    // The rows of the NEXT tile (and the gather index of the one after it) travel while the current tile is multiplied: a wave's tiles form a chain of
    // ~60 matrix instructions behind two dependent global loads (index, then the gathered row), and at two waves per SIMD nothing else covers them
    // (round 4: 112 -> see DESIGN section 5).
    const int it_first = tpb ? lw : wave, it_end = tpb ? lnt : ntiles, it_step = tpb ? lnw : nwaves;
    auto tile_of = [&](int it) { return tpb ? (((int)blockIdx.x & 7) + 8 * (it / tpb)) * tpb + it % tpb : it; };
    auto row_of = [&](int it) { const int r = tile_of(it) * 16 + lc; return (it < it_end && r < M) ? r : -1; };
    auto load8r = [&](const float* p, int ok, float4 (&v)[2]) {
        v[0] = make_float4(0.f, 0.f, 0.f, 0.f); v[1] = v[0];
        if (ok) { v[0] = *reinterpret_cast<const float4*>(p + 8 * lg); v[1] = *reinterpret_cast<const float4*>(p + 8 * lg + 4); }
    };
    auto unpack8 = [](const float4 (&a)[2], float (&v)[8]) { v[0] = a[0].x; v[1] = a[0].y; v[2] = a[0].z; v[3] = a[0].w; v[4] = a[1].x; v[5] = a[1].y; v[6] = a[1].z; v[7] = a[1].w; };
    float4 n_sk[2], n_up[2];                 // rows of the next tile (PRE: skip and gathered; otherwise n_sk holds x)
    int idx_next = 0;                        // PRE: gather index of the tile after the next one's rows are requested with
    auto issue_rows = [&](int it, int idx) {
        const int r = row_of(it);
        if (PRE) {
            const int rw = r < 0 ? 0 : r;
            load8r(pre.skip + (size_t)rw * 32, r >= 0, n_sk);
            load8r(pre.up + ((size_t)(rw / pre.m_per_batch) * pre.up_rows_per_batch + (size_t)idx) * 32, r >= 0, n_up);
        } else load8r(x + (size_t)(r < 0 ? 0 : r) * 32, r >= 0, n_sk);
    };
    auto load_idx = [&](int it) { const int r = row_of(it); return (PRE && r >= 0) ? pre.idx[r] : 0; };
    {
        const int i0 = load_idx(it_first);
        idx_next = load_idx(it_first + it_step);
        issue_rows(it_first, i0);
    }
    for (int it = it_first; it < it_end; it += it_step) {
        const int tile = tile_of(it);
        const int row = tile * 16 + lc;
        float c_sk[8], c_up[8];
        unpack8(n_sk, c_sk);
        if (PRE) unpack8(n_up, c_up);
        {
            const int idx2 = load_idx(it + 2 * it_step);
            issue_rows(it + it_step, idx_next);
            idx_next = idx2;
        }
        SSDR_SCHED_FENCE();                  // the requests above stay above the products below
        u32x4 xf[2];
        if (PRE) {
            u32x4 sf[2], uf[2]; frag(c_sk, sf); frag(c_up, uf);
            f32x4 d0 = f32x4{BD[0].x, BD[0].y, BD[0].z, BD[0].w}, d1 = f32x4{BD[1].x, BD[1].y, BD[1].z, BD[1].w};
            { u32x4 w[2]; wfr(F_WD + 0, w); d0 = mma_split<TERMS>(w, sf, d0); wfr(F_WD + 2, w); d0 = mma_split<TERMS>(w, uf, d0);
              wfr(F_WD + 1, w); d1 = mma_split<TERMS>(w, sf, d1); wfr(F_WD + 3, w); d1 = mma_split<TERMS>(w, uf, d1); }
            const float v[8] = {lrelu(d0[0]), lrelu(d0[1]), lrelu(d0[2]), lrelu(d0[3]), lrelu(d1[0]), lrelu(d1[1]), lrelu(d1[2]), lrelu(d1[3])};
            frag(v, xf);
        } else frag(c_sk, xf);
        // fc1, transposed: lane (point lc, g) gets channels 16 ct + 4 g + reg
        float h1[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            f32x4 acc = f32x4{B1[ct].x, B1[ct].y, B1[ct].z, B1[ct].w};
            { u32x4 w[2]; wfr(F_W1 + ct, w); acc = mma_split<TERMS>(w, xf, acc); }
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[ct][r] = lrelu(acc[r]);
        }
        // fc2 in both orientations
        f32x4 at[2] = {f32x4{B2t[0].x, B2t[0].y, B2t[0].z, B2t[0].w}, f32x4{B2t[1].x, B2t[1].y, B2t[1].z, B2t[1].w}};
        f32x4 ac[2] = {f32x4{B2c[0], B2c[0], B2c[0], B2c[0]}, f32x4{B2c[1], B2c[1], B2c[1], B2c[1]}};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float v[8] = {h1[2 * kb][0], h1[2 * kb][1], h1[2 * kb][2], h1[2 * kb][3], h1[2 * kb + 1][0], h1[2 * kb + 1][1], h1[2 * kb + 1][2], h1[2 * kb + 1][3]};
            u32x4 hf[2]; frag(v, hf);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) { u32x4 w[2]; wfr(F_W2 + 2 * kb + ct, w); at[ct] = mma_split<TERMS>(w, hf, at[ct]); ac[ct] = mma_split<TERMS>(hf, w, ac[ct]); }
        }
        // last_second_features: accumulator layout (point 4 lg + reg, channel 16 ct + lc): 64 contiguous bytes per 16 lanes
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int rr = tile * 16 + 4 * lg + r; if (rr < M) feat32[(size_t)rr * 32 + 16 * ct + lc] = lrelu(ac[ct][r]); }
        // fc, transposed: classes 4 g + reg of point lc
        const float v2[8] = {lrelu(at[0][0]), lrelu(at[0][1]), lrelu(at[0][2]), lrelu(at[0][3]), lrelu(at[1][0]), lrelu(at[1][1]), lrelu(at[1][2]), lrelu(at[1][3])};
        u32x4 ff[2]; frag(v2, ff);
        f32x4 lg3 = f32x4{B3.x, B3.y, B3.z, B3.w};
        { u32x4 w[2]; wfr(F_W3, w); lg3 = mma_split<TERMS>(w, ff, lg3); }
        float z[4], m = -3.402823466e+38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { z[r] = 4 * lg + r < C ? lg3[r] : -3.402823466e+38f; m = fmaxf(m, z[r]); }
        m = rows_max(m);
        float e[4], sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = 4 * lg + r < C ? expf(z[r] - m) : 0.f; sum += e[r]; }
        sum = rows_sum(sum);
        if (row < M)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * lg + r < C) probs[(size_t)row * C + 4 * lg + r] = e[r] / sum;
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------
template <int TM, int KC>
static int launch_dense_bf16_t(const DenseArgs& a, int prec, bool vec, hipStream_t s) {
    dim3 grid((unsigned)((a.M + TM - 1) / TM), (unsigned)((a.N + 63) / 64));
    if (prec == PREC_BF16X3) {
        if (vec) hipLaunchKernelGGL((dense_bf16_kernel<TM, KC, 2, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((dense_bf16_kernel<TM, KC, 2, false>), grid, dim3(256), 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL((dense_bf16_kernel<TM, KC, 1, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((dense_bf16_kernel<TM, KC, 1, false>), grid, dim3(256), 0, s, a);
    }
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int launch_dense_bf16(const DenseArgs& a, int prec, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return SSDR_OK;
    if (a.N == 8 && a.k2 == 0 && a.k1 <= 16) return launch_dense(a, s);      // thin layers: HBM streams on fp32 FMAs in every mode
    if (!a.wt_hi || (prec == PREC_BF16X3 && !a.wt_lo)) { set_error("dense (bf16): the layer has no bf16 weight pieces"); return SSDR_ERR_INVALID; }
    if (a.xyz) SSDR_TRY(launch_xyz_fill(a, s));
    // executed: rows / columns padded to the tile, K to the chunk, one or three bf16 products
    const bool small = a.M <= 16384;
    const double tm = small ? 32.0 : 128.0, kc = small ? 64.0 : 32.0;
    const double exec = 2.0 * std::ceil(a.M / tm) * tm * std::ceil((a.k1 + a.k2) / kc) * kc * std::ceil(a.N / 64.0) * 64.0 * (prec == PREC_BF16X3 ? 3.0 : 1.0);
    ProfScope prof(small ? "dense_bf16_kernel<32,64>" : "dense_bf16_kernel<128,32>", s, 2.0 * (double)a.M * (double)(a.k1 + a.k2) * (double)a.N, exec);
    const bool vec = a.k1 % 4 == 0 && a.k2 % 4 == 0 && ((uintptr_t)a.x1 & 15) == 0 && ((uintptr_t)a.x2 & 15) == 0;
    // (a split-K form for these layers — every wave a quarter of the k steps, both operands straight from global memory in the MFMA layout, no LDS or
    // barrier in the K loop, partial tiles summed in LDS — was built and measured: 32 us per launch against 18.7: a 32-row tile re-reads its whole
    // 64-column weight panel and the CU takes ~30 bytes per clock from L2, which is what the 15 launches of this size are bound by, not the K chain)
    // (round 4: TM x 128 tiles on the 32 x 32 x 16 instruction — half the LDS operand bytes per FLOP, half the activation re-reads — were built for the 19
    // launches with N a multiple of 128 and measured: 534 us against 461 for those launches.  64 x 128 tiles leave the small-M layers 160-320 workgroups whose
    // K loop waits a global-load latency per 32-k chunk with nothing else resident to cover it; 128 x 128 tiles at M = 41-164 k ran level with the kernel below,
    // which is bound by what a CU fetches from L2 per clock, not by the matrix or LDS pipes.  Two more forms of THIS kernel for the small-M layers, same conclusion: 64-row
    // tiles (waves 2 x 2, half the weight re-reads) for K x N >= 512 x 512: 37.1 -> 35.0 and 37.5 -> 36.6 us; two K chunks in flight in registers: 470 -> 486 us over the 25 launches)
    // (round 6: 64-deep K chunks for the 128-row tiles — half the barriers and global-load round trips of the short K loops, 55 KB of LDS, two workgroups per CU:
    // 0.27 against 0.214 ms for the ten launches)
    if (small) return launch_dense_bf16_t<32, 64>(a, prec, vec, s);      // too few 128-row tiles to fill the chip
    return launch_dense_bf16_t<128, 32>(a, prec, vec, s);
}

// att_pooling's mlp + lrelu(mlp2 + shortcut) in one launch; SSDR_ERR_UNSUPPORTED (no error text) for shapes without an instantiation
int launch_dense_chain(const DenseArgs& l1, const DenseArgs& l2, int prec, hipStream_t s) {
    const int d = l1.k1;
    const bool shapes = l1.k2 == 0 && l1.N == d && l1.act && l2.act && l2.k1 == d && l2.N == 2 * d && l2.x1 == l1.y && l2.M == l1.M && !l2.idx2 && !l1.ldy && !l2.ldy && !l1.xyz && !l2.xyz &&
                        l2.k2 % 8 == 0 && l2.k2 > 0 && l1.wt_hi && l2.wt_hi && (prec != PREC_BF16X3 || (l1.wt_lo && l2.wt_lo)) &&
                        (((uintptr_t)l1.x1 | (uintptr_t)l2.x2) & 15) == 0 && l1.kp % 8 == 0 && l2.kp % 8 == 0;
    if (!shapes || l1.M <= 0) return SSDR_ERR_UNSUPPORTED;
    int k2p = 0;
    if (d == 16 && d + l2.k2 <= 32) k2p = 32; else if (d == 64 && d + l2.k2 <= 96) k2p = 96; else return SSDR_ERR_UNSUPPORTED;
    ChainArgs a{l1.x1, l2.x2, l2.k2, l1.wt_hi, l1.wt_lo, l1.kp, l1.b, l2.wt_hi, l2.wt_lo, l2.kp, l2.b, l2.y, l1.M};
    const double np = prec == PREC_BF16X3 ? 3.0 : 1.0, m128 = std::ceil(l1.M / 128.0) * 128.0;
    ProfScope prof(d == 16 ? "dense_chain_kernel<16>" : "dense_chain_kernel<64>", s, 2.0 * (double)l1.M * ((double)d * d + (double)(d + l2.k2) * 2.0 * d),
                   2.0 * m128 * ((d == 16 ? 32.0 * 16 : 64.0 * 64) + (double)k2p * (d == 16 ? 64.0 : 128.0)) * np);
    const bool t64 = d == 64;          // measured: d = 64 48.6 us on 64-row tiles (three workgroups per CU) against 60.1 on 128-row tiles; d = 16 35.7 against 33.0
    const dim3 grid((unsigned)((l1.M + (t64 ? 63 : 127)) / (t64 ? 64 : 128)));
#define SSDR_CHAIN(D_, K_, TM_) do { if (prec == PREC_BF16X3) hipLaunchKernelGGL((dense_chain_kernel<D_, K_, TM_, 2>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((dense_chain_kernel<D_, K_, TM_, 1>), grid, dim3(256), 0, s, a); } while (0)
    if (d == 16) { if (t64) SSDR_CHAIN(16, 32, 64); else SSDR_CHAIN(16, 32, 128); }
    else { if (t64) SSDR_CHAIN(64, 96, 64); else SSDR_CHAIN(64, 96, 128); }
#undef SSDR_CHAIN
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int launch_tail_bf16(const TailArgs& t, int prec, hipStream_t s) {
    if (t.M <= 0) return SSDR_OK;
    if ((t.C != 13 && t.C != 8) || (!t.skip && ((uintptr_t)t.x & 15)) || (t.skip && (((uintptr_t)t.skip | (uintptr_t)t.up) & 15)) || !t.w1h || !t.w2h || !t.w3h || (prec == PREC_BF16X3 && (!t.w1l || !t.w2l || !t.w3l))) return SSDR_ERR_UNSUPPORTED;
    const double m16 = std::ceil(t.M / 16.0) * 16.0;
    ProfScope prof("tail_kernel", s, (double)t.M * 4.0 * (32 + 32 + t.C), 2.0 * m16 * ((t.skip ? 64.0 * 32 : 0.0) + 32.0 * 64 + 2.0 * 64 * 32 + 32.0 * 16) * (prec == PREC_BF16X3 ? 3.0 : 1.0));
    // four workgroups per CU = what its registers keep resident since the weight fragments moved to LDS (one copy per workgroup; eight per CU with the fragments in
    // registers, waves of 5 tiles: 117 us; two: 90; fragments in LDS, 124 registers: two 85, three 80, four 77, six 78)
    const dim3 g((unsigned)std::max(1, std::min((t.M + 63) / 64, ctx().num_cu * 4)));
    TailPre pre{t.skip, t.up, t.idx, t.m_per_batch, t.up_rows_per_batch, t.wdh, t.wdl, t.kpd, t.bd};
#define SSDR_TAIL(TERMS_, C_) do { if (t.skip) hipLaunchKernelGGL((tail_bf16_kernel<TERMS_, C_, true>), g, dim3(256), 0, s, pre, t.x, t.w1h, t.w1l, t.kp1, t.b1, t.w2h, t.w2l, t.kp2, t.b2, t.w3h, t.w3l, t.kp3, t.b3, t.M, t.feat32, t.probs); \
                               else hipLaunchKernelGGL((tail_bf16_kernel<TERMS_, C_, false>), g, dim3(256), 0, s, pre, t.x, t.w1h, t.w1l, t.kp1, t.b1, t.w2h, t.w2l, t.kp2, t.b2, t.w3h, t.w3l, t.kp3, t.b3, t.M, t.feat32, t.probs); } while (0)
    if (prec == PREC_BF16X3) { if (t.C == 13) SSDR_TAIL(2, 13); else SSDR_TAIL(2, 8); }
    else { if (t.C == 13) SSDR_TAIL(1, 13); else SSDR_TAIL(1, 8); }
#undef SSDR_TAIL
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

template <int D> static int launch_lfa_bf16_d(const LfaArgs& a, bool second, int B, int prec, hipStream_t s) {
    using C = LfaBf16Cfg<D>;
    dim3 grid((unsigned)((a.n + C::PTS - 1) / C::PTS), (unsigned)B);
    const int terms = prec == PREC_BF16X3 ? 2 : 1;
    const size_t lds = C::lds_bytes(terms);
    static std::once_flag attr_once;
    hipError_t ae = hipSuccess;
    std::call_once(attr_once, [&] {
        ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&lfa_bf16_kernel<D, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes(2));
        if (ae == hipSuccess) ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&lfa_bf16_kernel<D, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes(2));
        if (ae == hipSuccess) ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&lfa_bf16_kernel<D, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes(1));
        if (ae == hipSuccess) ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&lfa_bf16_kernel<D, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes(1));
    });
    SSDR_HIP(ae);
    const double rows = (double)B * (double)a.n * 16.0;       // algorithmic FLOPs of the reference's formulation (as lfa_att_kernel)
    // executed on the matrix cores: LocSE on the f32 MFMA (K padded to 12; both orientations in the first half), LFAmlp2 in both orientations and
    // the position half of the attention product on the bf16 MFMA (one or three products)
    const double np = terms == 2 ? 3.0 : 1.0;
    const double exec = rows * (2.0 * 12 * C::H * (second ? 1.0 : 2.0) + (second ? 2.0 * 2.0 * C::H * C::H * np : 0.0) + 2.0 * C::H * D * np);
    ProfScope prof("lfa_att_kernel", s, rows * (2.0 * 10 * C::H + (second ? 2.0 * C::H * C::H : 0.0) + 2.0 * D * D + 2.0 * D), exec);
    if (terms == 2) {
        if (second) hipLaunchKernelGGL((lfa_bf16_kernel<D, true, 2>), grid, dim3(C::NT), lds, s, a);
        else hipLaunchKernelGGL((lfa_bf16_kernel<D, false, 2>), grid, dim3(C::NT), lds, s, a);
    } else {
        if (second) hipLaunchKernelGGL((lfa_bf16_kernel<D, true, 1>), grid, dim3(C::NT), lds, s, a);
        else hipLaunchKernelGGL((lfa_bf16_kernel<D, false, 1>), grid, dim3(C::NT), lds, s, a);
    }
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int launch_lfa_bf16(int D, const LfaArgs& a, bool second, int B, int prec, hipStream_t s) {
    if (a.n <= 0 || B <= 0) return SSDR_OK;
    if (!a.g || !a.fc_hi || (second && !a.l2_hi)) { set_error("lfa (bf16): missing G rows or bf16 weight pieces"); return SSDR_ERR_INVALID; }
    switch (D) {
        case 64: return launch_lfa_bf16_d<64>(a, second, B, prec, s);
        case 128: return launch_lfa_bf16_d<128>(a, second, B, prec, s);
        case 256: return launch_lfa_bf16_d<256>(a, second, B, prec, s);
        case 512: return launch_lfa_bf16_d<512>(a, second, B, prec, s);
        default: set_error("d_out=%d has no bf16 LFA kernel (64, 128, 256, 512)", D); return SSDR_ERR_UNSUPPORTED;
    }
}

}  // namespace ssdr
