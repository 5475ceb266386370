// RandLA-Net inference kernels for gfx950 (fp32 data, exact-f32 MFMA v_mfma_f32_16x16x4_f32).
//
// Reference network: S3/RandLANet.py:140-180 (inference), :505-585 (dilated_res_block, building_block,
// relative_pos_encoding, att_pooling, random_sample, nearest_interpolation) over S3/helper_tf_util.py:111-246
// (1x1 conv + bias + BN + leaky-relu 0.2).  BN is folded into W/b on the host, so every conv is
// y = act(x W + b).
//
//   lfa_att_kernel   the K-expanded part of one building_block half, fused so that no [N,K,d] tensor ever
//                    reaches HBM: neighbour gather + relative position encoding (:529-535) + LocSE 1x1 conv
//                    (+ for the second half the LFAmlp2 conv of the *encoded* positions, :523) + concat +
//                    attention scores (dense d x d, :578) + softmax over the K neighbours (:579) + weighted
//                    sum (:580-581).  One 16x16 MFMA tile = the 16 neighbours of one point x 16 channels, so
//                    the softmax over K is a reduction over the 4 accumulator registers and 4 lane groups.
//   dense_kernel     per-point 1x1 convs as a tiled GEMM with up to two concatenated inputs, the second one
//                    optionally row-gathered (decoder: concat[skip, nearest_interpolation(feature)], :165-170;
//                    residual: mlp2(agg) + shortcut(feature) as one GEMM over the stacked weights, :508-512).
//   gather_max       random_sample (:537-548): gather K rows, max over K.
//   head_kernel      fc (32 -> C, no BN / no act, :176) + softmax (:84).
#include "ssdr_internal.hpp"
#include "randla.hpp"
#include "randla_dev.hpp"

namespace ssdr {

// ---- dense --------------------------------------------------------------------------------------------
// 128 x 64 output tile per workgroup, K in chunks of 32; wave w owns rows [32w, 32w+32) x all 64 columns as 2 x 4
// MFMA tiles.  The next K chunk is fetched into registers (float4 when every row segment is 16-byte aligned) while
// the current one is multiplied out of LDS, so global latency hides behind 64 MFMAs per wave and chunk.
constexpr int DTM = 128, DTN = 64, DKC = 32, DAS = DKC + 2, DBS = DTN + 16;

template <bool VEC>
__global__ __launch_bounds__(256) void dense_kernel(DenseArgs a) {
    __shared__ float As[DTM * DAS];
    __shared__ float Bs[DKC * DBS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row0 = blockIdx.x * DTM, col0 = blockIdx.y * DTN;
    const int K = a.k1 + a.k2;
    // A staging: thread -> rows ar, ar+64 ; 8 consecutive k at ak.   B staging: k row bk, 8 consecutive columns at bc
    const int ar = tid >> 2, ak = (tid & 3) * 8;
    const float* x1r[2]; const float* x2r[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int grow = row0 + ar + 64 * h;
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
    const int bk = tid >> 3, bc = (tid & 7) * 8;
    float ra[2][8], rb[8];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (VEC) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int gk = kc + ak + 4 * q;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (x1r[h] && gk < K) v = (gk < a.k1) ? *reinterpret_cast<const float4*>(x1r[h] + gk) : *reinterpret_cast<const float4*>(x2r[h] + (gk - a.k1));
                    ra[h][4 * q] = v.x; ra[h][4 * q + 1] = v.y; ra[h][4 * q + 2] = v.z; ra[h][4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int gk = kc + ak + j;
                    float v = 0.f;
                    if (x1r[h]) { if (gk < a.k1) v = x1r[h][gk]; else if (gk < K) v = x2r[h][gk - a.k1]; }
                    ra[h][j] = v;
                }
            }
        }
        const int gk = kc + bk;
        if (VEC) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int gc = col0 + bc + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gk < K && gc < a.N) v = *reinterpret_cast<const float4*>(a.W + (size_t)gk * a.N + gc);
                rb[4 * q] = v.x; rb[4 * q + 1] = v.y; rb[4 * q + 2] = v.z; rb[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int gc = col0 + bc + j; rb[j] = (gk < K && gc < a.N) ? a.W[(size_t)gk * a.N + gc] : 0.f; }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) As[(ar + 64 * h) * DAS + ak + j] = ra[h][j];
#pragma unroll
        for (int j = 0; j < 8; ++j) Bs[bk * DBS + bc + j] = rb[j];
    };
    f32x4 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    fetch(0);
    for (int kc = 0; kc < K; kc += DKC) {
        stash();
        __syncthreads();
        if (kc + DKC < K) fetch(kc + DKC);          // in flight while the MFMAs below run
#pragma unroll
        for (int ks = 0; ks < DKC / 4; ++ks) {
            float av[2], bv[4];
#pragma unroll
            for (int r = 0; r < 2; ++r) av[r] = As[(w * 32 + r * 16 + (lane & 15)) * DAS + ks * 4 + (lane >> 4)];
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[c] = Bs[(ks * 4 + (lane >> 4)) * DBS + c * 16 + (lane & 15)];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = mfma16(av[r], bv[c], acc[r][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = col0 + c * 16 + (lane & 15);
        if (col >= a.N) continue;
        const float bias = a.b ? a.b[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = row0 + w * 32 + r * 16 + (lane >> 4) * 4 + q;
                if (row < a.M) { float v = acc[r][c][q] + bias; a.y[(size_t)row * (a.ldy ? a.ldy : a.N) + col] = a.act ? lrelu(v) : v; }
            }
    }
}

// ---- dense, few rows ----------------------------------------------------------------------------------------
// The deep levels have 1280-10240 rows: 128-row tiles leave most of the 256 CUs without a workgroup and every
// workgroup walks K = 512-1536 as a serial chain of chunks.  32 x 64 tiles give 4x the workgroups; wave w owns the
// 16 columns [16w, 16w+16) of both 16-row tiles.  Same staging / prefetch scheme as dense_kernel (VEC only).
constexpr int STM = 32;
__global__ __launch_bounds__(256) void dense_small_kernel(DenseArgs a) {
    __shared__ float As[STM * DAS];
    __shared__ float Bs[DKC * DBS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row0 = blockIdx.x * STM, col0 = blockIdx.y * DTN;
    const int K = a.k1 + a.k2;
    const int ar = tid >> 3, ak = (tid & 7) * 4;           // A staging: row ar, 4 consecutive k
    const float* x1r = nullptr; const float* x2r = nullptr;
    {
        const int grow = row0 + ar;
        if (grow < a.M) {
            x1r = a.x1 + (size_t)grow * a.k1;
            if (a.k2) {
                size_t r2 = (size_t)grow;
                if (a.idx2) r2 = (size_t)(grow / a.m_per_batch) * a.x2_rows_per_batch + (size_t)a.idx2[grow];
                x2r = a.x2 + r2 * a.k2;
            }
        }
    }
    const int bk = tid >> 3, bc = (tid & 7) * 8;           // B staging: k row bk, 8 consecutive columns
    float ra[4], rb[8];
    auto fetch = [&](int kc) {
        const int gk = kc + ak;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x1r && gk < K) v = (gk < a.k1) ? *reinterpret_cast<const float4*>(x1r + gk) : *reinterpret_cast<const float4*>(x2r + (gk - a.k1));
        ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3] = v.w;
        const int gkb = kc + bk;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int gc = col0 + bc + 4 * q;
            float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gkb < K && gc < a.N) u = *reinterpret_cast<const float4*>(a.W + (size_t)gkb * a.N + gc);
            rb[4 * q] = u.x; rb[4 * q + 1] = u.y; rb[4 * q + 2] = u.z; rb[4 * q + 3] = u.w;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) As[ar * DAS + ak + j] = ra[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) Bs[bk * DBS + bc + j] = rb[j];
    };
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    fetch(0);
    for (int kc = 0; kc < K; kc += DKC) {
        stash();
        __syncthreads();
        if (kc + DKC < K) fetch(kc + DKC);          // in flight while the MFMAs below run
#pragma unroll
        for (int ks = 0; ks < DKC / 4; ++ks) {
            const float bv = Bs[(ks * 4 + (lane >> 4)) * DBS + w * 16 + (lane & 15)];
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[r] = mfma16(As[(r * 16 + (lane & 15)) * DAS + ks * 4 + (lane >> 4)], bv, acc[r]);
        }
        __syncthreads();
    }
    const int col = col0 + w * 16 + (lane & 15);
    if (col < a.N) {
        const float bias = a.b ? a.b[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = row0 + r * 16 + (lane >> 4) * 4 + q;
                if (row < a.M) { float v = acc[r][q] + bias; a.y[(size_t)row * (a.ldy ? a.ldy : a.N) + col] = a.act ? lrelu(v) : v; }
            }
    }
}

// ---- fused local-feature-aggregation attention half -----------------------------------------------------
template <int D> struct LfaCfg {
    static constexpr int H = D / 2;
    // points per workgroup: small enough that 3-5 workgroups share a CU's LDS at the low levels (their phases are
    // latency-bound gathers), large enough at d >= 256 that a W element fetched from L2 serves several points
    static constexpr int PTS = D == 16 ? 8 : (D == 64 ? 8 : (D == 128 ? 4 : (D == 256 ? 4 : 2)));      // re-measured after the gather fix: 8/4/2/2 variants within 2 %
    static constexpr int ROWS = PTS * 16;
    static constexpr bool WIDE = D >= 64;                              // 16-byte operand fetches in the attention GEMM
    static constexpr int LD = WIDE ? D + 4 : D + 2;                    // LDS row stride: 16-byte aligned rows (WIDE) / == 2 (mod 32): conflict-free 4-byte A reads
    static constexpr int NCT = D / 16;                                 // column tiles of the attention GEMM
    static constexpr int NC_W = (D / 64) > 1 ? (D / 64) : 1;           // column tiles per wave
    static constexpr int NP_W = (PTS * NCT / 4) / NC_W;                // point tiles per wave
    static constexpr int NCT2 = (H / 16) > 1 ? (H / 16) : 1;           // column tiles of the LFAmlp2 GEMM
    static constexpr size_t LDS_BYTES = sizeof(float) * ((size_t)ROWS * LD + (size_t)ROWS * 10) + sizeof(int) * ROWS;
};

template <int D, bool SECOND>
__global__ __launch_bounds__(256) void lfa_att_kernel(LfaArgs a) {
    using C = LfaCfg<D>;
    constexpr int H = C::H, PTS = C::PTS, ROWS = C::ROWS, LD = C::LD;
    SSDR_DYN_SHARED(float, smem);
    float* F = smem;                              // [ROWS][LD]   concat(f_neighbours, f_xyz)
    float* REL = smem + (size_t)ROWS * LD;        // [ROWS][10]
    int* NBR = reinterpret_cast<int*>(REL + (size_t)ROWS * 10);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bx_, b; xcd_tile_map(bx_, b);
    const int pt0 = bx_ * PTS;
    const float* xyz = a.xyz + (size_t)b * a.xyz_batch_stride;
    const int* neigh = a.neigh + (size_t)b * a.n * 16;

    // LocSE weights of this wave's tiles: requested first, so their latency hides behind the position encoding (inside
    // the tile loop every tile waited for its own three loads: 100 us of the 280 us level-0 launch)
    constexpr int NCT1 = C::NCT2, T1W = PTS * NCT1 / 4;
    static_assert(PTS * NCT1 % 4 == 0, "LocSE tiles split evenly over the 4 waves");
    constexpr int NB1 = T1W < NCT1 ? T1W : NCT1;       // distinct column tiles among this wave's tiles (tile t uses slot t % NB1)
    static_assert(T1W % NB1 == 0 && (T1W >= NCT1 ? T1W % NCT1 == 0 : NCT1 % T1W == 0), "slot rule");
    float w1[NB1][3], b1[NB1];
#pragma unroll
    for (int t = 0; t < NB1; ++t) {
        const int col = ((w * T1W + t) % NCT1) * 16 + (lane & 15);
        const int colc = col < H ? col : H - 1;            // clamped, no divergent loads; masked below
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int k = ks * 4 + (lane >> 4);
            const float v = a.w_l1[(k < 10 ? k : 9) * H + colc];
            w1[t][ks] = (k < 10 && col < H) ? v : 0.f;
        }
        const float bv = a.b_l1[colc];
        b1[t] = col < H ? bv : 0.f;
    }

    // relative_pos_encoding (:529-535): [ |d|, d(3), p(3), p_nbr(3) ]
    for (int row = tid; row < ROWS; row += 256) {
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

    // LocSE conv 10 -> H (LFAmlp1, :518) on the matrix cores: [16 neighbour rows] x [10 (padded to 12)] x [16 channels]
    // tiles, 3 MFMAs each (the per-output VALU loop costs more than the whole attention GEMM at d <= 64).
    // First half: straight into the f_xyz columns [H,2H).  Second half: into columns [0,H) as the A operand of the
    // LFAmlp2 GEMM below.
    {
        constexpr int XOFF = SECOND ? 0 : H;
#pragma unroll
        for (int t = 0; t < T1W; ++t) {
            const int tile = w * T1W + t, p = tile / NCT1, ct = tile % NCT1;
            const int col = ct * 16 + (lane & 15);
            f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int k = ks * 4 + (lane >> 4);
                const float av1 = (k < 10) ? REL[(p * 16 + (lane & 15)) * 10 + k] : 0.f;
                acc1 = mfma16(av1, w1[t % NB1][ks], acc1);
            }
            if (col < H) {
#pragma unroll
                for (int r = 0; r < 4; ++r) F[(p * 16 + (lane >> 4) * 4 + r) * LD + XOFF + col] = lrelu(acc1[r] + b1[t % NB1]);
            }
        }
    }
    __syncthreads();

    if (SECOND) {
        // f_xyz <- conv H -> H of the encoded positions (LFAmlp2, :523) on the matrix cores
        constexpr int NCT2 = C::NCT2, T2W = PTS * NCT2 / 4;
        f32x4 acc2[T2W];
#pragma unroll
        for (int t = 0; t < T2W; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // k outer, tiles inner, operands of the next k-step requested before this step's MFMAs (as in the main GEMM)
        float a2[T2W], b2[T2W], a2n[T2W], b2n[T2W];
        auto load2 = [&](int kb, float (&av)[T2W], float (&bv)[T2W]) {
            const int k = kb + (lane >> 4);
#pragma unroll
            for (int t = 0; t < T2W; ++t) {
                const int tile = w * T2W + t, p = tile / NCT2, ct = tile % NCT2;
                const int col = ct * 16 + (lane & 15);
                av[t] = (k < H) ? F[(p * 16 + (lane & 15)) * LD + k] : 0.f;
                bv[t] = (k < H && col < H) ? a.w_l2[k * H + col] : 0.f;
            }
        };
        load2(0, a2, b2);
        for (int kb = 0; kb < H; kb += 4) {
            if (kb + 4 < H) load2(kb + 4, a2n, b2n);
#pragma unroll
            for (int t = 0; t < T2W; ++t) acc2[t] = mfma16(a2[t], b2[t], acc2[t]);
#pragma unroll
            for (int t = 0; t < T2W; ++t) { a2[t] = a2n[t]; b2[t] = b2n[t]; }
        }
#pragma unroll
        for (int t = 0; t < T2W; ++t) {
            const int tile = w * T2W + t, p = tile / NCT2, ct = tile % NCT2;
            const int col = ct * 16 + (lane & 15);
            if (col < H) {
                const float bv = a.b_l2[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) F[(p * 16 + (lane >> 4) * 4 + r) * LD + H + col] = lrelu(acc2[t][r] + bv);
            }
        }
        __syncthreads();
    }

    // gather_neighbour (:519 / :524): neighbour features into columns [0,H).  Every thread first requests ALL of its
    // 16-byte pieces and only then stores them to LDS (a load -> wait -> store loop is one memory round trip per
    // iteration, 32 of them at d = 512).
    {
        const float* fin = a.fin + (size_t)b * a.n * H;
        constexpr int H4 = H / 4, NV = ROWS * H4 / 256;
        static_assert(ROWS * H4 % 256 == 0 && NV >= 1, "gather tiling");
        float4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + 256 * i, row = e / H4, c4 = e % H4;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pt0 + (row >> 4) < a.n) v[i] = *reinterpret_cast<const float4*>(fin + (size_t)NBR[row] * H + 4 * c4);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + 256 * i, row = e / H4, c4 = e % H4;
            if (C::WIDE) *reinterpret_cast<float4*>(&F[row * LD + 4 * c4]) = v[i];           // rows are 16-byte aligned
            else {
                float2* dst = reinterpret_cast<float2*>(&F[row * LD + 4 * c4]);      // LD is even: 8-byte aligned
                dst[0] = make_float2(v[i].x, v[i].y); dst[1] = make_float2(v[i].z, v[i].w);
            }
        }
    }
    __syncthreads();

    // attention scores S = F * Wfc (:578), one 16x16 tile = 16 neighbours x 16 channels
    constexpr int NC_W = C::NC_W, NP_W = C::NP_W;
    const int ct0 = (D >= 64) ? w * NC_W : 0;
    const int p0 = (D >= 64) ? 0 : w * NP_W;
    f32x4 acc[NP_W][NC_W];
#pragma unroll
    for (int p = 0; p < NP_W; ++p)
#pragma unroll
        for (int c = 0; c < NC_W; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (C::WIDE) {
        // The neighbour-feature half of the product depends on the neighbour alone: (f_nbr | f_xyz) W = f_nbr W[0:H] + f_xyz W[H:D],
        // and G = f W[0:H] was computed once per POINT by a plain dense launch (16x fewer rows).  The accumulators start from the
        // gathered rows of G, the MFMA loop only covers the position half: half the matrix-core work of the attention GEMM.
        const int kb0 = a.g ? H / 16 : 0;
        if (a.g) {
            const float* G = a.g + (size_t)b * a.n * D;
#pragma unroll
            for (int p = 0; p < NP_W; ++p) {
                int nb4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) nb4[r] = NBR[(p0 + p) * 16 + (lane >> 4) * 4 + r];
                const bool ok = pt0 + p0 + p < a.n;
#pragma unroll
                for (int c = 0; c < NC_W; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[p][c][r] = ok ? G[(size_t)nb4[r] * D + (ct0 + c) * 16 + (lane & 15)] : 0.f;
            }
        }
        // k is taken in blocks of 16: in MFMA step u of a block lane group g supplies k = 16 kb + 4 g + u (the sum over k does not
        // care about the order), so a lane's four A values are 16 contiguous bytes of its LDS row and its four W values 16
        // contiguous bytes of the transposed weights: one ds_read_b128 + one global dwordx4 feed four MFMAs.  The next block
        // is requested before the current block's 4 x NP_W x NC_W MFMAs issue.
        float av[NP_W][4], bv[NC_W][4], an[NP_W][4], bn[NC_W][4];
        auto fetch16 = [&](int kb, float (&fa)[NP_W][4], float (&fb)[NC_W][4]) {
            const int k0 = kb * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int p = 0; p < NP_W; ++p) {
                const float4 v = *reinterpret_cast<const float4*>(&F[((p0 + p) * 16 + (lane & 15)) * LD + k0]);
                fa[p][0] = v.x; fa[p][1] = v.y; fa[p][2] = v.z; fa[p][3] = v.w;
            }
#pragma unroll
            for (int c = 0; c < NC_W; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(a.w_fc_t + (size_t)((ct0 + c) * 16 + (lane & 15)) * D + k0);
                fb[c][0] = v.x; fb[c][1] = v.y; fb[c][2] = v.z; fb[c][3] = v.w;
            }
        };
        fetch16(kb0, av, bv);
        for (int kb = kb0; kb < D / 16; ++kb) {
            if (kb + 1 < D / 16) fetch16(kb + 1, an, bn);
#pragma unroll
            for (int u = 0; u < 4; ++u)                      // consecutive MFMAs go to different accumulators
#pragma unroll
                for (int p = 0; p < NP_W; ++p)
#pragma unroll
                    for (int c = 0; c < NC_W; ++c) acc[p][c] = mfma16(av[p][u], bv[c][u], acc[p][c]);
#pragma unroll
            for (int p = 0; p < NP_W; ++p)
#pragma unroll
                for (int u = 0; u < 4; ++u) av[p][u] = an[p][u];
#pragma unroll
            for (int c = 0; c < NC_W; ++c)
#pragma unroll
                for (int u = 0; u < 4; ++u) bv[c][u] = bn[c][u];
        }
    } else {
        // software pipeline: the operands of the NEXT two k-steps (A from LDS, W from L2) are requested before the 2 x 16
        // MFMAs of the current two issue, i.e. ~1000 MFMA cycles of cover for an L2 round trip
        constexpr int KS = (D >= 8) ? 2 : 1;           // k-steps (of 4) per pipeline stage
        float av[KS][NP_W], bv[KS][NC_W], an[KS][NP_W], bn[KS][NC_W];
        auto fetch = [&](int kb, float (&fa)[KS][NP_W], float (&fb)[KS][NC_W]) {
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                const int k = kb + 4 * u + (lane >> 4);
#pragma unroll
                for (int p = 0; p < NP_W; ++p) fa[u][p] = F[((p0 + p) * 16 + (lane & 15)) * LD + k];
#pragma unroll
                for (int c = 0; c < NC_W; ++c) fb[u][c] = a.w_fc[(size_t)k * D + (ct0 + c) * 16 + (lane & 15)];
            }
        };
        fetch(0, av, bv);
        for (int kb = 0; kb < D; kb += 4 * KS) {
            if (kb + 4 * KS < D) fetch(kb + 4 * KS, an, bn);
#pragma unroll
            for (int u = 0; u < KS; ++u)
#pragma unroll
                for (int p = 0; p < NP_W; ++p)
#pragma unroll
                    for (int c = 0; c < NC_W; ++c) acc[p][c] = mfma16(av[u][p], bv[u][c], acc[p][c]);
#pragma unroll
            for (int u = 0; u < KS; ++u) {
#pragma unroll
                for (int p = 0; p < NP_W; ++p) av[u][p] = an[u][p];
#pragma unroll
                for (int c = 0; c < NC_W; ++c) bv[u][c] = bn[u][c];
            }
        }
    }

    // softmax over the 16 neighbours (:579) and weighted sum (:580-581)
    float* out = a.out + (size_t)b * a.n * D;
#pragma unroll
    for (int p = 0; p < NP_W; ++p) {
        const int n = pt0 + p0 + p;
#pragma unroll
        for (int c = 0; c < NC_W; ++c) {
            const int col = (ct0 + c) * 16 + (lane & 15);
            float m = fmaxf(fmaxf(acc[p][c][0], acc[p][c][1]), fmaxf(acc[p][c][2], acc[p][c][3]));
            m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
            float e[4], s = 0.f, v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { e[r] = __expf(acc[p][c][r] - m); s += e[r]; }    // v_exp_f32; within 2 ulp, far inside the 1e-3 feature tolerance
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
#pragma unroll
            for (int r = 0; r < 4; ++r) v += F[((p0 + p) * 16 + (lane >> 4) * 4 + r) * LD + col] * e[r];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            v = v * fast_rcp(s);                                         // sum_k f_k e_k / sum_k e_k: one reciprocal (1 ulp) per output
            if ((lane >> 4) == 0 && n < a.n) out[(size_t)n * D + col] = v;
        }
    }
}

// ---- random_sample (:537-548) --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_max_kernel(const float* __restrict__ f, const int* __restrict__ idx, int n_in, int n_out,
                                                         int idx_rows, int C, float* __restrict__ out) {
    int bx, b; xcd_tile_map(bx, b);
    const float* fb = f + (size_t)b * n_in * C;
    const int* ib = idx + (size_t)b * idx_rows * 16;
    float* ob = out + (size_t)b * n_out * C;
    const size_t total = (size_t)n_out * C;
    for (size_t e = (size_t)bx * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int m = (int)(e / C), c = (int)(e % C);
        float v = -3.402823466e+38f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v = fmaxf(v, fb[(size_t)ib[(size_t)m * 16 + k] * C + c]);
        ob[e] = v;
    }
}

// four channels per thread: the sixteen indices arrive as four 16-byte loads, every gather is one 16-byte load per lane and the offsets are 32-bit (the
// one-channel form above spends 60 % of its time on address arithmetic for loads the texture path prices per dword anyway)
__global__ __launch_bounds__(256) void gather_max4_kernel(const float* __restrict__ f, const int* __restrict__ idx, int n_in, int n_out,
                                                          int idx_rows, int C, float* __restrict__ out) {
    int bx, b; xcd_tile_map(bx, b);
    const float* fb = f + (size_t)b * n_in * C;
    const int* ib = idx + (size_t)b * idx_rows * 16;
    float* ob = out + (size_t)b * n_out * C;
    const unsigned C4 = (unsigned)C / 4u, total = (unsigned)n_out * C4;
    for (unsigned e = (unsigned)bx * 256u + threadIdx.x; e < total; e += (unsigned)gridDim.x * 256u) {
        const unsigned m = e / C4, c4 = e - m * C4;
        const int4* ip = reinterpret_cast<const int4*>(ib + (size_t)m * 16);
        const int4 i0 = ip[0], i1 = ip[1], i2 = ip[2], i3 = ip[3];
        const int id[16] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w, i3.x, i3.y, i3.z, i3.w};
        float4 x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = *reinterpret_cast<const float4*>(fb + ((unsigned)id[k] * (unsigned)C + 4u * c4));
        float4 v = x[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) { v.x = fmaxf(v.x, x[k].x); v.y = fmaxf(v.y, x[k].y); v.z = fmaxf(v.z, x[k].z); v.w = fmaxf(v.w, x[k].w); }
        *reinterpret_cast<float4*>(ob + (size_t)e * 4) = v;
    }
}

// ---- fc (32 -> C) + softmax (:176, :84) ----------------------------------------------------------------
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                   int M, int C, float* __restrict__ probs) {
    __shared__ float Ws[32 * 32 + 32];
    for (int i = threadIdx.x; i < 32 * C; i += 256) Ws[i] = W[i];
    for (int i = threadIdx.x; i < C; i += 256) Ws[32 * 32 + i] = bias[i];
    __syncthreads();
    for (int r = blockIdx.x * 256 + threadIdx.x; r < M; r += gridDim.x * 256) {
        float xin[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) xin[i] = x[(size_t)r * 32 + i];
        float logit[32]; float m = -3.402823466e+38f;
        for (int c = 0; c < C; ++c) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) s += xin[i] * Ws[i * C + c];
            s += Ws[32 * 32 + c]; logit[c] = s; m = fmaxf(m, s);
        }
        float sum = 0.f;
        for (int c = 0; c < C; ++c) { logit[c] = expf(logit[c] - m); sum += logit[c]; }
        for (int c = 0; c < C; ++c) probs[(size_t)r * C + c] = logit[c] / sum;
    }
}

// ---- thin layers: one row per lane -------------------------------------------------------------------------
// For k1 <= 16 and N = 8 (fc0 and the level-0 encoder convs into 8 channels) the 128 x 64 MFMA tile is
// mostly padding and a workgroup lives for one latency-bound K chunk; these layers are pure HBM streams (a few hundred
// FLOP per row).  One lane = one row: the row in registers, the weights broadcast from LDS, fused multiply-adds.
template <int K1, int K2, int N>
__global__ __launch_bounds__(256) void dense_rows_kernel(DenseArgs a) {
    constexpr int K = K1 + K2;
    __shared__ float Ws[K * N + N];
    for (int i = threadIdx.x; i < K * N; i += 256) Ws[i] = a.W[i];
    for (int i = threadIdx.x; i < N; i += 256) Ws[K * N + i] = a.b ? a.b[i] : 0.f;
    __syncthreads();
    // one row per thread, no row loop: the weight reads would be hoisted out of it as loop invariants (K*N live registers)
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row < a.M) {
        float x[K];
        const float* r1 = a.x1 + (size_t)row * K1;
        if (K1 % 4 == 0) {
#pragma unroll
            for (int q = 0; q < K1 / 4; ++q) { const float4 v = reinterpret_cast<const float4*>(r1)[q]; x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w; }
        } else {
#pragma unroll
            for (int k = 0; k < K1; ++k) x[k] = r1[k];
        }
        if (K2 > 0) {
            size_t r2 = (size_t)row;
            if (a.idx2) r2 = (size_t)(row / a.m_per_batch) * a.x2_rows_per_batch + (size_t)a.idx2[row];
            const float4* p2 = reinterpret_cast<const float4*>(a.x2 + r2 * K2);
#pragma unroll
            for (int q = 0; q < K2 / 4; ++q) { const float4 v = p2[q]; x[K1 + 4 * q] = v.x; x[K1 + 4 * q + 1] = v.y; x[K1 + 4 * q + 2] = v.z; x[K1 + 4 * q + 3] = v.w; }
        }
        float acc[N];
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = Ws[K * N + n];
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = fmaf(x[k], Ws[k * N + n], acc[n]);
            SSDR_SCHED_FENCE();
        }
        float4* yo = reinterpret_cast<float4*>(a.y + (size_t)row * (a.ldy ? a.ldy : N));
        if (a.xyz) {      // the row's coordinates in front of its features: one 64-byte row of the level-0 gather table
            const int be = row / a.xyz_rows_per_batch, i = row - be * a.xyz_rows_per_batch;
            const float* c = a.xyz + (size_t)be * a.xyz_batch_stride + 3 * (size_t)i;
            yo[-1] = make_float4(c[0], c[1], c[2], 0.f);
        }
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            if (a.act) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
            yo[q] = v;
        }
    }
}

// Two thin layers in one pass over the rows (fc0 -> level 0's mlp1, RandLANet.py:139-141 / :506): h = act(x W1 + b1) is written (the residual
// shortcut reads it later) AND fed to y = act(h W2 + b2) from registers — the same fused multiply-adds in the same order as two launches of
// dense_rows_kernel, one read of h less.
template <int K1, int N1, int N2>
__global__ __launch_bounds__(256) void dense_rows2_kernel(DenseArgs a, DenseArgs b) {
    __shared__ float W1s[K1 * N1 + N1], W2s[N1 * N2 + N2];
    for (int i = threadIdx.x; i < K1 * N1; i += 256) W1s[i] = a.W[i];
    for (int i = threadIdx.x; i < N1; i += 256) W1s[K1 * N1 + i] = a.b ? a.b[i] : 0.f;
    for (int i = threadIdx.x; i < N1 * N2; i += 256) W2s[i] = b.W[i];
    for (int i = threadIdx.x; i < N2; i += 256) W2s[N1 * N2 + i] = b.b ? b.b[i] : 0.f;
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;          // one row per thread (see dense_rows_kernel)
    if (row < a.M) {
        float x[K1];
        const float* r1 = a.x1 + (size_t)row * K1;
#pragma unroll
        for (int k = 0; k < K1; ++k) x[k] = r1[k];
        float h[N1];
#pragma unroll
        for (int n = 0; n < N1; ++n) h[n] = W1s[K1 * N1 + n];
#pragma unroll
        for (int k = 0; k < K1; ++k) {
#pragma unroll
            for (int n = 0; n < N1; ++n) h[n] = fmaf(x[k], W1s[k * N1 + n], h[n]);
            SSDR_SCHED_FENCE();
        }
        if (a.act) {
#pragma unroll
            for (int n = 0; n < N1; ++n) h[n] = lrelu(h[n]);
        }
        float4* ho = reinterpret_cast<float4*>(a.y + (size_t)row * N1);
#pragma unroll
        for (int q = 0; q < N1 / 4; ++q) ho[q] = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
        float acc[N2];
#pragma unroll
        for (int n = 0; n < N2; ++n) acc[n] = W2s[N1 * N2 + n];
#pragma unroll
        for (int k = 0; k < N1; ++k) {
#pragma unroll
            for (int n = 0; n < N2; ++n) acc[n] = fmaf(h[k], W2s[k * N2 + n], acc[n]);
            SSDR_SCHED_FENCE();
        }
        float4* yo = reinterpret_cast<float4*>(b.y + (size_t)row * (b.ldy ? b.ldy : N2));
        if (b.xyz) {      // the row's coordinates in front of its features: one 64-byte row of the level-0 gather table
            const int be = row / b.xyz_rows_per_batch, i = row - be * b.xyz_rows_per_batch;
            const float* c = b.xyz + (size_t)be * b.xyz_batch_stride + 3 * (size_t)i;
            yo[-1] = make_float4(c[0], c[1], c[2], 0.f);
        }
#pragma unroll
        for (int q = 0; q < N2 / 4; ++q) {
            float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            if (b.act) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
            yo[q] = v;
        }
    }
}

// fc1 (32 -> 64) + fc2 (64 -> 32 = last_second_features) + fc (32 -> C) + softmax (:174-178, :84) in one pass over the
// points: 128 bytes in, 128 + 4C bytes out per point instead of three round trips through HBM.
template <int C>
__global__ __launch_bounds__(256) void tail_kernel(const float* __restrict__ x, const float* __restrict__ W1, const float* __restrict__ b1,
                                                   const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ W3,
                                                   const float* __restrict__ b3, int M, float* __restrict__ feat32, float* __restrict__ probs) {
    __shared__ float S1[32 * 64 + 64], S2[64 * 32 + 32], S3[32 * C + C];
    for (int i = threadIdx.x; i < 32 * 64; i += 256) { S1[i] = W1[i]; S2[i] = W2[i]; }
    for (int i = threadIdx.x; i < 64; i += 256) S1[32 * 64 + i] = b1[i];
    for (int i = threadIdx.x; i < 32; i += 256) S2[64 * 32 + i] = b2[i];
    for (int i = threadIdx.x; i < 32 * C; i += 256) S3[i] = W3[i];
    for (int i = threadIdx.x; i < C; i += 256) S3[32 * C + i] = b3[i];
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;          // one row per thread (see dense_rows_kernel)
    if (row < M) {
        float xin[32], h1[64], f[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 v = reinterpret_cast<const float4*>(x + (size_t)row * 32)[q]; xin[4 * q] = v.x; xin[4 * q + 1] = v.y; xin[4 * q + 2] = v.z; xin[4 * q + 3] = v.w; }
#pragma unroll
        for (int n = 0; n < 64; ++n) h1[n] = S1[32 * 64 + n];
#pragma unroll
        for (int half = 0; half < 2; ++half)                 // 32 outputs at a time: 32 weights live per k
#pragma unroll
            for (int k = 0; k < 32; ++k) {
#pragma unroll
                for (int n = 0; n < 32; ++n) h1[32 * half + n] = fmaf(xin[k], S1[k * 64 + 32 * half + n], h1[32 * half + n]);
                SSDR_SCHED_FENCE();
            }
#pragma unroll
        for (int n = 0; n < 32; ++n) f[n] = S2[64 * 32 + n];
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const float hk = lrelu(h1[k]);
#pragma unroll
            for (int n = 0; n < 32; ++n) f[n] = fmaf(hk, S2[k * 32 + n], f[n]);
            SSDR_SCHED_FENCE();
        }
#pragma unroll
        for (int n = 0; n < 32; ++n) f[n] = lrelu(f[n]);
#pragma unroll
        for (int q = 0; q < 8; ++q) reinterpret_cast<float4*>(feat32 + (size_t)row * 32)[q] = make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
        float logit[C]; float m = -3.402823466e+38f;
#pragma unroll
        for (int c = 0; c < C; ++c) logit[c] = S3[32 * C + c];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
#pragma unroll
            for (int c = 0; c < C; ++c) logit[c] = fmaf(f[k], S3[k * C + c], logit[c]);
            SSDR_SCHED_FENCE();
        }
#pragma unroll
        for (int c = 0; c < C; ++c) m = fmaxf(m, logit[c]);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { logit[c] = expf(logit[c] - m); sum += logit[c]; }
#pragma unroll
        for (int c = 0; c < C; ++c) probs[(size_t)row * C + c] = logit[c] / sum;
    }
}

__global__ __launch_bounds__(256) void xyz_fill_kernel(DenseArgs a) {      // the coordinate columns of the gather table when the fused thin-layer kernel did not run
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= a.M) return;
    const int be = row / a.xyz_rows_per_batch, i = row - be * a.xyz_rows_per_batch;
    const float* c = a.xyz + (size_t)be * a.xyz_batch_stride + 3 * (size_t)i;
    float* yo = a.y + (size_t)row * (a.ldy ? a.ldy : a.N) - 4;
    yo[0] = c[0]; yo[1] = c[1]; yo[2] = c[2]; yo[3] = 0.f;
}
int launch_xyz_fill(const DenseArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(xyz_fill_kernel, dim3((unsigned)((a.M + 255) / 256)), dim3(256), 0, s, a);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

// ---- launchers ---------------------------------------------------------------------------------------
int launch_dense(const DenseArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return SSDR_OK;
    dim3 grid((unsigned)((a.M + DTM - 1) / DTM), (unsigned)((a.N + DTN - 1) / DTN));
    const bool vec = a.k1 % 4 == 0 && a.k2 % 4 == 0 && a.N % 4 == 0 && ((uintptr_t)a.x1 & 15) == 0 && ((uintptr_t)a.x2 & 15) == 0 && ((uintptr_t)a.W & 15) == 0;
    // thin layers over many rows: one row per lane (x2 / y rows must be 16-byte aligned, x1 too unless k1 % 4 != 0)
    const bool rows_form = a.M >= 1024 && ((uintptr_t)a.y & 15) == 0 && ((uintptr_t)a.x2 & 15) == 0 && (a.k1 % 4 != 0 || ((uintptr_t)a.x1 & 15) == 0) && a.k2 == 0 && a.N == 8 &&
                           (a.k1 == 6 || a.k1 == 8 || a.k1 == 16);
    // (the thin one-row-per-lane layers stream their rows on fp32 FMAs: charged by the bytes they move, 4 (k + N) per row)
    ProfScope prof(rows_form ? "dense_rows_kernel" : "dense_kernel", s, rows_form ? 4.0 * (double)a.M * (double)(a.k1 + a.N) : 2.0 * (double)a.M * (double)(a.k1 + a.k2) * (double)a.N,
                   rows_form ? 0.0 : 2.0 * (double)a.M * (double)(a.k1 + a.k2) * (double)a.N);
    if (rows_form) {
        const dim3 g((unsigned)((a.M + 255) / 256));
#define SSDR_ROWS(K1_, K2_, N_) if (a.k1 == K1_ && a.k2 == K2_ && a.N == N_) { hipLaunchKernelGGL((dense_rows_kernel<K1_, K2_, N_>), g, dim3(256), 0, s, a); SSDR_HIP(hipGetLastError()); return SSDR_OK; }
        SSDR_ROWS(6, 0, 8) SSDR_ROWS(8, 0, 8) SSDR_ROWS(16, 0, 8)      // wider outputs: register allocation degrades, the MFMA tile wins
#undef SSDR_ROWS
    }
    if (a.xyz) SSDR_TRY(launch_xyz_fill(a, s));
    if (vec && a.M <= 16384) {      // too few 128-row tiles to fill the chip
        dim3 gs((unsigned)((a.M + STM - 1) / STM), (unsigned)((a.N + DTN - 1) / DTN));
        hipLaunchKernelGGL(dense_small_kernel, gs, dim3(256), 0, s, a);
    } else if (vec) hipLaunchKernelGGL((dense_kernel<true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dense_kernel<false>), grid, dim3(256), 0, s, a);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

// fc0 and level 0's mlp1 (6 -> 8 -> 8) in one launch; SSDR_ERR_UNSUPPORTED (no error text) for any other pair
int launch_dense_rows2(const DenseArgs& a, const DenseArgs& b, hipStream_t s) {
    const bool ok = a.k1 == 6 && a.k2 == 0 && a.N == 8 && b.k1 == 8 && b.k2 == 0 && b.N == 8 && b.x1 == a.y && a.M == b.M && a.M >= 1024 && !a.ldy && !a.xyz && !a.idx2 && !b.idx2 &&
                    (((uintptr_t)a.y | (uintptr_t)b.y) & 15) == 0 && (!b.ldy || b.ldy % 4 == 0);
    if (!ok) return SSDR_ERR_UNSUPPORTED;
    ProfScope prof("dense_rows_kernel", s, 4.0 * (double)a.M * (6.0 + 8.0 + 8.0), 0.0);      // bytes: 6 inputs, 8 + 8 outputs per row
    hipLaunchKernelGGL((dense_rows2_kernel<6, 8, 8>), dim3((unsigned)((a.M + 255) / 256)), dim3(256), 0, s, a, b);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

template <int D> static int launch_lfa_d(const LfaArgs& a, bool second, int B, hipStream_t s) {
    using C = LfaCfg<D>;
    dim3 grid((unsigned)((a.n + C::PTS - 1) / C::PTS), (unsigned)B);
    static bool attr_done = false;   // more than 64 KiB of dynamic LDS needs the opt-in
    if (!attr_done) {
        SSDR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lfa_att_kernel<D, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        SSDR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lfa_att_kernel<D, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_done = true;
    }
    // algorithmic FLOPs: LocSE 10->h, (second half) h->h, attention d->d on n*16 neighbour rows, + the weighted sum
    const double rows = (double)B * (double)a.n * 16.0;
    // executed on the f32 MFMA: LocSE (K padded to 12), LFAmlp2, and the attention product (its position half only when G rows are gathered)
    const double exec = rows * (2.0 * 12 * C::H + (second ? 2.0 * C::H * C::H : 0.0) + (a.g ? 2.0 * C::H * D : 2.0 * D * D));
    ProfScope prof("lfa_att_kernel", s, rows * (2.0 * 10 * C::H + (second ? 2.0 * C::H * C::H : 0.0) + 2.0 * D * D + 2.0 * D), exec);
    if (second) hipLaunchKernelGGL((lfa_att_kernel<D, true>), grid, dim3(256), C::LDS_BYTES, s, a);
    else hipLaunchKernelGGL((lfa_att_kernel<D, false>), grid, dim3(256), C::LDS_BYTES, s, a);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int launch_lfa(int D, const LfaArgs& a, bool second, int B, hipStream_t s) {
    if (a.n <= 0 || B <= 0) return SSDR_OK;
    switch (D) {
        case 16: return launch_lfa_d<16>(a, second, B, s);
        case 64: return launch_lfa_d<64>(a, second, B, s);
        case 128: return launch_lfa_d<128>(a, second, B, s);
        case 256: return launch_lfa_d<256>(a, second, B, s);
        case 512: return launch_lfa_d<512>(a, second, B, s);
        default: set_error("d_out=%d is not supported (16, 64, 128, 256, 512)", D); return SSDR_ERR_UNSUPPORTED;
    }
}

int launch_gather_max(const float* f, const int* idx, int n_in, int n_out, int idx_rows, int C, float* out, int B, hipStream_t s) {
    if (n_out <= 0 || B <= 0) return SSDR_OK;
    const size_t total = (size_t)n_out * C;
    // random_sample (RandLANet.py:537-548): every input row read once, 16 indices + one output row per output point (fp32 here, SURVEY 8d counts bf16)
    ProfScope prof("gather_max_kernel", s, (double)B * (4.0 * (double)n_in * C + (double)n_out * (64.0 + 4.0 * C)));
    if (C % 4 == 0 && (size_t)n_in * C < (1ull << 30) && total / 4 < (1ull << 31) && (((uintptr_t)f | (uintptr_t)out | (uintptr_t)idx) & 15) == 0) {
        dim3 grid4((unsigned)std::max<size_t>(1, std::min<size_t>((total / 4 + 255) / 256, 4096)), (unsigned)B);
        hipLaunchKernelGGL(gather_max4_kernel, grid4, dim3(256), 0, s, f, idx, n_in, n_out, idx_rows, C, out);
        SSDR_HIP(hipGetLastError());
        return SSDR_OK;
    }
    dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((total + 255) / 256, 4096)), (unsigned)B);
    hipLaunchKernelGGL(gather_max_kernel, grid, dim3(256), 0, s, f, idx, n_in, n_out, idx_rows, C, out);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

// fused fc1 + fc2 + fc + softmax; returns SSDR_ERR_UNSUPPORTED (without setting an error) when C has no instantiation
int launch_tail(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                int M, int C, float* feat32, float* probs, hipStream_t s) {
    if (M <= 0) return SSDR_OK;
    if ((C != 13 && C != 8) || ((uintptr_t)x & 15) || ((uintptr_t)feat32 & 15)) return SSDR_ERR_UNSUPPORTED;
    ProfScope prof("tail_kernel", s, (double)M * 4.0 * (32 + 32 + C));
    const dim3 g((unsigned)((M + 255) / 256));
    if (C == 13) hipLaunchKernelGGL((tail_kernel<13>), g, dim3(256), 0, s, x, W1, b1, W2, b2, W3, b3, M, feat32, probs);
    else hipLaunchKernelGGL((tail_kernel<8>), g, dim3(256), 0, s, x, W1, b1, W2, b2, W3, b3, M, feat32, probs);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int launch_head(const float* x, const float* W, const float* b, int M, int C, float* probs, hipStream_t s) {
    if (M <= 0) return SSDR_OK;
    if (C > 32) { set_error("num_classes=%d > 32 is not supported", C); return SSDR_ERR_UNSUPPORTED; }
    dim3 grid((unsigned)std::max(1, std::min((M + 255) / 256, 4096)));
    hipLaunchKernelGGL(head_kernel, grid, dim3(256), 0, s, x, W, b, M, C, probs);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

}  // namespace ssdr
