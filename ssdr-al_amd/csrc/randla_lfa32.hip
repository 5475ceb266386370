// The K-expanded half of RandLA-Net's building_block (RandLANet.py:514-527, :529-535, :572-585) on v_mfma_f32_32x32x16_bf16,
// laid out so that everything that reduces over the 16 neighbours of a point stays inside one lane.
//
// A 32 x 32 accumulator tile holds its 16 registers of a lane in ONE column and 16 ROWS.  Rows are numbered so that the rows of a
// lane are the 16 neighbours of one point: row r of a tile = (point (r >> 2) & 1, neighbour (r & 3) + 4 (r >> 3)) of a pair of
// points, and register q of lane (column c, half h) is then (point h, neighbour q, channel c): softmax over the neighbours and the
// weighted sum (att_pooling, :578-581) are 16-term loops over a lane's own registers — no lane exchanges, all 64 lanes store.
//
// The chain  rel -> LocSE conv -> (LFAmlp2) -> attention scores  never leaves the registers either: a product computed TRANSPOSED,
// T = W^T X^T (A = weights, B = activations), leaves the 32 neighbour rows on the lanes and the channels in the registers, which is
// the operand shape of the next product in either role (cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's
// operand"): as B it continues the transposed chain (T2 = W2^T T1), as A it gives the plain orientation (X2 = T1^T W2, scores =
// T^T Wfc) with the channel back on the lane.  The k order inside a 16-wide step is permuted by that reuse (slot (h, j) is row
// 8 (j >> 2) + 4 h + (j & 3)); the weights are stored in that order by the host (randla_model.hip, permuted_pieces).
//
// relative_pos_encoding (:529-535) enters as 7 inputs instead of 10: [|d|, d, p, p_nbr] W = |d| w0 + d (w1..3 - w7..9) + p (w4..6 + w7..9)
// since p_nbr = p - d; hi and lo bf16 pieces of the 7 inputs fill the 16 k slots of ONE MFMA step, so the four products
// (hi + lo)(Whi + Wlo) of the LocSE conv are two instructions.
//
// Weights: the position half of the attention matrix (rows H..D of W, x log2 e) and LFAmlp2 are staged through LDS in chunks of
// <= 32 KB shared by the four waves of a workgroup (double buffered, one barrier per chunk); each wave owns one pair of points and
// all D output columns.  The neighbour half of the scores arrives as gathered rows of G = f W[0:H] (one dense launch per point).
#include "ssdr_internal.hpp"
#include "randla.hpp"
#include "randla_dev.hpp"
#include <cmath>

namespace ssdr {

#ifndef HIPEMU
typedef __amdgpu_buffer_rsrc_t rsrc32_t;
__device__ __forceinline__ rsrc32_t make_rsrc32(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); }
// per-lane byte offset + wave-uniform byte offset (an SGPR): one instruction, no address arithmetic per column tile
__device__ __forceinline__ float buf_load_s(rsrc32_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0)); }
#else
struct rsrc32_t { const char* p; unsigned bytes; };
static inline rsrc32_t make_rsrc32(const void* p, unsigned bytes) { return rsrc32_t{reinterpret_cast<const char*>(p), bytes}; }
static inline float buf_load_s(rsrc32_t r, unsigned voff, unsigned soff) {      // out-of-range raw buffer loads return 0, as on the hardware
    const unsigned long long o = (unsigned long long)voff + soff;
    return o + 4 <= r.bytes ? *reinterpret_cast<const float*>(r.p + o) : 0.f;
}
#endif

__device__ __forceinline__ u32x4 ld128g(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st128s(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

template <int D> struct Lfa32Cfg {
    static constexpr int H = D / 2;
    static constexpr int KS = H / 16;                                  // k steps of the products over H channels
    static constexpr int HT = H / 32;                                  // 32-channel tiles per half
    static constexpr int CT = D / 32;                                  // column tiles of the scores
    static constexpr int CCF = D < 8192 / H ? D : 8192 / H;            // attention columns per staged chunk
    static constexpr int CC2 = H < 8192 / H ? H : 8192 / H;            // LFAmlp2 output channels per staged chunk
    static constexpr int NF = D / CCF, N2 = H / CC2;
    static constexpr int RS = 2 * H + 16;                              // bytes per staged row: 16-byte slots rotate through the banks
    // waves per workgroup: each owns one pair of points and shares the staged chunk; 8 (two per SIMD) where the registers allow two waves per SIMD anyway
    static constexpr int nw(bool second) { return D >= 256 && !second ? 8 : 4; }
    static constexpr int tpw(bool second) { return D == 128 && second ? 4 : 1; }      // pairs of points a wave takes one after the other (d = 128, second half: the staged weights stay;
                                                                                      // the first half loses a wave per SIMD to the loop's registers: 68.5 -> 72.6 us, the second gains: 103.8 -> 93.7; two pairs 94.6, eight 106.7)
    static constexpr int SPR = RS / 16;                                // 16-byte slots per staged row (the last one is padding)
    static constexpr size_t buf_bytes(int terms) { return ((size_t)terms * CCF * RS + 4095) / 4096 * 4096; }      // whole 1 KB DMA blocks, the same number for each of the 4 waves
};

template <int TERMS> __device__ __forceinline__ f32x16 mma32_split(const u32x4 (&a)[TERMS], const u32x4 (&b)[TERMS], f32x16 c) {
    c = mfma32_bf16(a[0], b[0], c);
    if constexpr (TERMS == 2) { c = mfma32_bf16(a[1], b[0], c); c = mfma32_bf16(a[0], b[1], c); }
    return c;
}

// lrelu + bf16 pieces of an accumulator tile whose rows are channels: registers 8 s2 .. 8 s2 + 7 are the fragment of k step s2
template <int TERMS> __device__ __forceinline__ void acc_to_frags(const f32x16& v, u32x4 (&f0)[TERMS], u32x4 (&f1)[TERMS]) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        unsigned h, l;
        split_bf16(lrelu(v[2 * jj]), lrelu(v[2 * jj + 1]), h, l); f0[0][jj] = h; if constexpr (TERMS == 2) f0[1][jj] = l;
        split_bf16(lrelu(v[8 + 2 * jj]), lrelu(v[8 + 2 * jj + 1]), h, l); f1[0][jj] = h; if constexpr (TERMS == 2) f1[1][jj] = l;
    }
}

// bias of 32 output channels ch0.. in the two accumulator layouts
__device__ __forceinline__ f32x16 bias_rows(const float* b, int ch0, int lh) {       // channel = the register's row
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 t = *reinterpret_cast<const float4*>(b + ch0 + 8 * g + 4 * lh);
        v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
    }
    return v;
}
__device__ __forceinline__ f32x16 bias_cols(const float* b, int ch0, int lr) {       // channel = the lane's column
    const float t = b[ch0 + lr];
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = t;
    return v;
}

// the seven inputs of the reformulated position encoding of this lane's (point, neighbour) row as one operand fragment:
// k slots [hi0..hi6, lo0 | lo1..lo6, 0, 0]
template <int TERMS> __device__ __forceinline__ u32x4 rel_fragment_x(float px, float py, float pz, float qx, float qy, float qz, int lh) {
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    const float x[7] = {sqrtf(dx * dx + dy * dy + dz * dz), dx, dy, dz, px, py, pz};
    const unsigned p01 = pack_bf16(x[0], x[1]), p23 = pack_bf16(x[2], x[3]), p45 = pack_bf16(x[4], x[5]);
    u32x4 f;
    if constexpr (TERMS == 2) {
        const float r0 = x[0] - bf16_lo_f32(p01), r1 = x[1] - bf16_hi_f32(p01), r2 = x[2] - bf16_lo_f32(p23), r3 = x[3] - bf16_hi_f32(p23);
        const float r4 = x[4] - bf16_lo_f32(p45), r5 = x[5] - bf16_hi_f32(p45);
        const unsigned p6 = pack_bf16(x[6], r0);
        const float r6 = x[6] - bf16_lo_f32(p6);
        f[0] = lh ? pack_bf16(r1, r2) : p01; f[1] = lh ? pack_bf16(r3, r4) : p23; f[2] = lh ? pack_bf16(r5, r6) : p45; f[3] = lh ? 0u : p6;
    } else {
        f[0] = lh ? 0u : p01; f[1] = lh ? 0u : p23; f[2] = lh ? 0u : p45; f[3] = lh ? 0u : pack_bf16(x[6], 0.f);
    }
    return f;
}
template <int TERMS> __device__ __forceinline__ u32x4 rel_fragment(const float* xyz, int p, int j, int lh) {
    return rel_fragment_x<TERMS>(xyz[3 * (size_t)p], xyz[3 * (size_t)p + 1], xyz[3 * (size_t)p + 2], xyz[3 * (size_t)j], xyz[3 * (size_t)j + 1], xyz[3 * (size_t)j + 2], lh);
}

// softmax over the 16 registers (scores in base-2 units) and the weighted sum of f: sum_q f_q 2^(s_q - m) / sum_q 2^(s_q - m)
__device__ __forceinline__ float softmax_wsum(const f32x16& s, const float (&f)[16]) {
    float m = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
    m = fmaxf(m, fmaxf(fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11])), fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]))));
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const float e = fast_exp2(s[q] - m); den += e; num = fmaf(f[q], e, num); }
    return num * fast_rcp(den);
}

template <int D, bool SECOND, int TERMS>
__global__ __launch_bounds__(Lfa32Cfg<D>::nw(SECOND) * 64) SSDR_WAVES_PER_EU((D >= 256 && !SECOND) ? 2 : 1) void lfa32_kernel(Lfa32Args a) {
    using C = Lfa32Cfg<D>;
    constexpr int NW = C::nw(SECOND), NT = NW * 64;
    constexpr int H = C::H, KS = C::KS, HT = C::HT, CT = C::CT, CCF = C::CCF, CC2 = C::CC2, NF = C::NF, N2 = SECOND ? C::N2 : 0, RS = C::RS;
    constexpr int NCH = N2 + NF;
    // d = 128: at most two chunks, i.e. every weight stays in the two buffers once staged: a wave takes TPW pairs of points one after the other and only the
    // first passes the barriers (5120 workgroups of four pairs each re-staged 32-48 KB for 8 points)
    constexpr int TPW = C::tpw(SECOND);
    static_assert(TPW == 1 || NCH <= 2, "several pairs per wave only where the staged weights stay");
    constexpr size_t BUFB = C::buf_bytes(TERMS);
    constexpr bool OFFREG = H <= 64;            // both gather offset tables in registers
    constexpr bool W1REG = H <= 64;             // LocSE weight fragments in registers
    __shared__ __attribute__((aligned(16))) char ldsA[BUFB];
    __shared__ __attribute__((aligned(16))) char ldsB[NCH > 1 ? BUFB : 16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    int bx, b; xcd_tile_map(bx, b);
    const int n = a.n;
    const float* xyz = a.xyz + (size_t)b * a.xyz_batch_stride;
    const int* neigh = a.neigh + (size_t)b * n * 16;

    // ---- weight chunks through LDS: image [term][row][RS].  Fetched into registers while the previous chunk is multiplied, stored behind it.
    // (LDS DMA — global_load_lds_dwordx4, no staging registers — was built and measured slower: the compiler makes the first LDS read behind
    // an outstanding DMA wait for vmcnt(0), the chunk in flight and every gather with it, whatever LDS object the DMA writes.)
    constexpr int MAXP = (TERMS * CCF * (H / 8) + NT - 1) / NT;
    u32x4 stg[MAXP];
    auto stage_load = [&](int i) {
        const bool is2 = SECOND && i < N2;
        const int rows = is2 ? CC2 : CCF, per = rows * (H / 8);
        const size_t base = is2 ? (size_t)i * CC2 * H : (size_t)(i - N2) * CCF * H;
        const uint16_t* hi = (is2 ? a.w2_hi : a.fc_hi) + base; const uint16_t* lo = (is2 ? a.w2_lo : a.fc_lo) + base;
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const int e = tid + NT * p;
            if (e < TERMS * per) { const int t = e >= per ? 1 : 0; stg[p] = ld128g((t ? lo : hi) + (size_t)(e - t * per) * 8); }
        }
    };
    auto stage_store = [&](int i, char* buf) {
        const bool is2 = SECOND && i < N2;
        const int rows = is2 ? CC2 : CCF, per = rows * (H / 8);
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const int e = tid + NT * p;
            if (e < TERMS * per) { const int t = e >= per ? 1 : 0, r = e - t * per; st128s(buf + (size_t)t * rows * RS + (size_t)(r / (H / 8)) * RS + (r % (H / 8)) * 16, stg[p]); }
        }
    };
    stage_load(0);
    // LocSE weights (per wave, once)
    u32x4 w1r[W1REG ? HT : 1][TERMS];
    auto w1_frag = [&](int t, int term) { return ld128g(a.w1p + ((size_t)(32 * t + lr) * 2 + term) * 16 + 8 * lh); };
    if constexpr (W1REG) {
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int term = 0; term < TERMS; ++term) w1r[t][term] = w1_frag(t, term);
    }
    auto w1_get = [&](int t, u32x4 (&f)[TERMS]) {
#pragma unroll
        for (int term = 0; term < TERMS; ++term) { if constexpr (W1REG) f[term] = w1r[t][term]; else f[term] = w1_frag(t, term); }
    };
    const rsrc32_t rG = make_rsrc32(a.g + (size_t)b * n * D, (unsigned)n * D * 4u);
    const rsrc32_t rF = make_rsrc32(a.fin + (size_t)b * n * H, (unsigned)n * H * 4u);

    for (int it = 0; it < TPW; ++it) {
    const int rt = (bx * NW + w) * TPW + it;
    const bool active = 2 * rt < n;             // wave-uniform: a wave past the end still stages weights and meets the barriers (first pass)
    if (it > 0 && !active) break;

    // ---- this lane's row of the position encoding, and its point's neighbour table -----------------------------------------------
    u32x4 relf = {0u, 0u, 0u, 0u};
    unsigned goff[16], foff[OFFREG ? 16 : 1];
    const int pc = min(2 * rt + lh, n - 1);                    // the point whose 16 neighbours this lane's registers hold (column layout)
    {
        const int pr = min(2 * rt + ((lr >> 2) & 1), n - 1);   // the point of this lane's row (operand layout)
        const int j = neigh[(size_t)pr * 16 + (lr & 3) + 4 * (lr >> 3)];
        relf = rel_fragment<TERMS>(xyz, pr, j, lh);
        const int4* nb4 = reinterpret_cast<const int4*>(neigh + (size_t)pc * 16);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int4 v = nb4[g4];
            const int nb[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                goff[4 * g4 + r] = (unsigned)nb[r] * (unsigned)(D * 4) + (unsigned)lr * 4u;
                if constexpr (OFFREG) foff[4 * g4 + r] = (unsigned)nb[r] * (unsigned)(H * 4) + (unsigned)lr * 4u;
            }
        }
    }
    auto f_off = [&](int q) -> unsigned { if constexpr (OFFREG) return foff[q]; else return (goff[q] + (unsigned)lr * 4u) >> 1; };

    // ---- T1 = lrelu(W1^T rel^T + b1): channels in the registers, neighbour rows on the lanes (LFAmlp1, :518) ----------------------------
    u32x4 Tf[KS][TERMS];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        u32x4 wf[TERMS]; w1_get(t, wf);
        f32x16 acc = bias_rows(a.b1, 32 * t, lh);
        acc = mfma32_bf16(wf[0], relf, acc);
        if constexpr (TERMS == 2) acc = mfma32_bf16(wf[1], relf, acc);
        acc_to_frags<TERMS>(acc, Tf[2 * t], Tf[2 * t + 1]);
    }
    auto x1_tile = [&](int t, float (&x)[16]) {               // the same product in the plain orientation: channel on the lane
        u32x4 wf[TERMS]; w1_get(t, wf);
        f32x16 acc = bias_cols(a.b1, 32 * t, lr);
        acc = mfma32_bf16(relf, wf[0], acc);
        if constexpr (TERMS == 2) acc = mfma32_bf16(relf, wf[1], acc);
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = lrelu(acc[q]);
    };

    if (it == 0) stage_store(0, ldsA);
    float x2[SECOND ? HT : 1][16];
    u32x4 T2f[SECOND ? KS : 1][TERMS];
    auto score_frag = [&](int s) -> const u32x4 (&)[TERMS] { if constexpr (SECOND) return T2f[s]; else return Tf[s]; };

    // first gathered rows of G (in flight across the LFAmlp2 pass)
    float gn[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) gn[q] = buf_load_s(rG, goff[q], 0u);

    // weight fragments out of a staged image, two k steps ahead of the products that use them (an LDS read issued right in front of its
    // product exposes its ~100 cycles on every k step: the compiler's own schedule, half the matrix rate)
    auto wfrag = [&](const char* img, int rows, int tile, int s, u32x4 (&f)[TERMS]) {
#pragma unroll
        for (int term = 0; term < TERMS; ++term) f[term] = ld128g(img + (size_t)term * rows * RS + (size_t)(32 * tile + lr) * RS + (16 * s + 8 * lh) * 2);
    };

#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        char* buf = (i & 1) ? ldsB : ldsA;
        if (it == 0) {
            __syncthreads();                                   // chunk i is in place; everybody is done with the other buffer
            if (i + 1 < NCH) stage_load(i + 1);
        }
        if (SECOND && i < N2) {
            // ---- f_xyz <- lrelu(f_xyz W2 + b2) (LFAmlp2, :523) in both orientations from the same staged fragments ---------------------
            if (active) {
#pragma unroll
                for (int tt = 0; tt < CC2 / 32; ++tt) {
                    const int t2 = i * (CC2 / 32) + tt;
                    f32x16 accT = bias_rows(a.b2, 32 * t2, lh), accX = bias_cols(a.b2, 32 * t2, lr);
                    u32x4 wa[TERMS], wb[TERMS];
                    wfrag(buf, CC2, tt, 0, wa);
#pragma unroll
                    for (int s = 0; s < KS; s += 2) {              // (the scheduler sinks an LDS read to its first use: fences keep the next step's reads ahead of this step's products)
                        wfrag(buf, CC2, tt, s + 1, wb);
                        SSDR_SCHED_FENCE();
                        accT = mma32_split<TERMS>(wa, Tf[s], accT);
                        accX = mma32_split<TERMS>(Tf[s], wa, accX);
                        SSDR_SCHED_FENCE();
                        if (s + 2 < KS) wfrag(buf, CC2, tt, s + 2, wa);
                        SSDR_SCHED_FENCE();
                        accT = mma32_split<TERMS>(wb, Tf[s + 1], accT);
                        accX = mma32_split<TERMS>(Tf[s + 1], wb, accX);
                        SSDR_SCHED_FENCE();
                    }
                    if constexpr (SECOND) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) x2[t2][q] = lrelu(accX[q]);
                        acc_to_frags<TERMS>(accT, T2f[2 * t2], T2f[2 * t2 + 1]);
                    }
                }
            }
        } else {
            // ---- attention scores (:578), softmax over the 16 neighbours (:579), weighted sum (:580-581) ---------------------------------
            if (active) {
#pragma unroll
                for (int uu = 0; uu < CCF / 32; ++uu) {
                    const int u = (i - N2) * (CCF / 32) + uu;
                    u32x4 wa[TERMS], wb[TERMS];
                    wfrag(buf, CCF, uu, 0, wa);
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = gn[q];
                    if (u + 1 < CT) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) gn[q] = buf_load_s(rG, goff[q], (unsigned)(u + 1) * 128u);
                    }
                    float fv[16];
                    if (u < HT) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) fv[q] = buf_load_s(rF, f_off(q), (unsigned)u * 128u);
                    }
                    static_assert(KS % 2 == 0, "k steps in pairs");
                    SSDR_SCHED_FENCE();                            // the gathers above stay above: sunk to their use they cost a trip to L2 per column tile
#pragma unroll
                    for (int s = 0; s < KS; s += 2) {              // the fragments of the next k step are requested before this step's products
                        wfrag(buf, CCF, uu, s + 1, wb);
                        SSDR_SCHED_FENCE();
                        acc = mma32_split<TERMS>(score_frag(s), wa, acc);
                        SSDR_SCHED_FENCE();
                        if (s + 2 < KS) wfrag(buf, CCF, uu, s + 2, wa);
                        SSDR_SCHED_FENCE();
                        acc = mma32_split<TERMS>(score_frag(s + 1), wb, acc);
                        SSDR_SCHED_FENCE();
                    }
                    if (u >= HT) {
                        if constexpr (SECOND) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) fv[q] = x2[u - HT][q];
                        } else x1_tile(u - HT, fv);
                    }
                    const float v = softmax_wsum(acc, fv);
                    if (2 * rt + lh < n) a.out[((size_t)b * n + (size_t)(2 * rt + lh)) * D + 32 * u + lr] = v;
                }
            }
        }
        if (it == 0 && i + 1 < NCH) stage_store(i + 1, (i & 1) ? ldsA : ldsB);
    }
    }      // pairs of this wave
}

// ---- levels whose weights fit in LDS whole (d = 64, 128): no G table ----------------------------------------------------------------------------
// What bounds the gathering levels is the vector-memory return path: a gather-type load costs the CU ~10 cycles per dword of a wave instruction
// (PMC: TD busy 80-98 % of these kernels' time, the same with a third of the arithmetic), i.e. ~25-30 bytes per clock from L2 — the rate
// MI355X_MICROARCH.md quotes for row gathers.  So the kernel fetches each neighbour row ONCE, in the operand layout: lane (row r, half h) reads
// the 8 features of k step s of ITS neighbour straight into the A fragment of the neighbour half of the scores (multiplied here: no G rows to
// gather, no G launch), and the same registers go through a wave-private LDS image [32 rows][H] from which the weighted sum reads them back in
// the accumulator layout (channel on the lane, neighbour in the register).  Per pair of points: 1 + 6 + H / 2 dwords per lane instead of
// 23 + 24 (H / 16).
// (Chaining the per-point convs behind the pool into this kernel — att_pooling's mlp, then lrelu(mlp2 + shortcut), on the workgroup's 32 or 64 pooled rows out
// of LDS — was built and measured at d = 64: the three dense launches it removes cost 87 us, the epilogues 60-110 us: a workgroup re-reads the layers' weights
// for 32-64 rows where the tiled dense kernel reads them for 128.  Not kept.)
template <int D> struct Lfa32ResCfg {
    static constexpr int H = D / 2, RS = 2 * H + 16;
    static constexpr int NW = 4;                                       // waves per workgroup (they share the staged weights)
    static constexpr int RPW = 8;                                      // pairs of points per wave
    static constexpr int FRS = H * 4 + 16;                             // bytes per row of the feature image
    static constexpr size_t w_bytes(int terms, bool second) { return (size_t)terms * ((second ? H : 0) + 2 * D) * RS; }
    static constexpr size_t lds_bytes(int terms, bool second) { return w_bytes(terms, second) + (size_t)NW * 32 * FRS; }
};

template <int D, bool SECOND, int TERMS>
__global__ __launch_bounds__(Lfa32ResCfg<D>::NW * 64) void lfa32_res_kernel(Lfa32Args a) {
    using C = Lfa32ResCfg<D>;
    constexpr int H = C::H, KS = H / 16, HT = H / 32, CT = D / 32, RS = C::RS, RPW = C::RPW, NW = C::NW, FRS = C::FRS;
    SSDR_DYN_SHARED(float, smem);
    char* lds = reinterpret_cast<char*>(smem);
    char* const img2 = lds;                                                    // LFAmlp2         [TERMS][H rows][RS], k permuted
    char* const imgp = lds + (SECOND ? (size_t)TERMS * H * RS : 0);            // attention, position half [TERMS][D rows][RS], k permuted
    char* const imgn = imgp + (size_t)TERMS * D * RS;                          // attention, neighbour half [TERMS][D rows][RS], natural k
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    char* const fimg = lds + C::w_bytes(TERMS, SECOND) + (size_t)w * 32 * FRS; // this wave's gathered features [32 rows][H] f32
    const int lr = lane & 31, lh = lane >> 5;
    int bx, b; xcd_tile_map(bx, b);
    const int n = a.n;
    const float* xyz = a.xyz + (size_t)b * a.xyz_batch_stride;
    const int* neigh = a.neigh + (size_t)b * n * 16;
    const float* finb = a.fin + (size_t)b * n * H;
    {   // weights -> LDS (16-byte pieces; rows of H bf16 are contiguous in the source: fc = [position half [D][H] | neighbour half [D][H]])
        constexpr int P8 = H / 8, N2P = SECOND ? TERMS * H * P8 : 0, NFP = TERMS * 2 * D * P8;
        for (int e = tid; e < N2P + NFP; e += NW * 64) {
            const bool is2 = e < N2P;
            const int r = is2 ? e : e - N2P, rows = is2 ? H : 2 * D, per = rows * P8, t = r >= per ? 1 : 0, q = r - t * per;
            const uint16_t* src = (is2 ? (t ? a.w2_lo : a.w2_hi) : (t ? a.fc_lo : a.fc_hi)) + (size_t)q * 8;
            const int row = q / P8;
            char* dst = is2 ? img2 + (size_t)t * H * RS + (size_t)row * RS : (row < D ? imgp + (size_t)t * D * RS + (size_t)row * RS : imgn + (size_t)t * D * RS + (size_t)(row - D) * RS);
            st128s(dst + (q % P8) * 16, ld128g(src));
        }
    }
    u32x4 w1r[HT][TERMS];
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int term = 0; term < TERMS; ++term) w1r[t][term] = ld128g(a.w1p + ((size_t)(32 * t + lr) * 2 + term) * 16 + 8 * lh);
    const int nbrrow = (lr & 3) + 4 * (lr >> 3), prow = (lr >> 2) & 1;
    __syncthreads();                                     // the weights are in place (no workgroup barrier below: waves run at their own pace)

    const int rt0 = ((int)bx * NW + w) * RPW;
    for (int it = 0; it < RPW; ++it) {
        const int rt = rt0 + it;
        if (2 * rt >= n) break;                          // wave-uniform
        // this lane's (point, neighbour) row: index, coordinates, and its share of the neighbour's features (k = 16 s + 8 h ..)
        const int pr = min(2 * rt + prow, n - 1);
        const int j = neigh[(size_t)pr * 16 + nbrrow];
        const float* fj = finb + (size_t)j * H + 8 * lh;
        float4 fa[KS][2];
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) { fa[s2][0] = *reinterpret_cast<const float4*>(fj + 16 * s2); fa[s2][1] = *reinterpret_cast<const float4*>(fj + 16 * s2 + 4); }
        const u32x4 relf = rel_fragment<TERMS>(xyz, pr, j, lh);
        u32x4 fnf[KS][TERMS];
        wave_sync();                                     // the previous pair's reads of the feature image are done
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
            float4* dst = reinterpret_cast<float4*>(fimg + (size_t)lr * FRS + (16 * s2 + 8 * lh) * 4);
            dst[0] = fa[s2][0]; dst[1] = fa[s2][1];
            const float f[8] = {fa[s2][0].x, fa[s2][0].y, fa[s2][0].z, fa[s2][0].w, fa[s2][1].x, fa[s2][1].y, fa[s2][1].z, fa[s2][1].w};
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) { unsigned hh, ll; split_bf16(f[2 * k2], f[2 * k2 + 1], hh, ll); fnf[s2][0][k2] = hh; if constexpr (TERMS == 2) fnf[s2][1][k2] = ll; }
        }
        wave_sync();
        u32x4 Tf[KS][TERMS];
#pragma unroll
        for (int t = 0; t < HT; ++t) {                   // T1 = lrelu(W1^T rel^T + b1) (LFAmlp1, :518)
            f32x16 acc = bias_rows(a.b1, 32 * t, lh);
            acc = mfma32_bf16(w1r[t][0], relf, acc);
            if constexpr (TERMS == 2) acc = mfma32_bf16(w1r[t][1], relf, acc);
            acc_to_frags<TERMS>(acc, Tf[2 * t], Tf[2 * t + 1]);
        }
        float x2[SECOND ? HT : 1][16];
        if constexpr (SECOND) {                          // f_xyz <- lrelu(f_xyz W2 + b2) (LFAmlp2, :523), both orientations
            u32x4 T2f[KS][TERMS];
#pragma unroll
            for (int t2 = 0; t2 < HT; ++t2) {
                f32x16 accT = bias_rows(a.b2, 32 * t2, lh), accX = bias_cols(a.b2, 32 * t2, lr);
                u32x4 wa[TERMS], wb[TERMS];
                auto w2f = [&](int s2, u32x4 (&f)[TERMS]) {
#pragma unroll
                    for (int term = 0; term < TERMS; ++term) f[term] = ld128g(img2 + (size_t)term * H * RS + (size_t)(32 * t2 + lr) * RS + (16 * s2 + 8 * lh) * 2);
                };
                w2f(0, wa);
#pragma unroll
                for (int s2 = 0; s2 < KS; s2 += 2) {
                    w2f(s2 + 1, wb);
                    accT = mma32_split<TERMS>(wa, Tf[s2], accT);
                    accX = mma32_split<TERMS>(Tf[s2], wa, accX);
                    if (s2 + 2 < KS) w2f(s2 + 2, wa);
                    accT = mma32_split<TERMS>(wb, Tf[s2 + 1], accT);
                    accX = mma32_split<TERMS>(Tf[s2 + 1], wb, accX);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) x2[t2][q] = lrelu(accX[q]);
                acc_to_frags<TERMS>(accT, T2f[2 * t2], T2f[2 * t2 + 1]);
            }
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
                for (int term = 0; term < TERMS; ++term) Tf[s2][term] = T2f[s2][term];
        }
#pragma unroll
        for (int u = 0; u < CT; ++u) {                   // scores (:578), softmax over the neighbours (:579), weighted sum (:580-581)
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            float fv[16];
            if (u < HT) {                                // the gathered features back in the accumulator layout: row (half lh, neighbour q), channel 32 u + lr
#pragma unroll
                for (int q = 0; q < 16; ++q) fv[q] = *reinterpret_cast<const float*>(fimg + (size_t)((q & 3) + 8 * (q >> 2) + 4 * lh) * FRS + (32 * u + lr) * 4);
            }
            u32x4 wa[2][TERMS], wb[2][TERMS];            // [neighbour half, position half]
            auto wsf = [&](int s2, u32x4 (&f)[2][TERMS]) {
#pragma unroll
                for (int term = 0; term < TERMS; ++term) {
                    f[0][term] = ld128g(imgn + (size_t)term * D * RS + (size_t)(32 * u + lr) * RS + (16 * s2 + 8 * lh) * 2);
                    f[1][term] = ld128g(imgp + (size_t)term * D * RS + (size_t)(32 * u + lr) * RS + (16 * s2 + 8 * lh) * 2);
                }
            };
            wsf(0, wa);
#pragma unroll
            for (int s2 = 0; s2 < KS; s2 += 2) {         // (fences that keep the next k step's LDS reads ahead of this step's products were measured slower here: 134 / 152 us against 117 / 141)
                wsf(s2 + 1, wb);
                acc = mma32_split<TERMS>(fnf[s2], wa[0], acc);
                acc = mma32_split<TERMS>(Tf[s2], wa[1], acc);
                if (s2 + 2 < KS) wsf(s2 + 2, wa);
                acc = mma32_split<TERMS>(fnf[s2 + 1], wb[0], acc);
                acc = mma32_split<TERMS>(Tf[s2 + 1], wb[1], acc);
            }
            if (u < HT) {
            } else if constexpr (SECOND) {
#pragma unroll
                for (int q = 0; q < 16; ++q) fv[q] = x2[u - HT][q];
            } else {
                f32x16 ax = bias_cols(a.b1, 32 * (u - HT), lr);
                ax = mfma32_bf16(relf, w1r[u - HT][0], ax);
                if constexpr (TERMS == 2) ax = mfma32_bf16(relf, w1r[u - HT][1], ax);
#pragma unroll
                for (int q = 0; q < 16; ++q) fv[q] = lrelu(ax[q]);
            }
            const float v = softmax_wsum(acc, fv);
            if (2 * rt + lh < n) a.out[((size_t)b * n + (size_t)(2 * rt + lh)) * D + 32 * u + lr] = v;
        }
    }
}

// ---- level 0 (d = 16, h = 8): two pairs of points per tile ----------------------------------------------------------------------------
// With 8 position channels the k dimension of a 16-deep MFMA step holds TWO independent problems: k slots 0..7 (the lanes of half 0)
// carry the 8 inputs of point pair A, slots 8..15 those of pair B, and the weight operand is block diagonal — columns 0..15 are pair A's 16
// output channels and see only slots 0..7, columns 16..31 are pair B's.  One 32 x 32 tile is then 4 points x 16 neighbours x 16 channels
// with every lane busy: lane (column c, half h) register q = (point 2 (c >> 4) + h, neighbour q, channel c & 15).  The operand fragments
// are lane-dependent constants of the layer, tabulated by the host (randla_model.hip, level0_tables): [fragment][64 lanes][8 bf16].
// LocSE slot 7 carries the constant 1 against the bias row; the neighbour-feature columns (c & 15 < 8) of the "plain" product have zero
// weights and start from the neighbour's feature, so one accumulator holds [f_nbr | f_xyz] in the layout of the weighted sum.
// As above the neighbour half of the scores is multiplied here and every neighbour row is fetched once, by the lane that owns the row: the level
// keeps ONE gather table with a point's coordinates and features in the same 64-byte row [x y z 0 | f0..f7 | -] (written by the thin layers that
// produce the features, randla_kernels.hip), a lane reads 48 bytes of its neighbour's row, and the accumulator layout of the features comes
// back out of a wave-private LDS image (64 rows of 32 bytes; the position columns read a block of zeros).
constexpr int L0_TPW = 8;          // tiles (of 4 points) per wave
constexpr int L0_IMG = 64 * 32 + 32, L0_ZERO = 1024, L0_WAVE = L0_IMG + L0_ZERO;      // bytes of LDS per wave

template <bool SECOND, int TERMS>
__global__ __launch_bounds__(256) void lfa32_l0_kernel(Lfa32Args a) {
    __shared__ __attribute__((aligned(16))) char l0lds[4 * L0_WAVE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5, cc = lr & 15, sg = lr >> 4;
    int bx, b; xcd_tile_map(bx, b);
    const int n = a.n;
    const int* neigh = a.neigh + (size_t)b * n * 16;
    const float* tabb = a.fin + (size_t)b * n * 16;                    // gather table: one 64-byte row per point, [x y z 0 | f0..f7 | -]
    char* const img = l0lds + w * L0_WAVE;
    for (int i = lane; i < L0_ZERO / 4; i += 64) reinterpret_cast<float*>(img + L0_IMG)[i] = 0.f;
    auto tab = [&](const uint16_t* t, int f) { return ld128g(t + ((size_t)f * 64 + lane) * 8); };
    u32x4 locT[TERMS], locX[TERMS], w2T[SECOND ? TERMS : 1], w2X[SECOND ? TERMS : 1], fcB[TERMS], fcN[TERMS];
#pragma unroll
    for (int t = 0; t < TERMS; ++t) {
        locT[t] = tab(a.w1p, t); locX[t] = tab(a.w1p, 2 + t);
        fcB[t] = tab(t ? a.fc_lo : a.fc_hi, 0); fcN[t] = tab(t ? a.fc_lo : a.fc_hi, 1);
        if constexpr (SECOND) { w2T[t] = tab(t ? a.w2_lo : a.w2_hi, 0); w2X[t] = tab(t ? a.w2_lo : a.w2_hi, 1); }
    }
    u32x4 onesA = {0u, 0u, 0u, 0u}, biasB = {0u, 0u, 0u, 0u};      // SECOND: the plain product's bias as a fourth product (1, 1 | hi, lo)
    float b2r[8];
    if constexpr (SECOND) {
        if (lh == 0) { onesA[0] = 0x3f803f80u; if (cc >= 8) { const float bv = a.b2[cc - 8]; unsigned h, l; split_bf16(bv, 0.f, h, l); biasB[0] = (h & 0xffffu) | (TERMS == 2 ? (l << 16) : 0u); } }
#pragma unroll
        for (int q = 0; q < 8; ++q) b2r[q] = a.b2[q];
    }
    const float slope = cc < 8 ? 1.f : 0.2f;
    float* out = a.out + (size_t)b * n * 16;
    const int g0 = ((int)bx * 4 + w) * L0_TPW;
    const int nbrrow = (lr & 3) + 4 * (lr >> 3), prow = 2 * lh + ((lr >> 2) & 1), pcol = 2 * sg + lh;
    // the image row of lane L sits at 32 L + 32 (L >> 5): the two halves of the wave on different banks
    char* const wrow = img + lane * 32 + lh * 32;
    // accumulator layout: register q of this lane = (point pcol, neighbour q) = the row owned by lane 32 (pcol >> 1) + 4 (pcol & 1) + (q & 3) + 8 (q >> 2)
    const char* const rbase = cc < 8 ? img + (32 * (pcol >> 1) + 4 * (pcol & 1)) * 32 + (pcol >> 1) * 32 + cc * 4 : img + L0_IMG;
    for (int it = 0; it < L0_TPW; ++it) {
        const int t = g0 + it;
        if (4 * t >= n) break;                                         // wave-uniform
        const int pr = min(4 * t + prow, n - 1);
        const int j = neigh[(size_t)pr * 16 + nbrrow];
        const float4 pp = *reinterpret_cast<const float4*>(tabb + (size_t)pr * 16);
        const float4* rowj = reinterpret_cast<const float4*>(tabb + (size_t)j * 16);
        const float4 qq = rowj[0], fa0 = rowj[1], fa1 = rowj[2];
        wave_sync();                                                   // the previous tile's reads of the image are done
        reinterpret_cast<float4*>(wrow)[0] = fa0; reinterpret_cast<float4*>(wrow)[1] = fa1;
        // this lane's (point, neighbour) row: position encoding [x0..x6, 1] and the neighbour's features, as hi / lo operand fragments
        u32x4 rel[TERMS], fnA[TERMS];
        {
            const float dx = pp.x - qq.x, dy = pp.y - qq.y, dz = pp.z - qq.z;
            const float v[8] = {sqrtf(dx * dx + dy * dy + dz * dz), dx, dy, dz, pp.x, pp.y, pp.z, 1.f};
            const float f[8] = {fa0.x, fa0.y, fa0.z, fa0.w, fa1.x, fa1.y, fa1.z, fa1.w};
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                unsigned h, l;
                split_bf16(v[2 * k2], v[2 * k2 + 1], h, l); rel[0][k2] = h; if constexpr (TERMS == 2) rel[1][k2] = l;
                split_bf16(f[2 * k2], f[2 * k2 + 1], h, l); fnA[0][k2] = h; if constexpr (TERMS == 2) fnA[1][k2] = l;
            }
        }
        wave_sync();
        f32x16 accX;
#pragma unroll
        for (int q = 0; q < 16; ++q) accX[q] = *reinterpret_cast<const float*>(rbase + ((q & 3) + 8 * (q >> 2)) * 32);
        f32x16 accS, accT;
#pragma unroll
        for (int q = 0; q < 16; ++q) { accS[q] = 0.f; accT[q] = 0.f; }
        accS = mma32_split<TERMS>(fnA, fcN, accS);                    // neighbour half of the scores
        // T1 = lrelu(W1^T rel^T) (bias through slot 7): rows 0..15 = (pair, channel), the rest of the tile is idle
        accT = mma32_split<TERMS>(locT, rel, accT);
        u32x4 Tf[TERMS];
        auto to_frag = [&](const f32x16& v, u32x4 (&f)[TERMS]) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { unsigned h, l; split_bf16(lrelu(v[2 * jj]), lrelu(v[2 * jj + 1]), h, l); f[0][jj] = h; if constexpr (TERMS == 2) f[1][jj] = l; }
        };
        to_frag(accT, Tf);
        if constexpr (SECOND) {
            // f_xyz <- lrelu(f_xyz W2 + b2) (LFAmlp2, :523) in both orientations
            f32x16 acc2;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc2[q] = q < 8 ? b2r[q] : 0.f;
            acc2 = mma32_split<TERMS>(w2T, Tf, acc2);
            accX = mfma32_bf16(onesA, biasB, accX);
            accX = mma32_split<TERMS>(Tf, w2X, accX);
            to_frag(acc2, Tf);
        } else {
            accX = mma32_split<TERMS>(rel, locX, accX);
        }
        float fv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) fv[q] = fmaxf(accX[q], accX[q] * slope);
        accS = mma32_split<TERMS>(Tf, fcB, accS);
        const float v = softmax_wsum(accS, fv);
        const int pcu = 4 * t + pcol;
        if (pcu < n) out[(size_t)pcu * 16 + cc] = v;
    }
}

static int launch_lfa32_l0(const Lfa32Args& a, bool second, int B, int prec, hipStream_t s) {
    dim3 grid((unsigned)((a.n + 16 * L0_TPW - 1) / (16 * L0_TPW)), (unsigned)B);
    const int terms = prec == PREC_BF16X3 ? 2 : 1;
    const double rows = (double)B * (double)a.n * 16.0, np = terms == 2 ? 3.0 : 1.0;
    // executed: every product is a 32 x 32 x 16 tile for 4 points (LocSE in both orientations, LFAmlp2 in both + its bias product, the two halves of the scores)
    const double exec = (double)B * std::ceil(a.n / 4.0) * 2.0 * 32 * 32 * 16 * (np * (second ? 5.0 : 4.0) + (second ? 1.0 : 0.0));
    ProfScope prof("lfa32_l0_kernel", s, rows * (2.0 * 10 * 8 + (second ? 2.0 * 8 * 8 : 0.0) + 2.0 * 16 * 16 + 2.0 * 16), exec);
    if (terms == 2) {
        if (second) hipLaunchKernelGGL((lfa32_l0_kernel<true, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((lfa32_l0_kernel<false, 2>), grid, dim3(256), 0, s, a);
    } else {
        if (second) hipLaunchKernelGGL((lfa32_l0_kernel<true, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((lfa32_l0_kernel<false, 1>), grid, dim3(256), 0, s, a);
    }
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

template <int D> static int launch_lfa32_d(const Lfa32Args& a, bool second, int B, int prec, hipStream_t s) {
    using C = Lfa32Cfg<D>;
    const int nw = C::nw(second);
    dim3 grid((unsigned)((a.n + 2 * nw * C::tpw(second) - 1) / (2 * nw * C::tpw(second))), (unsigned)B);
    const int terms = prec == PREC_BF16X3 ? 2 : 1;
    const double rows = (double)B * (double)a.n * 16.0;       // algorithmic FLOPs of the reference's formulation (as lfa_att_kernel)
    // executed on the matrix cores: LocSE with K padded to 16 in both orientations (two instructions carry the four products),
    // LFAmlp2 in both orientations and the position half of the attention product (one or three bf16 products)
    const double np = terms == 2 ? 3.0 : 1.0;
    const double exec = rows * (2.0 * 2.0 * 16 * C::H * terms + (second ? 2.0 * 2.0 * C::H * C::H * np : 0.0) + 2.0 * C::H * D * np);
    ProfScope prof(D == 128 ? "lfa32_kernel<128>" : D == 256 ? "lfa32_kernel<256>" : "lfa32_kernel<512>", s, rows * (2.0 * 10 * C::H + (second ? 2.0 * C::H * C::H : 0.0) + 2.0 * D * D + 2.0 * D), exec);
    if (terms == 2) {
        if (second) hipLaunchKernelGGL((lfa32_kernel<D, true, 2>), grid, dim3(nw * 64), 0, s, a);
        else hipLaunchKernelGGL((lfa32_kernel<D, false, 2>), grid, dim3(nw * 64), 0, s, a);
    } else {
        if (second) hipLaunchKernelGGL((lfa32_kernel<D, true, 1>), grid, dim3(nw * 64), 0, s, a);
        else hipLaunchKernelGGL((lfa32_kernel<D, false, 1>), grid, dim3(nw * 64), 0, s, a);
    }
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

template <int D> static int launch_lfa32_res(const Lfa32Args& a, bool second, int B, int prec, hipStream_t s) {
    using C = Lfa32ResCfg<D>;
    dim3 grid((unsigned)((a.n + 2 * C::NW * C::RPW - 1) / (2 * C::NW * C::RPW)), (unsigned)B);
    const int terms = prec == PREC_BF16X3 ? 2 : 1;
    const size_t lds = C::lds_bytes(terms, second);
    static std::once_flag attr_once;
    hipError_t ae = hipSuccess;
    std::call_once(attr_once, [&] {
        ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&lfa32_res_kernel<D, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes(2, true));
        if (ae == hipSuccess) ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&lfa32_res_kernel<D, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes(2, false));
        if (ae == hipSuccess) ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&lfa32_res_kernel<D, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes(1, true));
        if (ae == hipSuccess) ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&lfa32_res_kernel<D, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes(1, false));
    });
    SSDR_HIP(ae);
    const double rows = (double)B * (double)a.n * 16.0, np = terms == 2 ? 3.0 : 1.0;
    // executed: LocSE (K padded to 16, both orientations, two instructions for the four products), LFAmlp2 in both orientations, the whole d x d attention product
    const double exec = rows * (2.0 * 2.0 * 16 * C::H * terms + (second ? 2.0 * 2.0 * C::H * C::H * np : 0.0) + 2.0 * D * D * np);
    ProfScope prof("lfa32_res_kernel<64>", s, rows * (2.0 * 10 * C::H + (second ? 2.0 * C::H * C::H : 0.0) + 2.0 * D * D + 2.0 * D), exec);
    const dim3 blk(C::NW * 64);
    if (terms == 2) {
        if (second) hipLaunchKernelGGL((lfa32_res_kernel<D, true, 2>), grid, blk, lds, s, a);
        else hipLaunchKernelGGL((lfa32_res_kernel<D, false, 2>), grid, blk, lds, s, a);
    } else {
        if (second) hipLaunchKernelGGL((lfa32_res_kernel<D, true, 1>), grid, blk, lds, s, a);
        else hipLaunchKernelGGL((lfa32_res_kernel<D, false, 1>), grid, blk, lds, s, a);
    }
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int launch_lfa32(int D, const Lfa32Args& a, bool second, int B, int prec, hipStream_t s) {
    if (a.n <= 0 || B <= 0) return SSDR_OK;
    if ((!a.g && D > 64) || !a.fc_hi || !a.w1p || (second && !a.w2_hi)) { set_error("lfa32: missing G rows or permuted weight pieces"); return SSDR_ERR_INVALID; }
    switch (D) {
        case 16: return launch_lfa32_l0(a, second, B, prec, s);
        case 64: return launch_lfa32_res<64>(a, second, B, prec, s);
        case 128: return launch_lfa32_d<128>(a, second, B, prec, s);      // (the no-G form was measured at this width: 103 / 121 us against 67 / 100 — twice the matrix work, one workgroup per CU)
        case 256: return launch_lfa32_d<256>(a, second, B, prec, s);
        case 512: return launch_lfa32_d<512>(a, second, B, prec, s);
        default: return SSDR_ERR_UNSUPPORTED;       // no error text: the caller falls back to lfa_bf16_kernel
    }
}

}  // namespace ssdr
