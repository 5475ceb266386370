// Directed chamfer means between the candidate superpoints of a cloud (F1: fps_gcn_cpu.py:12-38 `chamfer_distance`, called pair by pair from
// :84-100) for gfx950:  dir[i*n + j] = mean over the points a of superpoint i of  min over the points b of superpoint j of |(a - c_i) - (b - c_j)|,
// float64, every superpoint centred on its own bounding box (:33).
//
// Where the time goes is the minimum over b for every a: 2.2 G point pairs per bench step.  Rounds 2-5 evaluated every pair on the float64 vector
// pipe (7, then 5 instructions per pair: 0.70, then 0.50 ms).  This file SCREENS the pairs on the matrix cores and evaluates in float64 only what
// the screening cannot decide:
//
//   * key(a, b) = |b|^2 - 2 a.b  (= |a - b|^2 - |a|^2: the same order over b).  Every coordinate is scaled by 128 and cut into two half-precision
//     pieces (hi + lo = 22 bits), |b|^2 likewise; the products hi*hi + lo*hi + hi*lo of the three dimensions and the two pieces of |b|^2 are 11 of
//     the 16 k-slots of ONE `v_mfma_f32_32x32x16_f16` — 32 target points x 32 source points per instruction, a lane holding 16 targets (four
//     runs of four consecutive ones) of one source, float32 accumulation.  The half-precision matrix instruction runs beside the vector pipe;
//     the float32-input one (first version of this file, 2 x `v_mfma_f32_32x32x2_f32` per tile) turned out to SHARE it, like the float64 one
//     (tools/micro/valu_rate.hip: one of them + 16 v_max3_f32 take 64 + 75 cycles, the half-precision one + the same 16: 82) — measured 0.50 ms, no gain.
//   * per run of four the vector pipe takes the minimum (v_min3 + v_min), and per source keeps the smallest run minimum m1, the run it came from
//     and the second smallest run minimum m2 (v_med3, compare, select, v_min): 6 instructions per 4 x 64 pairs where the float64 form needed 20.
//   * the screened key is off the true one by at most eps (bound below); m2 - m1 > 2 eps proves that the nearest target lies in the winning run:
//     its four points are then evaluated in float64 in the difference form exactly as before and their minimum IS the minimum over the whole
//     target.  Otherwise (near-ties between runs, duplicated points: ~3 per thousand sources) the source is swept over the whole target in
//     float64 by the wave, one target point per lane.
//
// eps, in the scaled units (A = 128 a, B = 128 b, key' = 2^14 key; Ra', Rb' the largest |A|, |B|): pieces: x = hi + lo + d, |d| <= 1.25 * 2^-22 |x|
// (float64 -> float32 -> two roundings to half precision) + 2^-25 where a piece is subnormal; dropped lo*lo <= 2^-22 |A||B|; so the eleven products
// miss 2 A.B - |B|^2 by <= 2^-20.1 (Ra'^2 + Rb'^2) + 2^-22 (Ra' + Rb') + 2^-13; the matrix instruction's float32 accumulation of eleven non-zero
// terms, in whatever order, each addition off by at most 2^-23 of a partial sum <= sum |terms| <= 1.01 (Ra'^2 + 2 Rb'^2): <= 2^-18.4 (Ra'^2 + Rb'^2).
// Together < 2^-17.9 (Ra'^2 + Rb'^2) + ...; the kernel uses eps = 2^-17 (Ra'^2 + Rb'^2) + 2^-22 (Ra' + Rb') + 2^-12.  Superpoints beyond 31 m
// from their box centre (half precision's range after the scaling) take the float64 path.
//
// So the value written is the float64 difference-form distance to the nearest target point in every case.  The float64 kernel (kept:
// SSDR_CHAMFER_F64=1, and for targets too large to stage) writes that distance for the target its float64 key names: the same bits wherever the
// nearest target is unique, and within the last ulp where two targets are equally near (a lattice); `tests/test_select.py` compares the two on
// random, duplicated and lattice clouds.
#include <atomic>
#include "select_chamfer.hpp"
#include "block_prims.hpp"
#include <cfloat>
#include <cstdlib>

namespace ssdr {
namespace {

#ifndef HIPEMU
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); }
__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) { return wave_reduce_u32(v, [](unsigned a, unsigned b) { return a < b ? a : b; }); }
#else
typedef hipemu_u32x4 u32x4;
typedef hipemu_f32x16 f32x16;
static inline f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) { return hipemu_mfma_f32_32x32x16_f16(a, b, c); }
static inline float med3(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
static inline unsigned wave_min_u32(unsigned v) { for (int o = 32; o > 0; o >>= 1) { const unsigned w = __shfl_xor(v, o); v = w < v ? w : v; } return v; }
#endif
// a (a value of the tile the LOWER half of the wave will finish) and b (of the tile the UPPER half will finish), each held by both lanes of a source with
// the lane half's own partial result: afterwards a = the lower half's partial result, b = the upper half's, of the tile this lane finishes — ONE
// v_permlane32_swap (a's upper 32 lanes <-> b's lower 32 lanes), no copies
__device__ __forceinline__ void swap_halves(unsigned& a, unsigned& b, int h) {
#ifndef HIPEMU
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; (void)h;
#else
    const unsigned ta = __shfl_xor(a, 32), tb_ = __shfl_xor(b, 32); a = h ? tb_ : a; b = h ? b : ta;
#endif
}
__device__ __forceinline__ void swap_halves_f32(float& a, float& b, int h) { unsigned x = __float_as_uint(a), y = __float_as_uint(b); swap_halves(x, y, h); a = __uint_as_float(x); b = __uint_as_float(y); }
constexpr int TS_MF_ = 3;      // doubles per staged target point of the screening kernel (TS_MF below)
// the staged float64 point b (a 32-bit multiply is a quarter-rate instruction; byte offsets are far below 2^24)
__device__ __forceinline__ const double* target_at(const double* tb, int b) {
#ifndef HIPEMU
    unsigned o;                 // (__umul24 is folded back into the 32-bit multiply)
    asm("v_mul_u32_u24_e32 %0, 24, %1" : "=v"(o) : "v"(b));
    static_assert(8 * TS_MF_ == 24, "byte stride of a staged target point");
    return reinterpret_cast<const double*>(reinterpret_cast<const char*>(tb) + o);
#else
    return tb + TS_MF_ * b;
#endif
}

// min of two non-NaN doubles in ONE instruction (`a < b ? a : b` compiles to a compare and two 32-bit selects)
__device__ __forceinline__ double min_f64(double a, double b) {
#ifndef HIPEMU
    double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#else
    return a < b ? a : b;
#endif
}
__device__ __forceinline__ double xor_f64(double v, int o) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = __shfl_xor((unsigned)b, o), hi = __shfl_xor((unsigned)(b >> 32), o);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += xor_f64(v, o);
    return v;
}
// minimum of non-negative doubles over the wave (they order like their bit patterns): two 32-bit reductions on the vector pipe
__device__ __forceinline__ double wave_min_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned hi = wave_min_u32((unsigned)(b >> 32));
    const unsigned lo = wave_min_u32((unsigned)(b >> 32) == hi ? (unsigned)b : 0xffffffffu);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

constexpr int TS_F64 = 4, TS_MF = 3;      // doubles per staged target point: x, y, z (+ |b|^2 for the float64 screening)

// ---- float64 screening (rounds 2-5; now for SSDR_CHAMFER_F64=1 and for targets beyond CH_TILE points) -----------------------------------------------
// squared distances from NV centred source points to the nearest point of target j: its staged points (tb), or streamed
__device__ __forceinline__ void chamfer_min(const double (&ax)[NV], const double (&ay)[NV], const double (&az)[NV], double (&out)[NV], const double* tb, int nj, bool staged,
                                            const float* __restrict__ xyz, const int* __restrict__ sp_pts, int loj, double cjx, double cjy, double cjz) {
    if (staged) {
        // Screening on |b|^2 - 2 a.b, three fused multiply-adds per pair instead of three differences, a product and two fused multiply-adds; the
        // target's index rides in the key's ten lowest mantissa bits (one 32-bit and-or), so the running minimum names its target and the distance
        // itself is then evaluated ONCE per source point in the difference form — the value is the exact one whenever the target it names is the
        // nearest.  A wrong name needs two targets whose keys agree to 2^-42 of |key| (<= ~1e-13 m^2 at room scale): the value returned is then
        // that target's exact distance, above the minimum by at most that band.  5 instructions per pair on the float64 pipe.
        double m[NV][2];      // independent chains: min is order-free
        double ax2[NV], ay2[NV], az2[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) { m[v][0] = m[v][1] = 1.0e300; ax2[v] = -2.0 * ax[v]; ay2[v] = -2.0 * ay[v]; az2[v] = -2.0 * az[v]; }
        // (key.lo & ~1023) | idx in ONE instruction: gfx950's three-operand encodings take no 32-bit literal and one scalar operand, so the compiler
        // emits v_and_b32 + v_or_b32 with the literal; with the mask in a vector register v_and_or_b32 does it (idx stays scalar)
        const unsigned keep = ~1023u;
        auto key = [&](double t, int idx) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(t);
            unsigned lo;
#ifndef HIPEMU
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(lo) : "v"((unsigned)bits), "v"(keep), "s"(idx));
#else
            lo = ((unsigned)bits & keep) | (unsigned)idx;
#endif
            return __longlong_as_double((long long)((bits & 0xffffffff00000000ull) | lo));
        };
        int b = 0;
        for (; b + 4 <= nj; b += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double tx = tb[TS_F64 * (b + u)], ty = tb[TS_F64 * (b + u) + 1], tz = tb[TS_F64 * (b + u) + 2], tn = tb[TS_F64 * (b + u) + 3];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const double t = fma(az2[v], tz, fma(ay2[v], ty, fma(ax2[v], tx, tn)));
                    m[v][u & 1] = min_f64(key(t, b + u), m[v][u & 1]);
                }
            }
        }
        for (; b < nj; ++b) {
            const double tx = tb[TS_F64 * b], ty = tb[TS_F64 * b + 1], tz = tb[TS_F64 * b + 2], tn = tb[TS_F64 * b + 3];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double t = fma(az2[v], tz, fma(ay2[v], ty, fma(ax2[v], tx, tn)));
                m[v][0] = min_f64(key(t, b), m[v][0]);
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = (int)((unsigned)__double_as_longlong(min_f64(m[v][0], m[v][1])) & 1023u);
            const double dx = ax[v] - tb[TS_F64 * idx], dy = ay[v] - tb[TS_F64 * idx + 1], dz = az[v] - tb[TS_F64 * idx + 2];
            // (an empty target leaves the 1e300 sentinel, whose low bits name no staged point: the streamed form's and the screening's answer, sqrt -> 1e150)
            out[v] = nj > 0 ? fma(dz, dz, fma(dy, dy, dx * dx)) : 1.0e300;      // fused: the chamfer terms are compared at 1e-12, not bit for bit
        }
        return;
    }
    double m[NV];                        // very large target: stream it from global memory
#pragma unroll
    for (int v = 0; v < NV; ++v) m[v] = 1.0e300;
    for (int b = 0; b < nj; ++b) {
        const size_t q = sp_pts[loj + b];
        const double tx = (double)xyz[3 * q] - cjx, ty = (double)xyz[3 * q + 1] - cjy, tz = (double)xyz[3 * q + 2] - cjz;
#pragma unroll
        for (int v = 0; v < NV; ++v) { const double dx = ax[v] - tx, dy = ay[v] - ty, dz = az[v] - tz; double d = dx * dx; d = d + dy * dy; d = d + dz * dz; m[v] = min_f64(m[v], d); }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) out[v] = m[v];
}

// ---- screening on the matrix cores -------------------------------------------------------------------------------------------------------------------
// The wave's 256 source slots against the staged target: sv[slot] = the distance (root taken) from slot's point to its nearest target point.
//   oper(slot, h): the slot's point as the source operand of lane half h (source_operand; the packer keeps it per slot);
//   comp(slot, d): coordinate d of the slot's centred point (float64); live(slot): the slot holds a point whose distance is used;
//   ta0 / ta1: the target as MFMA operand tiles (stage_target below), padded to whole tiles of 32 with keys above every real one;
//   tb: its float64 points, TS_MF doubles each;  thr = 2 eps of (this item, this target) in the scaled units.
// Lane (c = lane & 31, h = lane >> 5), source tile t: slot 32 t + c.  k-slots of the instruction (lane half h holds k = 8 h .. 8 h + 7):
//     k      0        1        2        3        4        5        6       7       8        9        10      11..15
//   source  Xh       Xl       Xh       Yh       Yl       Yh       2^12    2^12    Zh       Zl       Zh       0          (X = -256 ax, ...)
//   target  Xh       Xh       Xl       Yh       Yh       Yl       Wh      Wl      Zh       Zh       Zl       0          (X = 128 bx, ..., W = 4 |b|^2)
// The lane's sixteen results are rows (reg & 3) + 8 (reg >> 2) + 4 h of column c: run q = reg >> 2 holds targets 32 T + 8 q + 4 h .. + 3.
#ifndef MF_RUN
#define MF_RUN 4          // targets per run: 4 (the four consecutive rows of a result register quad) or 8 (two quads, rows r .. r + 3 and r + 8 .. r + 11)
#endif
struct TargetTiles { const u32x4* t0; const unsigned long long* t1; };      // [tile][32] operands of the lower / the upper lane half
template <class Oper, class Comp, class Live>
__device__ __forceinline__ void chamfer_min_mf(Oper oper, Comp comp, Live live, TargetTiles ta, const double* __restrict__ tb, int nj, float thr, double* sv, int lane) {
    const int c = lane & 31, h = lane >> 5;
    u32x4 B[8]; float m1[8], m2[8]; int id[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const uint4 o = oper(32 * t + c, h);
        B[t][0] = o.x; B[t][1] = o.y; B[t][2] = o.z; B[t][3] = o.w;
        m1[t] = m2[t] = FLT_MAX; id[t] = 0;
    }
    const int ntiles = (nj + 31) >> 5;
    for (int T = 0; T < ntiles; ++T) {
        u32x4 A = {0u, 0u, 0u, 0u};
        if (h) { const unsigned long long v = ta.t1[T * 32 + c]; A[0] = (unsigned)v; A[1] = (unsigned)(v >> 32); }
        else A = ta.t0[T * 32 + c];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x16 acc = {};
            acc = mfma_f16(A, B[t], acc);
#pragma unroll
            for (int q = 0; q < 16 / MF_RUN; ++q) {
                float cq = acc[MF_RUN * q];
#pragma unroll
                for (int e = 1; e < MF_RUN; ++e) cq = fminf(cq, acc[MF_RUN * q + e]);
                m2[t] = med3(cq, m1[t], m2[t]);                 // m1 <= m2 before and after
                id[t] = cq < m1[t] ? (16 / MF_RUN) * T + q : id[t];
                m1[t] = fminf(m1[t], cq);
            }
        }
    }
    // The two halves of the wave saw different runs of the same source.  The lower half finishes source tiles 0-3, the upper half tiles 4-7: one swap per
    // quantity and pair of tiles (u, 4 + u) hands every lane both halves' partial results of ITS tile.
    int first[4]; unsigned unsure = 0;          // bit u: this lane's source of tile 4 h + u is not decided by the screening
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float lo1 = m1[u], hi1 = m1[4 + u], lo2 = m2[u], hi2 = m2[4 + u]; unsigned lid = (unsigned)id[u], hid = (unsigned)id[4 + u];
        swap_halves_f32(lo1, hi1, h); swap_halves_f32(lo2, hi2, h); swap_halves(lid, hid, h);
        const bool up = hi1 < lo1;
        first[u] = 2 * MF_RUN * (int)(up ? hid : lid) + (up ? 4 : 0);
        const float n1 = fminf(lo1, hi1), n2 = fminf(fmaxf(lo1, hi1), fminf(lo2, hi2));
        unsure |= (n2 - n1 > thr) ? 0u : 1u << u;
    }
    // ... the winning run in float64: four distances per source
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int slot = 32 * (4 * h + u) + c, f = first[u];
        const double ax = comp(slot, 0), ay = comp(slot, 1), az = comp(slot, 2);
        double best = 1.0e300;
#pragma unroll
        for (int e = 0; e < MF_RUN; ++e) {
            const int b = max(min(f + (e & 3) + 8 * (e >> 2), nj - 1), 0);      // (a run that reaches into the padding: the last point again; an empty target: undecided, swept below)
            const double* tp = target_at(tb, b);
            const double dx = ax - tp[0], dy = ay - tp[1], dz = az - tp[2];
            best = min_f64(best, fma(dz, dz, fma(dy, dy, dx * dx)));
        }
        sv[slot] = sqrt(best);
    }
    // the undecided ones: the whole target in float64, a target point per lane
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        unsigned long long todo = __ballot(((unsure >> u) & 1u) && live(32 * (4 * h + u) + c));
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int s = 32 * (4 * (l >> 5) + u) + (l & 31);
            const double sx = comp(s, 0), sy = comp(s, 1), sz = comp(s, 2);
            double bm = 1.0e300;
            for (int b = lane; b < nj; b += 64) {
                const double* tp = target_at(tb, b);
                const double dx = sx - tp[0], dy = sy - tp[1], dz = sz - tp[2];
                bm = min_f64(bm, fma(dz, dz, fma(dy, dy, dx * dx)));
            }
            bm = wave_min_f64(bm);
            if (lane == 0) sv[s] = sqrt(bm);
        }
    }
}

// v[0] + ... + v[n-1], n <= SEQ_MAX, by one lane
__device__ __forceinline__ double sum_short(const double* v, int n) {
    double a = 0.0;
    for (int t = 0; t < n; ++t) a += v[t];
    return a;
}
// v[0] + ... + v[n-1] by the whole wave (uniform arguments): lane l adds v[l], v[l + 64], ... up, then the xor tree
__device__ __forceinline__ double sum_wave(const double* v, int n, int lane) {
    double a = 0.0;
    for (int t = lane; t < n; t += 64) a += v[t];
    return wave_sum_f64(a);
}

// One workgroup per target superpoint j (and slice of the sources): its centred points are staged in LDS once and re-used against
// every source item.  The mean is (sum_short | sum_wave) / size, so it can differ from NumPy's pairwise np.mean in the last ulps
// (sqrt is monotone: min of roots == root of min).
template <bool MF>
__device__ void chamfer_dir_body(const float* __restrict__ xyz, const int* __restrict__ sp_off, const int* __restrict__ sp_pts,
                                 const int* __restrict__ sel, int nsel, const double* __restrict__ centres, double* dir, ChamferPack P, const int* counts,
                                 double* tb, u32x4* ta0, unsigned long long* ta1, double (*s_val)[ITEM]) {
    constexpr int TS = MF ? TS_MF : TS_F64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nitems = counts[0], nbig = counts[1];
    for (int j = blockIdx.x; j < nsel; j += gridDim.x) {
        const int sj = sel[j], loj = sp_off[sj], nj = sp_off[sj + 1] - loj;
        const double cjx = centres[3 * j], cjy = centres[3 * j + 1], cjz = centres[3 * j + 2];
        const float r2j = MF ? P.r2sp[j] : 0.0f;
        const bool staged = nj <= CH_TILE && (!MF || r2j <= MF_R2_MAX);
        if (staged) {
            __syncthreads();
            const int st = P.start[j];
            const int padded = MF ? ((nj + 31) & ~31) : nj;
            for (int b = threadIdx.x; b < padded; b += 256) {
                double x = 0.0, y = 0.0, z = 0.0;
                if (b < nj) {
                    if (st >= 0) { x = P.x[st + b]; y = P.y[st + b]; z = P.z[st + b]; }
                    else { const size_t q = sp_pts[loj + b]; x = (double)xyz[3 * q] - cjx; y = (double)xyz[3 * q + 1] - cjy; z = (double)xyz[3 * q + 2] - cjz; }
                    const double w = fma(z, z, fma(y, y, x * x));
                    tb[TS * b] = x; tb[TS * b + 1] = y; tb[TS * b + 2] = z;
                    if constexpr (!MF) tb[TS * b + 3] = w;
                    if constexpr (MF) {             // the operand images of chamfer_min_mf's k-slot table
                        unsigned xh, xl, yh, yl, zh, zl, wh, wl;
                        split16((float)(x * (double)MF_SCALE), xh, xl); split16((float)(y * (double)MF_SCALE), yh, yl); split16((float)(z * (double)MF_SCALE), zh, zl);
                        split16((float)(w * 4.0), wh, wl);
                        u32x4 o; o[0] = xh | (xh << 16); o[1] = xl | (yh << 16); o[2] = yh | (yl << 16); o[3] = wh | (wl << 16);
                        ta0[b] = o; ta1[b] = (unsigned long long)(zh | (zh << 16)) | ((unsigned long long)zl << 32);
                    }
                } else if constexpr (MF) { u32x4 o = {0u, 0u, 0u, 0x7800u}; ta0[b] = o; ta1[b] = 0ull; }      // W = 32768: a key of 2^27, above every real one (<= 3 * 2^14 * 1000)
            }
            __syncthreads();
        }
        const TargetTiles ta{ta0, ta1};
        // the item's means from the roots in s_val[wid] (slot order): a superpoint's sum depends on its size alone
        auto emit_item = [&](int k) {
            (void)__ballot(1);                               // the wave's LDS writes are visible to its lanes
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int cnt = P.cnt[k + 64 * v], seg = P.seg[k + 64 * v];
                if (cnt > 0 && cnt <= SEQ_MAX) dir[(size_t)seg * nsel + j] = seg == j ? 0.0 : sum_short(&s_val[wid][lane + 64 * v], cnt) / (double)cnt;
                unsigned long long todo = __ballot(cnt > SEQ_MAX);          // the larger ones, one after the other, all lanes on each
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const int c = __shfl(cnt, src), sg = __shfl(seg, src);
                    const double sum = sum_wave(&s_val[wid][src + 64 * v], c, lane);
                    if (lane == 0) dir[(size_t)sg * nsel + j] = sg == j ? 0.0 : sum / (double)c;
                }
            }
            (void)__ballot(1);
        };
        if (nj > CH_TILE) continue;          // a large target: sel_chamfer_big (its own kernel: inlined here, that path's registers made this one spill)
        // 2 eps in the scaled units (file header), from the largest |a|^2 of the item and |b|^2 of the target
        auto thr_of = [&](float r2a) { return 2.0f * (0x1p-17f * (MF_SCALE * MF_SCALE) * (r2a + r2j) + 0x1p-22f * MF_SCALE * (sqrtf(r2a) + sqrtf(r2j)) + 0x1p-12f); };
        for (int it = blockIdx.y * 4 + wid; it < nitems; it += 4 * gridDim.y) {          // whole superpoints per wave
            const int k0 = P.item_slot[it], k = k0 + lane;
            const float r2i = MF ? P.r2item[k0 / ITEM] : 0.0f;
            if (MF && staged && r2i <= MF_R2_MAX) {
                const float thr = thr_of(r2i);
                chamfer_min_mf([&](int slot, int h) { if (!h) return P.src0[k0 + slot]; const unsigned long long v = P.src1[k0 + slot]; return uint4{(unsigned)v, (unsigned)(v >> 32), 0u, 0u}; },
                               [&](int slot, int d) { return (d == 0 ? P.x : d == 1 ? P.y : P.z)[k0 + slot]; }, [&](int slot) { return P.seg[k0 + slot] >= 0; },
                               ta, tb, nj, thr, s_val[wid], lane);
            } else {
                double ax[NV], ay[NV], az[NV], m[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) { ax[v] = P.x[k + 64 * v]; ay[v] = P.y[k + 64 * v]; az[v] = P.z[k + 64 * v]; }
                chamfer_min(ax, ay, az, m, tb, nj, staged && !MF, xyz, sp_pts, loj, cjx, cjy, cjz);      // (with MF: out of the screening's range)
#pragma unroll
                for (int v = 0; v < NV; ++v) s_val[wid][lane + 64 * v] = sqrt(m[v]);
            }
            emit_item(k);
        }
        for (int bi = blockIdx.y * 4 + wid; bi < nbig; bi += 4 * gridDim.y) {            // pair by pair: more than ITEM points (or none)
            const int i = P.big[bi];
            if (i == j) { if (lane == 0) dir[(size_t)i * nsel + j] = 0.0; continue; }
            const int si = sel[i], loi = sp_off[si], ni = sp_off[si + 1] - loi;
            const double cix = centres[3 * i], ciy = centres[3 * i + 1], ciz = centres[3 * i + 2];
            const float r2i = MF ? P.r2sp[i] : 0.0f, thr = MF ? thr_of(r2i) : 0.0f;
            double acc = 0.0;
            for (int a0 = 0; a0 < ni; a0 += ITEM) {
                if (MF && staged && r2i <= MF_R2_MAX) {
                    auto at = [&](int slot, int d) {                // beyond the end: the last point again, not summed
                        const size_t p = sp_pts[loi + min(a0 + slot, ni - 1)];
                        return (double)xyz[3 * p + d] - (d == 0 ? cix : d == 1 ? ciy : ciz);
                    };
                    chamfer_min_mf([&](int slot, int h) { return source_operand(at(slot, h ? 2 : 0), at(slot, 1), h); }, at,
                                   [&](int slot) { return a0 + slot < ni; }, ta, tb, nj, thr, s_val[wid], lane);
                } else {
                    double ax[NV], ay[NV], az[NV], m[NV];
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const int a = min(a0 + lane + 64 * v, ni - 1);          // beyond the end: the last point again, not summed
                        const size_t p = sp_pts[loi + a];
                        ax[v] = (double)xyz[3 * p] - cix; ay[v] = (double)xyz[3 * p + 1] - ciy; az[v] = (double)xyz[3 * p + 2] - ciz;
                    }
                    chamfer_min(ax, ay, az, m, tb, nj, staged && !MF, xyz, sp_pts, loj, cjx, cjy, cjz);
#pragma unroll
                    for (int v = 0; v < NV; ++v) s_val[wid][lane + 64 * v] = sqrt(m[v]);
                }
                (void)__ballot(1);
                if (ni <= SEQ_MAX) { if (lane == 0) acc = sum_short(&s_val[wid][0], ni); }      // same rule as inside an item
                else acc += sum_wave(&s_val[wid][0], min(ITEM, ni - a0), lane);
                (void)__ballot(1);
            }
            if (lane == 0) dir[(size_t)i * nsel + j] = ni > 0 ? acc / (double)ni : 0.0;
        }
    }
}


// ---- targets beyond the staging limit ---------------------------------------------------------------------------------------------------------------
// A LARGE target (a floor or a wall of a real partition: thousands of points; the synthetic stand-in's regions never get here).  Until round 6 every wave
// streamed such a target from global memory for every item (~30 x the staged cost per pair: 14 ms of a step with 117 such regions, tools/sp_probe.py).
// Here the workgroup walks its items in lockstep — four at a time, one per wave — and for each group stages the target CHUNK points at a time; a wave
// keeps its sources' running minima in registers across the chunks (float64 screening per chunk: the exact distance to the chunk's nearest point, so the
// minimum over the chunks is the exact distance to the target).  A kernel of its own behind the main one (a workgroup without a large target leaves at
// once): inlined into chamfer_dir_body the path's registers made the COMMON path spill (scratch 16 -> 52 bytes per lane, 320 -> 335 us); as a noinline
// call the frame cost more still (0.31 -> 0.47 ms).  Its grid takes more slices of the items than the main kernel's: the few large targets carry the work.
__device__ void chamfer_big_targets(const float* __restrict__ xyz, const int* __restrict__ sp_off, const int* __restrict__ sp_pts,
                                    const int* __restrict__ sel, int nsel, const double* __restrict__ centres, double* dir, ChamferPack P, const int* counts,
                                    double* tb, double (*s_val)[ITEM], int* s_ni, int* s_aux, double* s_r) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nitems = counts[0], nbig = counts[1];
    constexpr int CHUNK = CH_TILE;
    for (int bj = blockIdx.x; bj < nbig; bj += gridDim.x) {          // the large targets are among the packer's pair-by-pair superpoints (above ITEM points): usually none
        const int j = P.big[bj];
        const int sj = sel[j], loj = sp_off[sj], nj = sp_off[sj + 1] - loj;
        if (nj <= CH_TILE) continue;
        const double cjx = centres[3 * j], cjy = centres[3 * j + 1], cjz = centres[3 * j + 2];
        auto emit_item = [&](int k) {
            (void)__ballot(1);                               // the wave's LDS writes are visible to its lanes
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int cnt = P.cnt[k + 64 * v], seg = P.seg[k + 64 * v];
                if (cnt > 0 && cnt <= SEQ_MAX) dir[(size_t)seg * nsel + j] = seg == j ? 0.0 : sum_short(&s_val[wid][lane + 64 * v], cnt) / (double)cnt;
                unsigned long long todo = __ballot(cnt > SEQ_MAX);
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const int c = __shfl(cnt, src), sg = __shfl(seg, src);
                    const double sum = sum_wave(&s_val[wid][src + 64 * v], c, lane);
                    if (lane == 0) dir[(size_t)sg * nsel + j] = sg == j ? 0.0 : sum / (double)c;
                }
            }
            (void)__ballot(1);
        };
        // Both superpoints are centred on their own boxes, so the sources sit around the ORIGIN of the target's frame (a small region within ~0.3 m of it): the
        // target's points beyond R1 = (the workgroup's largest |a|) + 0.25 m cannot be nearest to a source whose best distance so far is below R1 - |a|.  Two passes
        // over the target: the points inside R1 (a quarter of a 1.5 m slab), then — only if some wave's sources are not settled by that bound — the ones outside.
        // A block's points of the pass are compacted into the staging buffer in their own order (ballots: the same layout on every run).
        auto wave_max = [&](double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, xor_f64(v, o)); return v; };
        auto stage_part = [&](int c0, int cn, double r1sq, bool inner) -> int {
            __syncthreads();
            int base = 0;
            for (int b0 = 0; b0 < cn; b0 += 256) {
                const int b = b0 + (int)threadIdx.x;
                double x = 0.0, y = 0.0, z = 0.0, w = 0.0; bool take = false;
                if (b < cn) {
                    const size_t q = sp_pts[loj + c0 + b];
                    x = (double)xyz[3 * q] - cjx; y = (double)xyz[3 * q + 1] - cjy; z = (double)xyz[3 * q + 2] - cjz;
                    w = fma(z, z, fma(y, y, x * x));
                    take = (w <= r1sq) == inner;
                }
                const unsigned long long bal = __ballot(take);
                if (lane == 0) s_aux[wid] = __popcll(bal);
                __syncthreads();
                int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                for (int w2 = 0; w2 < wid; ++w2) pos += s_aux[w2];
                if (take) { tb[TS_F64 * pos] = x; tb[TS_F64 * pos + 1] = y; tb[TS_F64 * pos + 2] = z; tb[TS_F64 * pos + 3] = w; }
                base += s_aux[0] + s_aux[1] + s_aux[2] + s_aux[3];
                __syncthreads();
            }
            return base;
        };
        auto min_over_target = [&](bool on, const double (&ax)[NV], const double (&ay)[NV], const double (&az)[NV], double (&m)[NV]) {      // (uniform over the workgroup)
            double ra2 = 0.0;
#pragma unroll
            for (int v = 0; v < NV; ++v) { m[v] = 1.0e300; if (on) ra2 = fmax(ra2, fma(az[v], az[v], fma(ay[v], ay[v], ax[v] * ax[v]))); }
            ra2 = wave_max(ra2);
            __syncthreads();
            if (lane == 0) s_r[wid] = ra2;
            __syncthreads();
            const double ra = sqrt(ra2), r1 = sqrt(fmax(fmax(s_r[0], s_r[1]), fmax(s_r[2], s_r[3]))) + 0.25, r1sq = r1 * r1;
            bool need = on;
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) {          // settled: every point outside R1 is at least R1 - |a| from each of the wave's sources
                    double mx = 0.0;
#pragma unroll
                    for (int v = 0; v < NV; ++v) mx = fmax(mx, m[v]);
                    mx = wave_max(on ? mx : 0.0);
                    need = on && !(sqrt(mx) * (1.0 + 1.0e-9) + 1.0e-12 <= r1 - ra);
                    __syncthreads();
                    if (threadIdx.x == 0) s_aux[4] = 0;
                    __syncthreads();
                    if (need && lane == 0) s_aux[4] = 1;
                    __syncthreads();
                    if (!s_aux[4]) break;
                }
                for (int c0 = 0; c0 < nj; c0 += CHUNK) {
                    const int cn = stage_part(c0, min(CHUNK, nj - c0), r1sq, pass == 0);
                    if (need && cn > 0) {
                        double mc[NV];
                        chamfer_min(ax, ay, az, mc, tb, cn, true, xyz, sp_pts, loj, cjx, cjy, cjz);
#pragma unroll
                        for (int v = 0; v < NV; ++v) m[v] = min_f64(m[v], mc[v]);
                    }
                }
            }
        };
        for (int itb = blockIdx.y * 4; itb < nitems; itb += 4 * gridDim.y) {
            const int it = itb + wid; const bool on = it < nitems;
            const int k0 = on ? P.item_slot[it] : 0, k = k0 + lane;
            double ax[NV], ay[NV], az[NV], m[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) { ax[v] = on ? P.x[k + 64 * v] : 0.0; ay[v] = on ? P.y[k + 64 * v] : 0.0; az[v] = on ? P.z[k + 64 * v] : 0.0; }
            min_over_target(on, ax, ay, az, m);
            if (on) {
#pragma unroll
                for (int v = 0; v < NV; ++v) s_val[wid][lane + 64 * v] = sqrt(m[v]);
                emit_item(k);
            }
        }
        for (int bib = blockIdx.y * 4; bib < nbig; bib += 4 * gridDim.y) {           // the large SOURCES, a wave each, their passes of ITEM points in lockstep
            const int bi = bib + wid; const bool act = bi < nbig;
            const int i = act ? P.big[bi] : j;
            const int si = sel[i], loi = sp_off[si], ni = (act && i != j) ? sp_off[si + 1] - loi : 0;
            const double cix = centres[3 * i], ciy = centres[3 * i + 1], ciz = centres[3 * i + 2];
            __syncthreads();
            if (lane == 0) s_ni[wid] = ni;
            __syncthreads();
            const int ni_max = max(max(s_ni[0], s_ni[1]), max(s_ni[2], s_ni[3]));
            double acc = 0.0;
            for (int a0 = 0; a0 < ni_max; a0 += ITEM) {
                const bool on = a0 < ni;
                double ax[NV], ay[NV], az[NV], m[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int a = max(min(a0 + lane + 64 * v, ni - 1), 0);     // beyond the end: the last point again, not summed
                    const size_t p = on ? sp_pts[loi + a] : 0;
                    ax[v] = on ? (double)xyz[3 * p] - cix : 0.0; ay[v] = on ? (double)xyz[3 * p + 1] - ciy : 0.0; az[v] = on ? (double)xyz[3 * p + 2] - ciz : 0.0;
                }
                min_over_target(on, ax, ay, az, m);
                if (on) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) s_val[wid][lane + 64 * v] = sqrt(m[v]);
                    (void)__ballot(1);
                    if (ni <= SEQ_MAX) { if (lane == 0) acc = sum_short(&s_val[wid][0], ni); }
                    else acc += sum_wave(&s_val[wid][0], min(ITEM, ni - a0), lane);
                    (void)__ballot(1);
                }
            }
            if (act && lane == 0) dir[(size_t)i * nsel + j] = (i != j && ni > 0) ? acc / (double)ni : 0.0;
        }
    }
}
__global__ __launch_bounds__(256) void sel_chamfer_big(const float* __restrict__ xyz, const int* __restrict__ sp_off, const int* __restrict__ sp_pts,
                                                      const int* __restrict__ sel, int nsel, const double* __restrict__ centres, double* dir, ChamferPack P) {
    __shared__ double tb[CH_TILE * TS_F64];
    __shared__ double s_val[4][ITEM];
    __shared__ int s_ni[4], s_aux[8];
    __shared__ double s_r[4];
    chamfer_big_targets(xyz, sp_off, sp_pts, sel, nsel, centres, dir, P, P.counts, tb, s_val, s_ni, s_aux, s_r);
}
__global__ __launch_bounds__(256) void sel_chamfer_big_batch(const float* __restrict__ xyz, const int* __restrict__ sp_off, const int* __restrict__ sp_pts,
                                                            const int* __restrict__ sel, const int* __restrict__ coff, const long long* __restrict__ boff,
                                                            const double* __restrict__ centres, double* dir, ChamferPack P) {
    __shared__ double tb[CH_TILE * TS_F64];
    __shared__ double s_val[4][ITEM];
    __shared__ int s_ni[4], s_aux[8];
    __shared__ double s_r[4];
    const int c = blockIdx.z, lo = coff[c], n = coff[c + 1] - lo;
    chamfer_big_targets(xyz, sp_off, sp_pts, sel + lo, n, centres + 3 * (size_t)lo, dir + boff[c], pack_at(P, lo), P.counts + 2 * c, tb, s_val, s_ni, s_aux, s_r);
}

template <bool MF>
__global__ __launch_bounds__(256) SSDR_WAVES_PER_EU(MF ? 4 : 5) void sel_chamfer_dir(const float* __restrict__ xyz, const int* __restrict__ sp_off, const int* __restrict__ sp_pts,
                                                       const int* __restrict__ sel, int nsel, const double* __restrict__ centres, double* dir, ChamferPack P) {
    __shared__ double tb[CH_TILE * (MF ? TS_MF : TS_F64)];
    __shared__ u32x4 ta0[MF ? CH_TILE : 1];
    __shared__ unsigned long long ta1[MF ? CH_TILE : 1];
    __shared__ double s_val[4][ITEM];
    chamfer_dir_body<MF>(xyz, sp_off, sp_pts, sel, nsel, centres, dir, P, P.counts, tb, ta0, ta1, s_val);
}
// all clouds of a batch in one launch: blockIdx.z = cloud
template <bool MF>
__global__ __launch_bounds__(256) SSDR_WAVES_PER_EU(MF ? 4 : 5) void sel_chamfer_dir_batch(const float* __restrict__ xyz, const int* __restrict__ sp_off, const int* __restrict__ sp_pts,
                                                             const int* __restrict__ sel, const int* __restrict__ coff, const long long* __restrict__ boff,
                                                             const double* __restrict__ centres, double* dir, ChamferPack P) {
    __shared__ double tb[CH_TILE * (MF ? TS_MF : TS_F64)];
    __shared__ u32x4 ta0[MF ? CH_TILE : 1];
    __shared__ unsigned long long ta1[MF ? CH_TILE : 1];
    __shared__ double s_val[4][ITEM];
    const int c = blockIdx.z, lo = coff[c], n = coff[c + 1] - lo;
    chamfer_dir_body<MF>(xyz, sp_off, sp_pts, sel + lo, n, centres + 3 * (size_t)lo, dir + boff[c], pack_at(P, lo), P.counts + 2 * c, tb, ta0, ta1, s_val);
}

// ---- the Semantic3D flavour: float32 CUDA-kernel chamfer values (SSRD_AL_semantic3d/fps_gcn_cuda.py:13-30) --------------------------------------------
// create_cd_cuda centres a superpoint in NumPy (float32 coordinates minus a float64 bounding-box centre -> float64), rounds the result to float32
// (torch.Tensor), runs chamfer3D.cu on the pair — squared float32 distances, (dx*dx + dy*dy) + dz*dz — and keeps mean(sqrt(dist1)) + mean(sqrt(dist2))
// in float32, widened into the float64 matrix.  dir[i][j] = the float32 mean over the points of superpoint i of sqrtf(min_b d2(a, b in j)), stored widened;
// the adjacency adds the two directions (in float64 here: 2^-24 relative from the reference's float32 addition, below what its own reduction order fixes —
// torch.mean on CUDA is a tree whose shape is the library's).  One wave per ordered pair (i, j); the target staged in LDS in slabs.
constexpr int C32_SLAB = 512;
__device__ void chamfer_dir_f32_body(const float* __restrict__ xyz, const int* __restrict__ sp_off, const int* __restrict__ sp_pts, const int* __restrict__ sel, int n,
                                     const double* __restrict__ centres, double* __restrict__ dir, float* slab /* [4][C32_SLAB * 3] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* tb = slab + (size_t)w * C32_SLAB * 3;
    for (long e = (long)blockIdx.x * 4 + w; e < (long)n * n; e += (long)gridDim.x * 4) {
        const int i = (int)(e / n), j = (int)(e % n);
        if (i == j) { if (lane == 0) dir[e] = 0.0; continue; }
        const int loi = sp_off[sel[i]], ni = sp_off[sel[i] + 1] - loi, loj = sp_off[sel[j]], nj = sp_off[sel[j] + 1] - loj;
        const double cix = centres[3 * (size_t)i], ciy = centres[3 * (size_t)i + 1], ciz = centres[3 * (size_t)i + 2];
        const double cjx = centres[3 * (size_t)j], cjy = centres[3 * (size_t)j + 1], cjz = centres[3 * (size_t)j + 2];
        float sum = 0.f;
        for (int a0 = 0; a0 < ni; a0 += 64) {
            const int a = a0 + lane;
            const size_t qa = sp_pts[loi + min(a, ni - 1)];
            const float ax = (float)((double)xyz[3 * qa] - cix), ay = (float)((double)xyz[3 * qa + 1] - ciy), az = (float)((double)xyz[3 * qa + 2] - ciz);
            float best = 3.402823466e+38f;
            for (int s0 = 0; s0 < nj; s0 += C32_SLAB) {
                const int cnt = min(nj - s0, C32_SLAB);
                (void)__ballot(1);          // the wave's lanes are done with the previous slab (a wave runs its LDS operations in order: this orders the CPU logic build's fibers)
                for (int t = lane; t < cnt; t += 64) {
                    const size_t qb = sp_pts[loj + s0 + t];
                    tb[3 * t] = (float)((double)xyz[3 * qb] - cjx); tb[3 * t + 1] = (float)((double)xyz[3 * qb + 1] - cjy); tb[3 * t + 2] = (float)((double)xyz[3 * qb + 2] - cjz);
                }
                (void)__ballot(1);
                for (int k = 0; k < cnt; ++k) {
                    const float dx = tb[3 * k] - ax, dy = tb[3 * k + 1] - ay, dz = tb[3 * k + 2] - az;      // chamfer3D.cu: b - a, every product and sum rounded on its own
                    const float dd = (dx * dx + dy * dy) + dz * dz;
                    best = (dd < best || (s0 + k == 0)) ? dd : best;
                }
            }
            sum += a < ni ? sqrtf(best) : 0.f;          // a lane's points in ascending order, then the lanes in a fixed tree
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) dir[e] = ni > 0 ? (double)(sum / (float)ni) : 0.0;
    }
}
__global__ __launch_bounds__(256) void sel_chamfer_dir_f32(const float* __restrict__ xyz, const int* __restrict__ sp_off, const int* __restrict__ sp_pts, const int* __restrict__ sel, int n,
                                                           const double* __restrict__ centres, double* __restrict__ dir) {
    __shared__ float slab[4 * C32_SLAB * 3];
    chamfer_dir_f32_body(xyz, sp_off, sp_pts, sel, n, centres, dir, slab);
}
__global__ __launch_bounds__(256) void sel_chamfer_dir_f32_batch(const float* __restrict__ xyz, const int* __restrict__ sp_off, const int* __restrict__ sp_pts, const int* __restrict__ sel,
                                                                 const int* __restrict__ coff, const long long* __restrict__ boff, const double* __restrict__ centres, double* __restrict__ dir) {
    __shared__ float slab[4 * C32_SLAB * 3];
    const int c = blockIdx.z, lo = coff[c], n = coff[c + 1] - lo;
    chamfer_dir_f32_body(xyz, sp_off, sp_pts, sel + lo, n, centres + 3 * (size_t)lo, dir + boff[c], slab);
}
static std::atomic<int> g_chamfer_mode{0};

bool chamfer_f64() { const char* e = getenv("SSDR_CHAMFER_F64"); return e && atoi(e) != 0; }      // read per launch: the tests run both forms in one process
// slices of the source items per target (blockIdx.y): a workgroup stages its target once and its four waves take items slice * 4 + wave, + 4 * slices, ...
// (measured for the float64 kernel, tools/gpu_chamfer_slices.sh, the bench's ~35 items per cloud: 16 slices 0.491-0.495 ms, 8: 0.481-0.484, 4: 0.53, 2: 0.62, 1: 0.83)
int chamfer_slices(int nm) {
    static const int env = [] { const char* e = getenv("SSDR_CHAMFER_SLICES"); return e ? atoi(e) : 0; }();
    return std::max(1, std::min((nm + 3) / 4, env > 0 ? env : 8));
}

}  // namespace

int chamfer_dir_launch(const float* d_xyz, const int* d_sp_off, const int* d_sp_pts, const int* d_sel, int n, const double* d_centres, double* d_dir,
                       const ChamferPack& P, hipStream_t s) {
    if (g_chamfer_mode.load() == 1) {
        hipLaunchKernelGGL(sel_chamfer_dir_f32, dim3((unsigned)std::min<long>(((long)n * n + 3) / 4, 65535)), dim3(256), 0, s, d_xyz, d_sp_off, d_sp_pts, d_sel, n, d_centres, d_dir);
        SSDR_HIP(hipGetLastError());
        return SSDR_OK;
    }
    const dim3 grid(std::min(n, 4096), std::max(1, std::min((n + 3) / 4, 16)));
    if (chamfer_f64()) hipLaunchKernelGGL(sel_chamfer_dir<false>, grid, dim3(256), 0, s, d_xyz, d_sp_off, d_sp_pts, d_sel, n, d_centres, d_dir, P);
    else hipLaunchKernelGGL(sel_chamfer_dir<true>, grid, dim3(256), 0, s, d_xyz, d_sp_off, d_sp_pts, d_sel, n, d_centres, d_dir, P);
    hipLaunchKernelGGL(sel_chamfer_big, dim3(16, std::min(8 * grid.y, 32u)), dim3(256), 0, s, d_xyz, d_sp_off, d_sp_pts, d_sel, n, d_centres, d_dir, P);      // the targets beyond the staging limit (usually none)
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int chamfer_dir_batch_launch(const float* d_xyz, const int* d_sp_off, const int* d_sp_pts, const int* d_sel, const int* d_coff, const long long* d_boff,
                             int n_max, unsigned nclouds, const double* d_centres, double* d_dir, const ChamferPack& P, hipStream_t s) {
    if (g_chamfer_mode.load() == 1) {
        hipLaunchKernelGGL(sel_chamfer_dir_f32_batch, dim3((unsigned)std::min<long>(((long)n_max * n_max + 3) / 4, 16384), 1, nclouds), dim3(256), 0, s, d_xyz, d_sp_off, d_sp_pts, d_sel,
                           d_coff, d_boff, d_centres, d_dir);
        SSDR_HIP(hipGetLastError());
        return SSDR_OK;
    }
    const dim3 grid(std::min(n_max, 1024), chamfer_slices(n_max), nclouds);
    if (chamfer_f64()) hipLaunchKernelGGL(sel_chamfer_dir_batch<false>, grid, dim3(256), 0, s, d_xyz, d_sp_off, d_sp_pts, d_sel, d_coff, d_boff, d_centres, d_dir, P);
    else hipLaunchKernelGGL(sel_chamfer_dir_batch<true>, grid, dim3(256), 0, s, d_xyz, d_sp_off, d_sp_pts, d_sel, d_coff, d_boff, d_centres, d_dir, P);
    hipLaunchKernelGGL(sel_chamfer_big_batch, dim3(16, std::min(8 * grid.y, 32u), grid.z), dim3(256), 0, s, d_xyz, d_sp_off, d_sp_pts, d_sel, d_coff, d_boff, d_centres, d_dir, P);      // the targets beyond the staging limit (usually none)
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

}  // namespace ssdr

/* Arithmetic of the chamfer term of the selection graph (ssdr_cloud_graph[_batch]_dev, ssdr_gcn_fps_sampling_dev and the sharded twins): 0 = float64 (the
 * S3DIS code: sklearn KDTree distances, fps_gcn_cpu.py:12-38 — the default), 1 = float32 as the Semantic3D code's CUDA kernel computes it
 * (SSRD_AL_semantic3d/fps_gcn_cuda.py:13-30: centred coordinates rounded to float32, squared float32 distances of chamfer3D.cu, sqrt and means in float32). */
extern "C" int ssdr_select_set_chamfer_mode(int mode) {
    if (mode < 0 || mode > 1) { ssdr::set_error("select_set_chamfer_mode: 0 (float64) or 1 (float32, CUDA-kernel flavour)"); return SSDR_ERR_INVALID; }
    ssdr::g_chamfer_mode.store(mode);
    return SSDR_OK;
}

