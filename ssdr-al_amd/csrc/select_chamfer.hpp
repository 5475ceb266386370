// Chamfer stage of the selection (F1, fps_gcn_cpu.py:12-38, 84-100): what select.hip (packer, launch sites) and select_chamfer.hip (the
// distance kernels) share.
#pragma once
#include "ssdr_internal.hpp"

namespace ssdr {

constexpr int CH_TILE = 640;    // target points staged per step; at most 1024 (the float64 screening key carries a 10-bit index)
constexpr int PACK_MAX = 4096;  // superpoints of one cloud the packer lays out (two int tables in LDS)
constexpr int ITEM = 256;       // source points one wave takes against a target
constexpr int NV = ITEM / 64;   // ... per lane
constexpr int SEQ_MAX = 16;     // superpoints up to this size are summed by one lane each, larger ones by the whole wave

// The staged target is the same for every lane, and a wave-wide LDS read of 24 bytes per lane costs the LDS pipe 12 cycles whether
// or not the addresses agree: with one source point per lane the kernel waited on LDS (33 % instruction issue), and a wave per
// (source, target) pair left the lanes beyond the source's size idle.  Hence:
//   * sel_chamfer_plan / _fill lay the centred points of the superpoints of a cloud out in 256-slot ITEMS — as many whole superpoints as
//     fit, never split — and the distance kernel takes one item per wave and target;
//   * the roots of a superpoint's points are added up in an order that depends on its size alone (segment sums), so the mean does
//     not depend on what else shares the item;
//   * superpoints above 256 points (and empty ones) are taken pair by pair in passes of 256 with the same summation rule.
struct ChamferPack {
    double* x; double* y; double* z;      // per slot: the centred point
    int* seg; int* cnt;                   // per slot: local index of its superpoint (-1: padding); the superpoint's size on its first slot, else 0
    uint4* src0; unsigned long long* src1; // per slot: the point as the screening's source operand, lower / upper lane half (source_operand)
    float* r2item;                        // per item: max |p|^2 over its points, rounded up (0 where unused) — the screening's error bound
    float* r2sp;                          // per superpoint: the same over its own points
    int* item_slot;                       // per item: its first slot
    int* big;                             // superpoints taken pair by pair
    int* start;                           // per superpoint: first slot, -1 for the pair-by-pair ones
    int* counts;                          // per cloud: items, pair-by-pair superpoints
};

__device__ __forceinline__ ChamferPack pack_at(ChamferPack P, int row0) {
    const size_t s0 = (size_t)ITEM * (size_t)row0;
    P.x += s0; P.y += s0; P.z += s0; P.src0 += s0; P.src1 += s0; P.seg += s0; P.cnt += s0; P.r2item += row0; P.r2sp += row0; P.item_slot += row0; P.big += row0; P.start += row0;
    return P;
}

// ---- half-precision pieces for the screening on the matrix cores (select_chamfer.hip) ----------------------------------------------------------------
constexpr float MF_SCALE = 128.0f;          // coordinates are screened as 128 x: pieces of centimetre-scale coordinates stay normal half-precision numbers
constexpr float MF_R2_MAX = 1000.0f;        // |p|^2 (m^2) up to which a superpoint is screened: 256 |p| and 4 |p|^2 fit half precision, padding keys stay above every real one
#ifndef HIPEMU
__device__ __forceinline__ unsigned f16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float f16_val(unsigned b) { return (float)__builtin_bit_cast(_Float16, (unsigned short)b); }
#else
static inline unsigned f16_bits(float x) { return hipemu_f32_to_f16(x); }
static inline float f16_val(unsigned b) { return hipemu_f16_to_f32((unsigned short)b); }
#endif
// x = hi + lo (+ 2^-22 |x|): two half-precision pieces, as bit patterns
__device__ __forceinline__ void split16(float x, unsigned& hi, unsigned& lo) { hi = f16_bits(x); lo = f16_bits(x - f16_val(hi)); }
// a source point's eight k-slots for lane half h (table in select_chamfer.hip): h = 0 from (ax, ay), h = 1 from (az, -)
__device__ __forceinline__ uint4 source_operand(double u, double v, int h) {
    unsigned uh, ul, vh, vl;
    split16((float)(u * (double)(-2.0f * MF_SCALE)), uh, ul); split16((float)(v * (double)(-2.0f * MF_SCALE)), vh, vl);
    const unsigned one = 0x6c00u;                                               // 4096.0
    return uint4{uh | (ul << 16), h ? uh : (uh | (vh << 16)), h ? 0u : (vl | (vh << 16)), h ? 0u : (one | (one << 16))};
}

// dir[i*n + j] of one cloud / of every cloud of a batch (blockIdx.z = cloud; coff[c] = first row of cloud c in sel / centres, boff[c] = first element of
// its n_c x n_c block in dir).  n_max = the largest cloud's superpoints.
int chamfer_dir_launch(const float* d_xyz, const int* d_sp_off, const int* d_sp_pts, const int* d_sel, int n, const double* d_centres, double* d_dir,
                       const ChamferPack& P, hipStream_t s);
int chamfer_dir_batch_launch(const float* d_xyz, const int* d_sp_off, const int* d_sp_pts, const int* d_sel, const int* d_coff, const long long* d_boff,
                             int n_max, unsigned nclouds, const double* d_centres, double* d_dir, const ChamferPack& P, hipStream_t s);

}  // namespace ssdr
