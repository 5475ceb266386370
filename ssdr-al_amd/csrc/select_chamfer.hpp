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
    float* r2item;                        // per item: max |p|^2 over its points, rounded up (0 where unused) — the screening's error bound
    float* r2sp;                          // per superpoint: the same over its own points
    int* item_slot;                       // per item: its first slot
    int* big;                             // superpoints taken pair by pair
    int* start;                           // per superpoint: first slot, -1 for the pair-by-pair ones
    int* counts;                          // per cloud: items, pair-by-pair superpoints
};

__device__ __forceinline__ ChamferPack pack_at(ChamferPack P, int row0) {
    const size_t s0 = (size_t)ITEM * (size_t)row0;
    P.x += s0; P.y += s0; P.z += s0; P.seg += s0; P.cnt += s0; P.r2item += row0; P.r2sp += row0; P.item_slot += row0; P.big += row0; P.start += row0;
    return P;
}

// dir[i*n + j] of one cloud / of every cloud of a batch (blockIdx.z = cloud; coff[c] = first row of cloud c in sel / centres, boff[c] = first element of
// its n_c x n_c block in dir).  n_max = the largest cloud's superpoints.
int chamfer_dir_launch(const float* d_xyz, const int* d_sp_off, const int* d_sp_pts, const int* d_sel, int n, const double* d_centres, double* d_dir,
                       const ChamferPack& P, hipStream_t s);
int chamfer_dir_batch_launch(const float* d_xyz, const int* d_sp_off, const int* d_sp_pts, const int* d_sel, const int* d_coff, const long long* d_boff,
                             int n_max, unsigned nclouds, const double* d_centres, double* d_dir, const ChamferPack& P, hipStream_t s);

}  // namespace ssdr
