// Types shared by the two grid-subsample implementations (subsample.hip: sort-based, any grid; frontend.hip: bucket partition + LDS, room-scale grids).
#pragma once
#include "ssdr_internal.hpp"
#include "block_prims.hpp"

namespace ssdr {

constexpr int REC_W = 8;        // words of a packed point record (rows of at most 8 words: the hot path's 3 + 3 + 1)

struct GsParams {
    float org[3]; float dl;
    unsigned long long nx, ny;
    int m;            // number of voxels
    int status;       // 1 = more than LAB_CAP distinct labels in one voxel
    unsigned long long key_and, key_or;      // what the sorter needs to know about the keys: no bit is set in all of them / only these can be set
};

__device__ __forceinline__ void gs_minmax_partial_body(const float* __restrict__ P, int n, float* partial) {
    __shared__ float s_mm[(BS / 64) * 6];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * BS + threadIdx.x; i < n; i += gridDim.x * BS) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { float v = P[3 * (size_t)i + d]; mn[d] = fminf(mn[d], v); mx[d] = fmaxf(mx[d], v); }
    }
    block_minmax3(mn, mx, s_mm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { partial[6 * blockIdx.x + d] = mn[d]; partial[6 * blockIdx.x + 3 + d] = mx[d]; }
    }
}

// Per-cloud tables of a batch, passed by value.  Cloud r: input rows [off[r], off[r+1]) of the concatenated arrays,
// sort slots [toff[r], toff[r+1]) (tile-aligned), segment-start slots from toff[r] + r.
struct CloudTab { int nr; int off[RADIX_MAX_SEG + 1]; int toff[RADIX_MAX_SEG + 1]; };
constexpr int PB = 256;     // partial min/max blocks per cloud


}  // namespace ssdr
