// Wave64 / workgroup primitives shared by the HIP translation units (gfx950).
#pragma once
#include "ssdr_internal.hpp"
#include <cfloat>

namespace ssdr {

constexpr int BS = 256;              // workgroup size of the build / scan style kernels
constexpr int U = 4;                 // positions per thread per sweep step
constexpr int CHUNK = BS * U;

// Wavefront reductions, result in every lane.  On gfx950 the partners come through DPP inside a row of 16 lanes (quad permutes,
// half-row and row mirrors) and through v_permlane16_swap / v_permlane32_swap across rows: pure vector-ALU instructions instead
// of six trips through the LDS crossbar (ds_bpermute) per reduction — the tree build does dozens of these per node.
#ifndef HIPEMU
template <class Op>
__device__ __forceinline__ unsigned wave_reduce_u32(unsigned x, Op op) {
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, true));      // quad_perm [2,3,0,1]
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, true));     // row_half_mirror
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, true));     // row_mirror
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);                                // rows 0<->1, 2<->3
    x = op(r[0], r[1]);
    r = __builtin_amdgcn_permlane32_swap(x, x, false, false);                                     // halves
    return op(r[0], r[1]);
}
__device__ __forceinline__ float wave_min(float v) {
    return __uint_as_float(wave_reduce_u32(__float_as_uint(v), [](unsigned a, unsigned b) { return __float_as_uint(fminf(__uint_as_float(a), __uint_as_float(b))); }));
}
__device__ __forceinline__ float wave_max(float v) {
    return __uint_as_float(wave_reduce_u32(__float_as_uint(v), [](unsigned a, unsigned b) { return __float_as_uint(fmaxf(__uint_as_float(a), __uint_as_float(b))); }));
}
__device__ __forceinline__ int wave_sum(int v) {
    return (int)wave_reduce_u32((unsigned)v, [](unsigned a, unsigned b) { return a + b; });
}
#else
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
#endif

// Launches whose grid is (blocks of one tile) x (tiles) and whose blocks gather rows of their tile's tables at random: blocks are dealt
// round-robin over the 8 XCDs in linear order (MI355X_MICROARCH.md, Workgroup dispatch), so with the plain mapping every XCD works on every
// tile and each of the eight L2s fetches every tile's table (measured: the LFA kernels read 8 x their gathered tables from beyond L2).  This
// mapping gives the blocks of a tile to ONE XCD (blocks b and b + 8 share an XCD): a tile's table (<= 4 MB) is fetched into one L2, once.
// Placement is a speed matter only.  Needs the tile count to be a multiple of 8; otherwise the identity.
__device__ __forceinline__ void xcd_tile_map(int& bx, int& tile) {
    const int gx = (int)gridDim.x, nb = (int)gridDim.y;
    bx = (int)blockIdx.x; tile = (int)blockIdx.y;
    if ((nb & 7) == 0) {
        const int lin = tile * gx + bx, xcd = lin & 7, j = lin >> 3;
        tile = xcd + 8 * (j / gx); bx = j % gx;
    }
}

// Orders the LDS traffic of the lanes of ONE wave (waves of a workgroup that take different trip counts cannot use the workgroup barrier):
// everything the wave's lanes wrote before is visible to its lanes after.
__device__ __forceinline__ void wave_sync() {
#ifndef HIPEMU
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
    (void)__shfl(0, 0);      // the CPU stand-in completes a wave operation once every live lane of the wave has reached it
#endif
}

// min/max of 3 coordinates over the workgroup; result broadcast to every thread.
template <int NT = BS>
__device__ inline void block_minmax3(float (&mn)[3], float (&mx)[3], float* s /* [NT/64][6] */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float a = wave_min(mn[d]), b = wave_max(mx[d]);
        if (lane == 0) { s[wid * 6 + d] = a; s[wid * 6 + 3 + d] = b; }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float a = s[d], b = s[3 + d];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) { a = fminf(a, s[w * 6 + d]); b = fmaxf(b, s[w * 6 + 3 + d]); }
        mn[d] = a; mx[d] = b;
    }
    __syncthreads();
}

template <int NT = BS>
__device__ inline void block_sum2(int& a, int& b, int* s /* [NT/64][2] */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int x = wave_sum(a), y = wave_sum(b);
    if (lane == 0) { s[wid * 2] = x; s[wid * 2 + 1] = y; }
    __syncthreads();
    x = 0; y = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { x += s[w * 2]; y += s[w * 2 + 1]; }
    a = x; b = y;
    __syncthreads();
}

// Writes every position i in [lo,hi) with pred(i) to dst(rank) where rank counts matches in increasing i.
// Returns the number of matches (uniform).  All threads of the workgroup must call it.
template <int NT = BS, class Pred, class Dst>
__device__ int block_compact(int lo, int hi, Pred pred, Dst dst, int (*s_w)[U][NT / 64] /* [2][U][NT/64] */) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int running = 0, par = 0;
    for (int base = lo; base < hi; base += NT * U, par ^= 1) {
        bool p[U]; unsigned long long m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int i = base + u * NT + tid;
            p[u] = (i < hi) && pred(i);
            m[u] = __ballot(p[u]);
            if (lane == 0) s_w[par][u][wid] = __popcll(m[u]);
        }
        __syncthreads();
        int off = running;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int wbase = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) { int c = s_w[par][u][w]; if (w < wid) wbase += c; tot += c; }
            if (p[u]) dst(off + wbase + __popcll(m[u] & lt), base + u * NT + tid);
            off += tot;
        }
        running = off;
    }
    __syncthreads();
    return running;
}

}  // namespace ssdr
