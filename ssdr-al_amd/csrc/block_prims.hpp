// Wave64 / workgroup primitives shared by the HIP translation units (gfx950).
#pragma once
#include "ssdr_internal.hpp"
#include <cfloat>

namespace ssdr {

constexpr int BS = 256;              // workgroup size of the build / scan style kernels
constexpr int U = 4;                 // positions per thread per sweep step
constexpr int CHUNK = BS * U;

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
