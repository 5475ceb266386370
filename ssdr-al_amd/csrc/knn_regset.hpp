// KNNResultSet (nanoflann.hpp:36-102) in registers, shared by the kd-tree walk and the grid search.
#pragma once
#include "ssdr_internal.hpp"
#include <cfloat>

namespace ssdr {

#ifndef HIPEMU
__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
#else
static inline float med3(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
#endif

template <int K>
struct RegSet {
    float d[K]; int id[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j) { d[j] = FLT_MAX; id[j] = 0; }
    }
    __device__ __forceinline__ float worst() const { return d[K - 1]; }
    __device__ __forceinline__ void add(float dist, int index) {   // requires dist < worst()
        // sorted insertion after the elements that are <= dist (KNNResultSet::addPoint :63-92): slot j takes the median of
        // (d[j-1], dist, d[j]) — one v_med3_f32 — and the id follows from the comparisons c_j = d[j] > dist
        bool c[K];
#pragma unroll
        for (int j = 0; j < K; ++j) c[j] = d[j] > dist;
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            id[j] = c[j - 1] ? id[j - 1] : (c[j] ? index : id[j]);
            d[j] = med3(d[j - 1], dist, d[j]);
        }
        if (c[0]) { d[0] = dist; id[0] = index; }
    }
    __device__ __forceinline__ int get(int j) const { return id[j]; }
};

}  // namespace ssdr
