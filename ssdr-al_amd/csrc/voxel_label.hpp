// Per-voxel majority label with the reference's tie rule, shared by the voxel reductions (subsample.hip, frontend.hip).
//
// grid_subsampling.cpp:97-101 over grid_subsampling.h:19,46-49: the per-voxel histogram is an unordered_map<int,int>; std::max_element
// returns the first maximum in *iteration order*.  The routines below take the voxel's labels through a getter `getl(j)`, j in [s, e),
// in INPUT order (the order the reference inserts them in).
#pragma once
#include "ssdr_internal.hpp"

namespace ssdr {

constexpr int LAB_CAP = 29;
constexpr int GS_UNROLL = 8;    // loads in flight per lane in the voxel reductions (a lane's loop is a latency chain otherwise)

__device__ __forceinline__ unsigned lab_bucket(int l, unsigned nb) { return (unsigned)((unsigned long long)(long long)l % nb); }

// The list below is kept in the iteration order of the reference's map (13 -> 29 buckets, most recently first-seen label first).
template <class GetL>
__device__ int voxel_label_t(GetL getl, int s, int e, int* status) {
    int lab[LAB_CAP], cnt[LAB_CAP];
    int nl = 0; unsigned nbk = 13;
    for (int j = s; j < e; ++j) {
        const int L = getl(j);
        int t = 0;
        for (; t < nl; ++t) if (lab[t] == L) { cnt[t]++; break; }
        if (t < nl) continue;
        if (nl == 13 && nbk == 13) {
            // rehash 13 -> 29 before the 14th insert: runs by first occurrence, members in list order, all reversed
            int tl[13], tc[13]; unsigned used = 0; int w = 0;
            for (int i = 0; i < 13; ++i) if (!((used >> i) & 1)) {
                const unsigned b = lab_bucket(lab[i], 29);
                for (int k = i; k < 13; ++k) if (!((used >> k) & 1) && lab_bucket(lab[k], 29) == b) { used |= 1u << k; tl[w] = lab[k]; tc[w] = cnt[k]; ++w; }
            }
            for (int i = 0; i < 13; ++i) { lab[i] = tl[12 - i]; cnt[i] = tc[12 - i]; }
            nbk = 29;
        }
        if (nl == LAB_CAP) { atomicOr(status, 1); break; }
        const unsigned b = lab_bucket(L, nbk);
        int pos = 0;
        for (int i = 0; i < nl; ++i) if (lab_bucket(lab[i], nbk) == b) { pos = i; break; }
        for (int i = nl; i > pos; --i) { lab[i] = lab[i - 1]; cnt[i] = cnt[i - 1]; }
        lab[pos] = L; cnt[pos] = 1; ++nl;
    }
    int best = 0;
    for (int t = 1; t < nl; ++t) if (cnt[best] < cnt[t]) best = t;
    return nl ? lab[best] : 0;
}

// Fast path for labels in [0,13): they hash to distinct buckets of the 13-bucket table and can never trigger the rehash, so the
// reference's iteration order is simply "most recently first-seen first" and the first maximum is the largest count, ties to the label
// first seen LAST.  Returns -1 when a label falls outside [0,13).
template <class GetL>
__device__ __forceinline__ int voxel_label_fast_t(GetL getl, int s, int e) {
    int cnt[13], seen[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) { cnt[k] = 0; seen[k] = -1; }
    for (int j0 = s; j0 < e; j0 += GS_UNROLL) {
        int lv[GS_UNROLL];
#pragma unroll
        for (int u = 0; u < GS_UNROLL; ++u) lv[u] = getl(min(j0 + u, e - 1));
#pragma unroll
        for (int u = 0; u < GS_UNROLL; ++u) {
            const int j = j0 + u, L = lv[u];
            if (j < e) {
                if (L < 0 || L >= 13) return -1;
#pragma unroll
                for (int k = 0; k < 13; ++k) if (L == k) { if (cnt[k] == 0) seen[k] = j; cnt[k]++; }
            }
        }
    }
    int best = 0;
#pragma unroll
    for (int k = 1; k < 13; ++k) if (cnt[k] > cnt[best] || (cnt[k] == cnt[best] && seen[k] > seen[best])) best = k;
    return best;
}

// The vote of one label column of one voxel from packed byte counters (labels 0..12, at most 255 points): the caller adds
// 1 << 8 (L & 7) to pk0 (L < 8) or pk1 (8 <= L < 16) per member and sets `exact` for a label >= 13 or more than 255 members.
// Only a voxel whose maximum is shared by two labels (the tie goes by first-seen order), or one outside those limits, is re-scanned.
template <class GetL>
__device__ __forceinline__ int voxel_label_packed(unsigned long long pk0, unsigned long long pk1, bool exact, GetL getl, int s, int e, int* status) {
    int best = 0, bestc = -1, nbest = 0;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        const int ck = (int)(((k < 8 ? pk0 : pk1) >> ((k & 7) * 8)) & 0xffull);
        if (ck > bestc) { bestc = ck; best = k; nbest = 1; } else if (ck == bestc) ++nbest;
    }
    if (exact) {
        best = voxel_label_fast_t(getl, s, e);
        if (best < 0) best = voxel_label_t(getl, s, e, status);
    } else if (nbest > 1) {                              // shared maximum: the label first seen LAST wins (see voxel_label_fast_t)
        unsigned seen = 0;
        for (int j = s; j < e; ++j) {
            const unsigned L = (unsigned)getl(j);
            const int ck = (int)(((L < 8 ? pk0 : pk1) >> ((L & 7) * 8)) & 0xffull);
            if (ck == bestc && !((seen >> L) & 1u)) { seen |= 1u << L; best = (int)L; }
        }
    }
    return best;
}

}  // namespace ssdr
