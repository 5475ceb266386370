/* TEST INFRASTRUCTURE ONLY — see oracle.h.
 *
 * CPU restatement of the reference KNN op: an exact kd-tree nearest-neighbour search
 * whose *tie order* is the tree traversal order, so the tree must be built and walked
 * exactly like nanoflann v1.2.3 does (paths relative to
 * /root/reference/SSDR_AL_s3dis/utils/nearest_neighbors/):
 *
 *   build   nanoflann.hpp:848-896 (divideTree), :898-937 (middleSplit_), :948-975 (planeSplit),
 *           :1136-1146 (buildIndex), :1241-1262 (computeBoundingBox); leaf size 10
 *           (knn_.cxx:28, KDTreeTableAdaptor.h:134-139)
 *   search  nanoflann.hpp:1164-1178 (findNeighbors), :977-993 (computeInitialDistances),
 *           :1271-1329 (searchLevel, eps = 0), :36-102 (KNNResultSet), :280-304 (evalMetric)
 *   drivers knn_.cxx:22-44, :72-135
 *
 * All arithmetic is IEEE fp32 without contraction (the file is compiled with
 * -ffp-contract=off), matching gcc -O2 on x86-64 SSE2.
 */
#include "oracle.h"
#include <float.h>
#include <stdlib.h>
#include <string.h>

#define MAXD 16

typedef struct {
    int32_t left, right;      /* leaf: [left,right) into vind */
    int32_t divfeat;          /* internal */
    int32_t child1, child2;   /* -1/-1 for a leaf */
    float divlow, divhigh;
} knode;

typedef struct {
    const float* pts; size_t n, dim, leaf_max;
    size_t* vind;
    knode* nodes; size_t nnodes, cap;
    float root_lo[MAXD], root_hi[MAXD];
} ktree;

static inline float pt(const ktree* t, size_t idx, int d) { return t->pts[idx * t->dim + d]; }

static void minmax(const ktree* t, const size_t* ind, size_t count, int d, float* mn, float* mx)
{
    float lo = pt(t, ind[0], d), hi = lo;
    for (size_t i = 1; i < count; ++i) {
        float v = pt(t, ind[i], d);
        if (v < lo) lo = v;
        if (v > hi) hi = v;
    }
    *mn = lo; *mx = hi;
}

/* nanoflann.hpp:948-975.  Two Hoare-style sweeps; the guard on `right` reproduces the
 * unsigned-index quirk (position 0 is never examined from the right). */
static void plane_split(const ktree* t, size_t* ind, size_t count, int cutfeat, float cutval,
                        size_t* lim1, size_t* lim2)
{
    size_t left = 0, right = count - 1;
    for (;;) {
        while (left <= right && pt(t, ind[left], cutfeat) < cutval) ++left;
        while (right && left <= right && pt(t, ind[right], cutfeat) >= cutval) --right;
        if (left > right || !right) break;
        size_t tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim1 = left;
    right = count - 1;
    for (;;) {
        while (left <= right && pt(t, ind[left], cutfeat) <= cutval) ++left;
        while (right && left <= right && pt(t, ind[right], cutfeat) > cutval) --right;
        if (left > right || !right) break;
        size_t tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim2 = left;
}

static int32_t new_node(ktree* t)
{
    if (t->nnodes == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nodes = (knode*)realloc(t->nodes, t->cap * sizeof(knode));
    }
    return (int32_t)t->nnodes++;
}

/* nanoflann.hpp:848-896.  lo/hi: in = clipped box handed down, out = tight box handed back. */
static int32_t divide(ktree* t, size_t left, size_t right, float* lo, float* hi)
{
    const int D = (int)t->dim;
    int32_t id = new_node(t);
    if (right - left <= t->leaf_max) {
        knode nd; nd.child1 = nd.child2 = -1; nd.left = (int32_t)left; nd.right = (int32_t)right;
        nd.divfeat = 0; nd.divlow = nd.divhigh = 0.f;
        for (int i = 0; i < D; ++i) lo[i] = hi[i] = pt(t, t->vind[left], i);
        for (size_t k = left + 1; k < right; ++k)
            for (int i = 0; i < D; ++i) {
                float v = pt(t, t->vind[k], i);
                if (lo[i] > v) lo[i] = v;
                if (hi[i] < v) hi[i] = v;
            }
        t->nodes[id] = nd;
        return id;
    }
    /* middleSplit_ :898-937 */
    size_t* ind = t->vind + left; size_t count = right - left;
    const float EPS = 0.00001f;
    float max_span = hi[0] - lo[0];
    for (int i = 1; i < D; ++i) { float s = hi[i] - lo[i]; if (s > max_span) max_span = s; }
    float max_spread = -1.f; int cutfeat = 0;
    for (int i = 0; i < D; ++i) {
        float s = hi[i] - lo[i];
        if (s > (1 - EPS) * max_span) {
            float mn, mx; minmax(t, ind, count, i, &mn, &mx);
            float spread = mx - mn;
            if (spread > max_spread) { cutfeat = i; max_spread = spread; }
        }
    }
    float split_val = (lo[cutfeat] + hi[cutfeat]) / 2;
    float mn, mx; minmax(t, ind, count, cutfeat, &mn, &mx);
    float cutval = split_val < mn ? mn : (split_val > mx ? mx : split_val);
    size_t lim1, lim2, idx;
    plane_split(t, ind, count, cutfeat, cutval, &lim1, &lim2);
    if (lim1 > count / 2) idx = lim1; else if (lim2 < count / 2) idx = lim2; else idx = count / 2;

    float llo[MAXD], lhi[MAXD], rlo[MAXD], rhi[MAXD];
    memcpy(llo, lo, sizeof(float) * D); memcpy(lhi, hi, sizeof(float) * D);
    memcpy(rlo, lo, sizeof(float) * D); memcpy(rhi, hi, sizeof(float) * D);
    lhi[cutfeat] = cutval; rlo[cutfeat] = cutval;
    int32_t c1 = divide(t, left, left + idx, llo, lhi);
    int32_t c2 = divide(t, left + idx, right, rlo, rhi);
    knode nd; nd.left = (int32_t)left; nd.right = (int32_t)right; nd.divfeat = cutfeat;
    nd.child1 = c1; nd.child2 = c2; nd.divlow = lhi[cutfeat]; nd.divhigh = rlo[cutfeat];
    t->nodes[id] = nd;
    for (int i = 0; i < D; ++i) {
        lo[i] = llo[i] < rlo[i] ? llo[i] : rlo[i];
        hi[i] = lhi[i] > rhi[i] ? lhi[i] : rhi[i];
    }
    return id;
}

static void tree_build(ktree* t, const float* pts, size_t n, size_t dim, size_t leaf_max)
{
    memset(t, 0, sizeof(*t));
    t->pts = pts; t->n = n; t->dim = dim; t->leaf_max = leaf_max;
    t->vind = (size_t*)malloc((n ? n : 1) * sizeof(size_t));
    for (size_t i = 0; i < n; ++i) t->vind[i] = i;
    if (!n) return;
    for (size_t d = 0; d < dim; ++d) t->root_lo[d] = t->root_hi[d] = pts[d];
    for (size_t k = 1; k < n; ++k)
        for (size_t d = 0; d < dim; ++d) {
            float v = pts[k * dim + d];
            if (v < t->root_lo[d]) t->root_lo[d] = v;
            if (v > t->root_hi[d]) t->root_hi[d] = v;
        }
    float lo[MAXD], hi[MAXD];
    memcpy(lo, t->root_lo, sizeof(lo)); memcpy(hi, t->root_hi, sizeof(hi));
    divide(t, 0, n, lo, hi);
    /* divideTree hands the tight box back into root_bbox (passed by reference, :1145) */
    memcpy(t->root_lo, lo, sizeof(lo)); memcpy(t->root_hi, hi, sizeof(hi));
}

static void tree_free(ktree* t) { free(t->vind); free(t->nodes); }

/* nanoflann.hpp:280-304 */
static inline float eval_metric(const ktree* t, const float* a, size_t b)
{
    float result = 0.f;
    const float* last = a + t->dim; const float* lastgroup = last - 3; size_t d = 0;
    while (a < lastgroup) {
        float d0 = a[0] - pt(t, b, (int)d++), d1 = a[1] - pt(t, b, (int)d++);
        float d2 = a[2] - pt(t, b, (int)d++), d3 = a[3] - pt(t, b, (int)d++);
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        a += 4;
    }
    while (a < last) { float d0 = *a++ - pt(t, b, (int)d++); result += d0 * d0; }
    return result;
}

typedef struct { size_t* ids; float* dists; size_t cap, count; } rset;

/* nanoflann.hpp:63-92 */
static inline void rs_add(rset* r, float dist, size_t index)
{
    size_t i;
    for (i = r->count; i > 0; --i) {
        if (r->dists[i - 1] > dist) {
            if (i < r->cap) { r->dists[i] = r->dists[i - 1]; r->ids[i] = r->ids[i - 1]; }
        } else break;
    }
    if (i < r->cap) { r->dists[i] = dist; r->ids[i] = index; }
    if (r->count < r->cap) r->count++;
}

/* nanoflann.hpp:1271-1329 with epsError == 1 */
static void search_level(const ktree* t, rset* r, const float* vec, int32_t nid, float mindistsq, float* dists)
{
    const knode* nd = &t->nodes[nid];
    if (nd->child1 < 0) {
        float worst = r->dists[r->cap - 1];
        for (int32_t i = nd->left; i < nd->right; ++i) {
            size_t index = t->vind[i];
            float dist = eval_metric(t, vec, index);
            if (dist < worst) rs_add(r, dist, index);
        }
        return;
    }
    int idx = nd->divfeat;
    float val = vec[idx];
    float diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
    int32_t best, other; float cut;
    if ((diff1 + diff2) < 0) { best = nd->child1; other = nd->child2; cut = (val - nd->divhigh) * (val - nd->divhigh); }
    else                     { best = nd->child2; other = nd->child1; cut = (val - nd->divlow) * (val - nd->divlow); }
    search_level(t, r, vec, best, mindistsq, dists);
    float dst = dists[idx];
    mindistsq = mindistsq + cut - dst;
    dists[idx] = cut;
    if (mindistsq * 1.0f <= r->dists[r->cap - 1]) search_level(t, r, vec, other, mindistsq, dists);
    dists[idx] = dst;
}

static void tree_query(const ktree* t, const float* q, size_t K, size_t* ids, float* ds)
{
    rset r; r.ids = ids; r.dists = ds; r.cap = K; r.count = 0;
    if (K) ds[K - 1] = FLT_MAX;
    if (!t->n || !K) return;
    float dists[MAXD]; float distsq = 0.f;
    for (size_t i = 0; i < t->dim; ++i) {
        dists[i] = 0.f;
        if (q[i] < t->root_lo[i]) { dists[i] = (q[i] - t->root_lo[i]) * (q[i] - t->root_lo[i]); distsq += dists[i]; }
        if (q[i] > t->root_hi[i]) { dists[i] = (q[i] - t->root_hi[i]) * (q[i] - t->root_hi[i]); distsq += dists[i]; }
    }
    search_level(t, &r, q, 0, distsq, dists);
}

void oracle_knn(const float* pts, size_t npts, size_t dim, const float* queries, size_t nq, size_t K,
                int64_t* out_idx, float* out_dist)
{
    ktree t; tree_build(&t, pts, npts, dim, 10);
    /* knn_.cxx:30-31: one zero-initialised id/dist buffer reused by every query, so slots
     * that no query ever fills (K > npts) stay 0. */
    size_t* ids = (size_t*)calloc(K ? K : 1, sizeof(size_t));
    float* ds = (float*)calloc(K ? K : 1, sizeof(float));
    for (size_t i = 0; i < nq; ++i) {
        tree_query(&t, queries + i * dim, K, ids, ds);
        for (size_t j = 0; j < K; ++j) {
            out_idx[i * K + j] = (int64_t)ids[j];
            if (out_dist) out_dist[i * K + j] = ds[j];
        }
    }
    free(ids); free(ds); tree_free(&t);
}

void oracle_knn_batch(const float* pts, size_t batch, size_t npts, size_t dim, const float* queries, size_t nq,
                      size_t K, int64_t* out_idx, int threads)
{
    /* knn_.cxx:104-135: OpenMP over the batch axis only. */
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (long b = 0; b < (long)batch; ++b)
        oracle_knn(pts + (size_t)b * npts * dim, npts, dim, queries + (size_t)b * nq * dim, nq, K,
                   out_idx + (size_t)b * nq * K, NULL);
}

long oracle_kdtree_dump(const float* pts, size_t npts, size_t dim, size_t leaf_max,
                        int64_t* vind, int32_t* node_i5, float* node_f2, size_t node_cap)
{
    ktree t; tree_build(&t, pts, npts, dim, leaf_max);
    for (size_t i = 0; i < npts; ++i) vind[i] = (int64_t)t.vind[i];
    for (size_t i = 0; i < t.nnodes && i < node_cap; ++i) {
        node_i5[i * 5 + 0] = t.nodes[i].left; node_i5[i * 5 + 1] = t.nodes[i].right;
        node_i5[i * 5 + 2] = t.nodes[i].divfeat; node_i5[i * 5 + 3] = t.nodes[i].child1;
        node_i5[i * 5 + 4] = t.nodes[i].child2;
        node_f2[i * 2 + 0] = t.nodes[i].divlow; node_f2[i * 2 + 1] = t.nodes[i].divhigh;
    }
    long nn = (long)t.nnodes; tree_free(&t); return nn;
}

/* ---- knn_batch_distance_pick (knn_.cxx:136-203) -----------------------------------------------------------------
 * Sequential "least used points first" query picking; the reference seeds std::mt19937 with time(0) and draws ONE
 * number per query, batch elements in order.  MT19937 restated below (Matsumoto & Nishimura 1998, the algorithm
 * std::mt19937 is specified to be); the seed is an argument. */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;
static void mt_seed(mt19937_t* g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(mt19937_t* g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

void oracle_knn_batch_distance_pick(const float* pts, size_t batch, size_t npts, size_t dim, float* out_queries, size_t nq,
                                    size_t K, int64_t* out_idx, uint32_t seed)
{
    mt19937_t g; mt_seed(&g, seed);
    size_t* ids = (size_t*)calloc(K ? K : 1, sizeof(size_t));
    float* ds = (float*)calloc(K ? K : 1, sizeof(float));
    int* used = (int*)malloc(sizeof(int) * (npts ? npts : 1));
    size_t* poss = (size_t*)malloc(sizeof(size_t) * (npts ? npts : 1));
    for (size_t b = 0; b < batch; ++b) {
        const float* P = pts + b * npts * dim;
        ktree t; tree_build(&t, P, npts, dim, 10);
        for (size_t i = 0; i < npts; ++i) used[i] = 0;
        int current = 0;
        for (size_t q = 0; q < nq; ++q) {
            size_t np_ = 0;
            while (np_ == 0) {                                           /* :155-165 */
                for (size_t i = 0; i < npts; ++i) if (used[i] == current) poss[np_++] = i;
                if (np_ == 0) { current = used[0]; for (size_t i = 1; i < npts; ++i) if (used[i] < current) current = used[i]; }
            }
            const size_t index = poss[mt_next(&g) % np_];                /* :168 */
            /* :177-180: fresh (zeroed) id / dist vectors for every query */
            for (size_t j = 0; j < K; ++j) { ids[j] = 0; ds[j] = 0.f; }
            tree_query(&t, P + index * dim, K, ids, ds);
            for (size_t j = 0; j < K; ++j) used[ids[j]]++;               /* :182-184 */
            used[index] += 100;                                          /* :185 */
            for (size_t j = 0; j < K; ++j) out_idx[(b * nq + q) * K + j] = (int64_t)ids[j];
            for (size_t d = 0; d < dim; ++d) out_queries[(b * nq + q) * dim + d] = P[index * dim + d];
        }
        tree_free(&t);
    }
    free(ids); free(ds); free(used); free(poss);
}
