/* TEST INFRASTRUCTURE ONLY — see oracle.h.
 *
 * CPU restatement of the reference voxel-grid barycentre subsampling
 * (/root/reference/SSDR_AL_s3dis/utils/cpp_wrappers/cpp_subsampling/grid_subsampling/
 *  grid_subsampling.cpp:5-106, grid_subsampling.h:10-80, ../cpp_utils/cloud/cloud.cpp:27-67,
 *  cloud.h:120-143).
 *
 * Two things in the reference are properties of libstdc++'s std::unordered_map rather than
 * of the algorithm, and are emulated here so the restatement is comparable row for row:
 *   (1) the output row order = iteration order of unordered_map<size_t,SampledData>
 *       (grid_subsampling.cpp:85);
 *   (2) the label of a voxel = first maximum in iteration order of the per-voxel
 *       unordered_map<int,int> histogram (grid_subsampling.cpp:97-101).
 * Emulated container behaviour (GCC 11 libstdc++, identity hash, max_load_factor 1):
 *   - bucket-count schedule 1 -> 13 -> 29 -> 59 -> 127 -> ... : a rehash happens *before* the insert
 *     that would make size exceed the bucket count (probed here with
 *     std::__detail::_Prime_rehash_policy; table below);
 *   - insert into an empty bucket puts the node at the global list head; into a non-empty bucket,
 *     at the head of that bucket's run;
 *   - rehash walks the old list in order and applies the same two rules.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const uint64_t BKT_SCHEDULE[] = {
    13ull, 29ull, 59ull, 127ull, 257ull, 541ull, 1109ull, 2357ull, 5087ull, 10273ull, 20753ull, 42043ull, 85229ull,
    172933ull, 351061ull, 712697ull, 1447153ull, 2938679ull, 5967347ull, 12117689ull, 24607243ull, 49969847ull,
    101473717ull, 206062531ull, 418451333ull, 849749479ull, 1725587117ull, 3504151727ull, 8589934583ull};

/* ---- emulated unordered_map<size_t, slot> for the voxel map -------------------------------- */
typedef struct {
    uint64_t* key; int64_t* next;       /* per node */
    int64_t* bucket;                    /* per bucket: node *before* the bucket's first node, -2 = empty, -1 = before_begin */
    int64_t head;                       /* before_begin.next */
    uint64_t nb; size_t n, cap; int sched;
} hsim;

static void hs_init(hsim* h, size_t cap)
{
    h->key = (uint64_t*)malloc((cap ? cap : 1) * sizeof(uint64_t));
    h->next = (int64_t*)malloc((cap ? cap : 1) * sizeof(int64_t));
    h->nb = 1; h->bucket = (int64_t*)malloc(sizeof(int64_t)); h->bucket[0] = -2;
    h->head = -1; h->n = 0; h->cap = cap; h->sched = -1;
}
static void hs_free(hsim* h) { free(h->key); free(h->next); free(h->bucket); }

static int64_t hs_find(const hsim* h, uint64_t k)
{
    int64_t prev = h->bucket[k % h->nb];
    if (prev == -2) return -1;
    int64_t p = prev == -1 ? h->head : h->next[prev];
    while (p >= 0 && h->key[p] % h->nb == k % h->nb) { if (h->key[p] == k) return p; p = h->next[p]; }
    return -1;
}
static void hs_link(hsim* h, int64_t* bucket, uint64_t nb, int64_t node)
{
    uint64_t b = h->key[node] % nb;
    if (bucket[b] != -2) {                       /* head of the bucket's run */
        int64_t prev = bucket[b];
        if (prev == -1) { h->next[node] = h->head; h->head = node; }
        else { h->next[node] = h->next[prev]; h->next[prev] = node; }
    } else {                                     /* global head */
        h->next[node] = h->head; h->head = node;
        if (h->next[node] >= 0) bucket[h->key[h->next[node]] % nb] = node;
        bucket[b] = -1;
    }
}
static void hs_rehash(hsim* h, uint64_t nb)
{
    int64_t* nbk = (int64_t*)malloc(nb * sizeof(int64_t));
    for (uint64_t i = 0; i < nb; ++i) nbk[i] = -2;
    int64_t p = h->head; h->head = -1;
    while (p >= 0) { int64_t nx = h->next[p]; hs_link(h, nbk, nb, p); p = nx; }
    free(h->bucket); h->bucket = nbk; h->nb = nb;
}
static int64_t hs_insert(hsim* h, uint64_t k)    /* key known to be absent */
{
    if (h->n + 1 > h->nb || h->sched < 0) { h->sched++; hs_rehash(h, BKT_SCHEDULE[h->sched]); }
    int64_t node = (int64_t)h->n++;
    h->key[node] = k;
    hs_link(h, h->bucket, h->nb, node);
    return node;
}

/* ---- emulated unordered_map<int,int> (label histogram) as an array in iteration order ------- */
typedef struct { int32_t* lab; int32_t* cnt; int32_t n, cap, sched; } lhist;

static uint64_t lab_bucket(int32_t l, uint64_t nb) { return (uint64_t)(int64_t)l % nb; }   /* std::hash<int> = value */

static void lh_add(lhist* h, int32_t l)
{
    for (int32_t i = 0; i < h->n; ++i) if (h->lab[i] == l) { h->cnt[i]++; return; }
    if (h->n == h->cap) {
        h->cap = h->cap ? 2 * h->cap : 4;
        h->lab = (int32_t*)realloc(h->lab, h->cap * sizeof(int32_t));
        h->cnt = (int32_t*)realloc(h->cnt, h->cap * sizeof(int32_t));
    }
    if (h->n == 0) h->sched = 0;
    else if ((uint64_t)h->n + 1 > BKT_SCHEDULE[h->sched]) {
        /* rehash: runs ordered by first occurrence, members in list order; then the whole list reversed */
        uint64_t nb = BKT_SCHEDULE[++h->sched];
        int32_t n = h->n; int32_t* tl = (int32_t*)malloc(n * sizeof(int32_t)); int32_t* tc = (int32_t*)malloc(n * sizeof(int32_t));
        char* used = (char*)calloc(n, 1); int32_t w = 0;
        for (int32_t i = 0; i < n; ++i) if (!used[i]) {
            uint64_t b = lab_bucket(h->lab[i], nb);
            for (int32_t j = i; j < n; ++j) if (!used[j] && lab_bucket(h->lab[j], nb) == b) { used[j] = 1; tl[w] = h->lab[j]; tc[w] = h->cnt[j]; ++w; }
        }
        for (int32_t i = 0; i < n; ++i) { h->lab[i] = tl[n - 1 - i]; h->cnt[i] = tc[n - 1 - i]; }
        free(tl); free(tc); free(used);
    }
    uint64_t nb = BKT_SCHEDULE[h->sched], b = lab_bucket(l, nb);
    int32_t pos = 0;
    for (int32_t i = 0; i < h->n; ++i) if (lab_bucket(h->lab[i], nb) == b) { pos = i; break; }
    memmove(h->lab + pos + 1, h->lab + pos, (h->n - pos) * sizeof(int32_t));
    memmove(h->cnt + pos + 1, h->cnt + pos, (h->n - pos) * sizeof(int32_t));
    h->lab[pos] = l; h->cnt[pos] = 1; h->n++;
}

static int cmp_key(const void* a, const void* b)
{
    uint64_t x = ((const uint64_t*)a)[0], y = ((const uint64_t*)b)[0];
    return x < y ? -1 : (x > y ? 1 : 0);
}

long oracle_grid_subsampling(const float* pts, size_t n, const float* feats, size_t fdim,
                             const int32_t* cls, size_t ldim, float dl, int order,
                             float* out_pts, float* out_feats, int32_t* out_cls, uint64_t* out_keys)
{
    if (!n) return 0;
    if (!feats) fdim = 0;
    if (!cls) ldim = 0;
    /* cloud.cpp:27-67 */
    float mn[3] = {pts[0], pts[1], pts[2]}, mx[3] = {pts[0], pts[1], pts[2]};
    for (size_t i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            float v = pts[3 * i + d];
            if (v < mn[d]) mn[d] = v;
            if (v > mx[d]) mx[d] = v;
        }
    /* grid_subsampling.cpp:27-31 */
    float inv = 1 / dl, org[3];
    for (int d = 0; d < 3; ++d) org[d] = floorf(mn[d] * inv) * dl;
    uint64_t nx = (uint64_t)(int64_t)floorf((mx[0] - org[0]) / dl) + 1;
    uint64_t ny = (uint64_t)(int64_t)floorf((mx[1] - org[1]) / dl) + 1;

    hsim h; hs_init(&h, n);
    int32_t* count = (int32_t*)calloc(n, sizeof(int32_t));
    float* sxyz = (float*)calloc(3 * n, sizeof(float));
    float* sf = (float*)calloc(fdim ? fdim * n : 1, sizeof(float));
    lhist* lh = (lhist*)calloc(ldim ? ldim * n : 1, sizeof(lhist));

    for (size_t i = 0; i < n; ++i) {                      /* :50-77 */
        uint64_t ix = (uint64_t)(int64_t)floorf((pts[3 * i + 0] - org[0]) / dl);
        uint64_t iy = (uint64_t)(int64_t)floorf((pts[3 * i + 1] - org[1]) / dl);
        uint64_t iz = (uint64_t)(int64_t)floorf((pts[3 * i + 2] - org[2]) / dl);
        uint64_t key = ix + nx * iy + nx * ny * iz;
        int64_t v = hs_find(&h, key);
        if (v < 0) v = hs_insert(&h, key);
        count[v] += 1;
        for (int d = 0; d < 3; ++d) sxyz[3 * v + d] += pts[3 * i + d];
        for (size_t f = 0; f < fdim; ++f) sf[fdim * v + f] += feats[fdim * i + f];
        for (size_t l = 0; l < ldim; ++l) lh_add(&lh[ldim * v + l], cls[ldim * i + l]);
    }

    size_t M = h.n;
    /* row order */
    int64_t* rows = (int64_t*)malloc(M * sizeof(int64_t));
    if (order == 0) { size_t w = 0; for (int64_t p = h.head; p >= 0; p = h.next[p]) rows[w++] = p; }
    else {
        uint64_t* kv = (uint64_t*)malloc(2 * M * sizeof(uint64_t));
        for (size_t i = 0; i < M; ++i) { kv[2 * i] = h.key[i]; kv[2 * i + 1] = i; }
        qsort(kv, M, 2 * sizeof(uint64_t), cmp_key);
        for (size_t i = 0; i < M; ++i) rows[i] = (int64_t)kv[2 * i + 1];
        free(kv);
    }
    for (size_t r = 0; r < M; ++r) {                      /* :85-103 */
        int64_t v = rows[r];
        float a = (float)(1.0 / count[v]);
        for (int d = 0; d < 3; ++d) out_pts[3 * r + d] = sxyz[3 * v + d] * a;
        float c = (float)count[v];
        for (size_t f = 0; f < fdim; ++f) out_feats[fdim * r + f] = sf[fdim * v + f] / c;
        for (size_t l = 0; l < ldim; ++l) {
            lhist* q = &lh[ldim * v + l]; int32_t best = 0;
            for (int32_t i = 1; i < q->n; ++i) if (q->cnt[best] < q->cnt[i]) best = i;   /* std::max_element: first max */
            out_cls[ldim * r + l] = q->lab[best];
        }
        if (out_keys) out_keys[r] = h.key[v];
    }
    for (size_t i = 0; i < (ldim ? ldim * n : 0); ++i) { free(lh[i].lab); free(lh[i].cnt); }
    free(lh); free(sf); free(sxyz); free(count); free(rows); hs_free(&h);
    return (long)M;
}
