// TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// extern "C" driver for the *real* reference C++ ops.  This file holds no
// reference code: the reference translation units are compiled where they lie
// under /root/reference by oracle/Makefile (target `ref`) and linked with this
// driver into oracle/_ref/libssdr_ref.so.  The driver takes the place of the
// reference's CPython glue (utils/cpp_wrappers/cpp_subsampling/wrapper.cpp:58-276,
// which does not compile against NumPy 2.x headers, and the Cython module
// utils/nearest_neighbors/knn.pyx:33-149) so that tests and golden-vector
// generators can call the reference cores through ctypes.
#include <cstddef>
#include <cstring>
#include <ctime>
#include <vector>
#include "grid_subsampling/grid_subsampling.h"   // -I<ref>/utils/cpp_wrappers/cpp_subsampling
#include "knn_.h"                                // -I<ref>/utils/nearest_neighbors

extern "C" {

// Mirrors wrapper.cpp:202-232: copy into vectors, call grid_subsampling(), copy out.
// Returns M (number of occupied voxels).  Two-phase: call with out_* == NULL to size.
static std::vector<PointXYZ> g_sp; static std::vector<float> g_sf; static std::vector<int> g_sc;
long ref_grid_subsampling(const float* pts, size_t n, const float* feats, size_t fdim,
                          const int* cls, size_t ldim, float dl)
{
    std::vector<PointXYZ> op((const PointXYZ*)pts, (const PointXYZ*)pts + n);
    std::vector<float> of; if (feats && fdim) of.assign(feats, feats + n * fdim);
    std::vector<int> oc;   if (cls && ldim)   oc.assign(cls, cls + n * ldim);
    g_sp.clear(); g_sf.clear(); g_sc.clear();
    grid_subsampling(op, g_sp, of, g_sf, oc, g_sc, dl, 0);
    return (long)g_sp.size();
}
void ref_grid_subsampling_fetch(float* out_pts, float* out_feats, int* out_cls)
{
    if (out_pts)   memcpy(out_pts, g_sp.data(), g_sp.size() * sizeof(PointXYZ));
    if (out_feats && !g_sf.empty()) memcpy(out_feats, g_sf.data(), g_sf.size() * sizeof(float));
    if (out_cls && !g_sc.empty())   memcpy(out_cls, g_sc.data(), g_sc.size() * sizeof(int));
}

void ref_knn(const float* p, size_t np_, size_t dim, const float* q, size_t nq, size_t K, long* out)
{ cpp_knn(p, np_, dim, q, nq, K, out); }
void ref_knn_omp(const float* p, size_t np_, size_t dim, const float* q, size_t nq, size_t K, long* out)
{ cpp_knn_omp(p, np_, dim, q, nq, K, out); }
void ref_knn_batch(const float* p, size_t b, size_t np_, size_t dim, const float* q, size_t nq, size_t K, long* out)
{ cpp_knn_batch(p, b, np_, dim, q, nq, K, out); }
void ref_knn_batch_omp(const float* p, size_t b, size_t np_, size_t dim, const float* q, size_t nq, size_t K, long* out)
{ cpp_knn_batch_omp(p, b, np_, dim, q, nq, K, out); }


// cpp_knn_batch_distance_pick seeds std::mt19937 with time(0) (knn_.cxx:141).  The library is linked with
// -Wl,-Bsymbolic-functions, so the reference object's call to time() binds to this definition: tests pin the clock to get
// a reproducible run of the UNMODIFIED reference code (ref_set_time(-1) = real clock again).
static long g_fake_time = -1;
time_t time(time_t* t)
{
    time_t v;
    if (g_fake_time >= 0) v = (time_t)g_fake_time;
    else { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); v = ts.tv_sec; }
    if (t) *t = v;
    return v;
}
void ref_set_time(long v) { g_fake_time = v; }
void ref_knn_batch_distance_pick(const float* p, size_t b, size_t np_, size_t dim, float* q, size_t nq, size_t K, long* out)
{ cpp_knn_batch_distance_pick(p, b, np_, dim, q, nq, K, out); }

}
