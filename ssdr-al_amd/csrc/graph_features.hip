// Superpoint-graph inputs: the two k-NN structures and the per-point geometric features that the
// reference computes before cut-pursuit (partition/compute_superpoint.py:47-55).
//
//   ssdr_knn_graph_dev   = compute_graph_nn_2 (partition/graphs.py:23-70, voronoi == 0 branch): sklearn's exact k-NN on
//                          float64-widened coordinates; source / target / distances of the k_nn1 graph and the k_nn2
//                          targets.  Same kd forest as the hot path, walked with float64 bounds (kdtree.hip).
//   ssdr_geof_dev        = libply_c.compute_geof (partition/ply_c/ply_c.cpp:385-455): covariance of a point and its k
//                          neighbours, eigen-decomposition, linearity / planarity / scattering / verticality.
//                          The reference solves the 3x3 problem with Eigen::EigenSolver<Matrix3f>; here a cyclic Jacobi
//                          iteration in float64 on the float32 covariance (PARITY UNPINNED: Eigen and Boost are not in the
//                          build image, the reference cannot be compiled; tolerance against oracle/graph_np.py).
#include "ssdr_internal.hpp"
#include <map>
#include <cfloat>

namespace ssdr {
namespace {

struct GraphState { KdForest forest; DevBuf idx, d2; };
GraphState& gst(hipStream_t st = nullptr) { return per_stream<GraphState>(st); }      // one per stream

// neighbours [n][K] (first column = the point itself) -> the reference's flat arrays (graphs.py:33-38, :62-67)
__global__ __launch_bounds__(256) void graph_emit(const int* __restrict__ idx, const double* __restrict__ d2, int n, int K, int k1, int k2,
                                                  uint32_t* source, uint32_t* target, float* dist, uint32_t* target2) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)n * k2; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / k2), j = (int)(e % k2);
        const int nb = idx[(size_t)i * K + 1 + j];
        target2[e] = (uint32_t)nb;
        if (j < k1) {
            const size_t o = (size_t)i * k1 + j;
            source[o] = (uint32_t)i; target[o] = (uint32_t)nb;
            dist[o] = (float)sqrt(d2[(size_t)i * K + 1 + j]);          // kneighbors returns sqrt(rdist); .astype('float32')
        }
    }
}

// symmetric 3x3 eigen-decomposition, cyclic Jacobi in float64; eigenvalues descending, eigenvectors as columns v[:,k]
__device__ void eig3(double a[3][3], double lam[3], double v[3][3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-40) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            if (fabs(a[p][q]) < 1e-300) continue;
            const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - sn * akq; a[k][q] = sn * akp + c * akq; }
            for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - sn * aqk; a[q][k] = sn * apk + c * aqk; }
            for (int k = 0; k < 3; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - sn * vkq; v[k][q] = sn * vkp + c * vkq; }
        }
    }
    int o[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i) lam[i] = a[i][i];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2 - i; ++j) if (lam[o[j]] < lam[o[j + 1]]) { const int t = o[j]; o[j] = o[j + 1]; o[j + 1] = t; }
    double l2[3], v2[3][3];
    for (int k = 0; k < 3; ++k) { l2[k] = lam[o[k]]; for (int i = 0; i < 3; ++i) v2[i][k] = v[i][o[k]]; }
    for (int k = 0; k < 3; ++k) { lam[k] = l2[k]; for (int i = 0; i < 3; ++i) v[i][k] = v2[i][k]; }
}

__global__ __launch_bounds__(256) void geof_kernel(const float* __restrict__ xyz, const uint32_t* __restrict__ target, int n, int k, float* __restrict__ geof) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        // mean and covariance of the k+1 positions in float32 like the reference's MatrixXf (ply_c.cpp:398-417)
        float mx = xyz[3 * (size_t)i], my = xyz[3 * (size_t)i + 1], mz = xyz[3 * (size_t)i + 2];
        for (int j = 0; j < k; ++j) { const size_t t = target[(size_t)i * k + j]; mx += xyz[3 * t]; my += xyz[3 * t + 1]; mz += xyz[3 * t + 2]; }
        const float inv = 1.0f / (float)(k + 1);
        mx *= inv; my *= inv; mz *= inv;
        float c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
        for (int j = -1; j < k; ++j) {
            const size_t t = j < 0 ? (size_t)i : (size_t)target[(size_t)i * k + j];
            const float x = xyz[3 * t] - mx, y = xyz[3 * t + 1] - my, z = xyz[3 * t + 2] - mz;
            c00 += x * x; c01 += x * y; c02 += x * z; c11 += y * y; c12 += y * z; c22 += z * z;
        }
        double a[3][3] = {{(double)(c00 * inv), (double)(c01 * inv), (double)(c02 * inv)},
                          {(double)(c01 * inv), (double)(c11 * inv), (double)(c12 * inv)},
                          {(double)(c02 * inv), (double)(c12 * inv), (double)(c22 * inv)}};
        double lam[3], v[3][3];
        eig3(a, lam, v);
        const float l0 = fmaxf((float)lam[0], 0.f), l1 = fmaxf((float)lam[1], 0.f), l2 = fmaxf((float)lam[2], 0.f);     // :427-429
        const float s0 = sqrtf(l0), s1 = sqrtf(l1), s2 = sqrtf(l2);
        const float linearity = (s0 - s1) / s0, planarity = (s1 - s2) / s0, scattering = s2 / s0;                    // :440-442
        float u[3];
        for (int d = 0; d < 3; ++d) u[d] = l0 * fabsf((float)v[d][0]) + l1 * fabsf((float)v[d][1]) + l2 * fabsf((float)v[d][2]);   // :444-447
        const float norm = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        geof[4 * (size_t)i] = linearity; geof[4 * (size_t)i + 1] = planarity; geof[4 * (size_t)i + 2] = scattering; geof[4 * (size_t)i + 3] = u[2] / norm;
    }
}

}  // namespace
}  // namespace ssdr

using namespace ssdr;

extern "C" {

int ssdr_knn_graph_dev(const float* d_xyz, size_t n, size_t k_nn1, size_t k_nn2, uint32_t* d_source, uint32_t* d_target, float* d_distances,
                       uint32_t* d_target2, void* stream) {
    if (!d_xyz || !d_source || !d_target || !d_distances || !d_target2) { set_error("knn_graph: NULL argument"); return SSDR_ERR_INVALID; }
    if (k_nn1 == 0 || k_nn1 > k_nn2) { set_error("knn1 must be smaller than knn2"); return SSDR_ERR_INVALID; }      // graphs.py:27
    if (n <= k_nn2 || n > 0x3fffffff || k_nn2 + 1 > 128) { set_error("knn_graph: need k_nn2 < n and k_nn2 <= 127"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream);
    GraphState& G = gst(s);
    const int K = (int)k_nn2 + 1;
    std::vector<KdTreeDesc> trees(1); trees[0].pts = d_xyz; trees[0].n = (int)n;
    SSDR_TRY(kd_build(G.forest, trees, s));
    SSDR_TRY(G.idx.reserve(4 * n * K)); SSDR_TRY(G.d2.reserve(8 * n * K));
    SSDR_TRY(kd_search_f64(G.forest, 0, 1, d_xyz, n * 3, (int)n, K, 0, G.idx.as<int32_t>(), G.d2.as<double>(), n * K, s));
    const size_t tot = n * k_nn2;
    hipLaunchKernelGGL(graph_emit, dim3((unsigned)std::min<size_t>((tot + 255) / 256, 8192)), dim3(256), 0, s, G.idx.as<int>(), G.d2.as<double>(), (int)n, K,
                       (int)k_nn1, (int)k_nn2, d_source, d_target, d_distances, d_target2);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_geof_dev(const float* d_xyz, size_t n, const uint32_t* d_target, size_t k_nn, float* d_geof, void* stream) {
    if (!d_xyz || !d_target || !d_geof || k_nn == 0) { set_error("geof: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (n == 0) return SSDR_OK;
    hipLaunchKernelGGL(geof_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, pick_stream(stream), d_xyz, d_target, (int)n, (int)k_nn, d_geof);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

}
